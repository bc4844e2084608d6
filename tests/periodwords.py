"""Adversarial PERIOD WORDS for k1_period.hip (test infrastructure): blocks T[i] = P[i mod p] whose period word P is chosen to stress the
closed form (p <= 64) and the three-period reduction (64 < p <= n/4) - long borders, near-periods inside the period, one defect -
at lengths n = 0, 1, p - 1 (mod p).  VERDICT r4 item 4: the families the judge attacked the kernels with, pinned in the suites."""
import numpy as np


def fib_word(n):
    a, b = b"b", b"a"
    while len(b) < n:
        a, b = b, b + a
    return np.frombuffer(b[:n], np.uint8).copy()


def thue_morse(n):
    i = np.arange(n, dtype=np.uint32)
    c = np.zeros(n, np.uint32)
    while i.any():
        c ^= i & 1
        i >>= 1
    return (c + 97).astype(np.uint8)


def period_words(p, rng):
    """Words of length exactly p (primitive or not: a word that is a power has a shorter true period, which is part of the attack)."""
    out = [("fib", fib_word(p)), ("thue", thue_morse(p))]
    wlen = int(rng.integers(1, 9))
    w = rng.integers(97, 100, wlen).astype(np.uint8)
    d = np.tile(w, p // wlen + 1)[:p].copy()
    d[int(rng.integers(0, p))] ^= 1                       # (w)^m with one defect
    out.append(("pow_defect", d))
    d = np.full(p, 97, np.uint8)
    d[-1] = 98                                            # a^k b
    out.append(("a_k_b", d))
    u = rng.integers(97, 99, max(1, p // 3)).astype(np.uint8)
    v = rng.integers(97, 123, p - 2 * u.size).astype(np.uint8)
    out.append(("uvu", np.concatenate([u, v, u])[:p]))     # a long border
    q = int(rng.integers(2, min(70, p) + 1))
    d = np.tile(rng.integers(0, 4, q).astype(np.uint8), p // q + 1)[:p].copy()
    d[int(rng.integers(0, p))] ^= 0x55                    # a short period with one flipped byte
    out.append(("short_flip", d))
    runs = rng.integers(97, 101, (p + 2) // 3).astype(np.uint8)
    out.append(("run_triples", np.repeat(runs, 3)[:p]))
    out.append(("binary_noise", rng.integers(0, 2, p).astype(np.uint8)))
    return out


def blocks(n_max, rng, count, small=False):
    """[(label, block)]: period words tiled to lengths n <= n_max with n = 0, 1, p - 1 (mod p); p from 2 .. 64 (closed form), 65 .. 200,
    just below n / 4 and random in between (three-period reduction)."""
    out = []
    k = 0
    while len(out) < count:
        kind = k % 5
        k += 1
        n0 = n_max if not small else int(rng.integers(max(600, n_max // 4), n_max + 1))
        if kind == 0:
            p = int(rng.integers(2, 65))
        elif kind == 1:
            p = int(rng.integers(65, 201))
        elif kind == 2:
            p = n0 // 4 - int(rng.integers(0, 5))
        elif kind == 3:
            p = int(rng.integers(201, max(202, n0 // 4)))
        else:
            p = int(rng.integers(65, max(66, n0 // 8)))
        words = period_words(p, rng)
        name, w = words[int(rng.integers(0, len(words)))]
        res = (0, 1, p - 1)[int(rng.integers(0, 3))]
        n = (n0 // p) * p + res
        if n > n_max:
            n -= p
        if n < 2 * p:
            continue
        d = np.tile(w, n // p + 2)[:n].copy()
        out.append(("%s p=%d n=%d" % (name, p, n), d))
    return out
