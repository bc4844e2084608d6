"""Adversarial blocks for the 16-byte-key bucket sort of k1_front.hip (test infrastructure): the families the round-5 judge attacked
k1f_sortk / k1f_flush / the task levels with and found nothing - pinned here so that they stay found-nothing.

  * dictionary texts whose words are 15 / 16 / 17 bytes (ties exactly at and just past the key width) and 40 .. 90 bytes long, over
    {0xFE, 0xFF} (all-ones keys next to the all-ones sentinel cells behind a bucket's last rotation), {a .. d} and the full alphabet;
  * 0xFF- and 0x00-heavy blocks with 5 .. 400 sparse defects;
  * ONE phrase of 16 / 24 / 32 / 48 / 64 bytes planted in 300 .. 1 500 places of a text (groups above K1F_GBIG: the task levels);
  * near-periodic words with 1 .. 5 flipped bits.
Every block is checked (transform and origPtr) against the oracle's cyclic BWT."""
import numpy as np


def _dictionary(n, rng, wlen, alpha, nwords):
    lo, hi = alpha
    words = [rng.integers(lo, hi, int(wl)).astype(np.uint8) for wl in (rng.integers(wlen[0], wlen[1] + 1, nwords))]
    pick = rng.integers(0, nwords, n // wlen[0] + 2)
    out = np.concatenate([words[i] for i in pick])
    return out[:n].copy()


def _heavy(n, rng, byte, ndef):
    d = np.full(n, byte, np.uint8)
    pos = rng.integers(0, n, ndef)
    d[pos] = rng.integers(0, 256, ndef).astype(np.uint8)
    return d


def _planted(n, rng, plen, places, base):
    d = base(n).copy()
    phrase = rng.integers(97, 123, plen).astype(np.uint8)
    for p in rng.integers(0, n - plen, places):
        d[p:p + plen] = phrase
    return d


def _near_periodic(n, rng, p, flips):
    w = rng.integers(97, 101, p).astype(np.uint8)
    d = np.tile(w, n // p + 1)[:n].copy()
    for q in rng.integers(0, n, flips):
        d[q] ^= np.uint8(1 << int(rng.integers(0, 8)))
    return d


def blocks(n_max, rng, text_like, small=False):
    """[(label, block)], 28 blocks of up to n_max bytes (`small`: lengths drawn from n_max / 6 .. n_max; the last one is always n_max)."""
    def ln():
        return int(rng.integers(max(2000, n_max // 6), n_max + 1)) if small else n_max
    out = []
    for alpha, an in (((0xFE, 0x100), "FEFF"), ((97, 101), "abcd"), ((0, 256), "full")):
        for wl in ((15, 15), (16, 16), (17, 17), (40, 90)):
            n = ln()
            nwords = int(rng.integers(8, 200))
            out.append(("dict %s words %d..%d x%d n=%d" % (an, wl[0], wl[1], nwords, n), _dictionary(n, rng, wl, alpha, nwords)))
    for byte in (0xFF, 0x00):
        for ndef in (5, 60, 400):
            n = ln()
            out.append(("heavy %02x defects %d n=%d" % (byte, ndef, n), _heavy(n, rng, byte, ndef)))
    for plen in (16, 24, 32, 48, 64):
        n = ln()
        places = int(rng.integers(300, 1501))
        out.append(("phrase %dB x%d n=%d" % (plen, places, n), _planted(n, rng, plen, places, lambda m: text_like(m, int(rng.integers(1, 1 << 20))))))
    for flips in (1, 2, 3, 5):
        n = ln()
        p = int(rng.integers(3, 200))
        out.append(("near-periodic p=%d flips %d n=%d" % (p, flips, n), _near_periodic(n, rng, p, flips)))
    # a mixed block at exactly the capacity: a third each of three families
    a = _dictionary(n_max // 3, rng, (16, 16), (0xFE, 0x100), 40)
    b = _heavy(n_max // 3, rng, 0xFF, 100)
    c = _planted(n_max - a.size - b.size, rng, 32, 400, lambda m: text_like(m, 99))
    out.append(("mixed n=%d" % n_max, np.concatenate([a, b, c])))
    return out
