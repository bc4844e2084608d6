"""Named byte streams of SURVEY.md section 8(d), shared by bench.py, the parity tests and
tests/golden/make_golden_big.py so that all three see exactly the same bytes.  Test/bench infrastructure (it reads
the staged reference fixtures), hence under tests/ and not in the product package.

  enwik  synth.enwik_like(n, 2025)   synthetic words + wiki markup + phrase reuse (ratio 0.29)
  text   synth.text_like(n, 2025)    the same without phrase reuse (ratio 0.38)
  lcg    synth.lcg_ascii(n, 7)       random printable ASCII, BASELINE.json configs[3]
  e8sa   test/sample5.ref || test/sample4.ref tiled to n bytes (E8S-A, SURVEY.md 8d "headline")
  e8sb   order-2 Markov chain trained on the same 3 MB (E8S-B, "control")

e8sa/e8sb need the reference's test fixtures (staged under oracle/_ref/fixtures by
__graft_entry__.build(); never committed).  Nothing here reads /root/reference at run time
on the GPU box: the fixtures travel with the snapshot."""
from __future__ import annotations

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from compressjs_amd import synth  # noqa: E402
FIXTURE_DIRS = [os.path.join(ROOT, "oracle", "_ref", "fixtures"), "/root/reference/test"]
NAMES = ("enwik", "text", "lcg", "e8sa", "e8sb")

DESCRIPTIONS = {
    "enwik": "synthetic enwik8-shaped text with phrase reuse calibrated to enwik8's bzip2 -9 ratio "
             "(compressjs_amd.synth.enwik_like, seed 2025)",
    "text": "synthetic enwik8-shaped text without phrase reuse (compressjs_amd.synth.text_like, seed 2025)",
    "lcg": "LCG(n, seed 7) random printable ASCII (BASELINE.json configs[3])",
    "e8sa": "test/sample5.ref || test/sample4.ref tiled (SURVEY.md 8d E8S-A)",
    "e8sb": "order-2 Markov chain trained on test/sample5.ref || test/sample4.ref (SURVEY.md 8d E8S-B)",
}


def fixture(name: str):
    for d in FIXTURE_DIRS:
        p = os.path.join(d, name)
        if os.path.exists(p):
            return p
    return None


def have_fixtures() -> bool:
    return bool(fixture("sample5.ref") and fixture("sample4.ref"))


def _e8_base() -> np.ndarray:
    parts = [fixture("sample5.ref"), fixture("sample4.ref")]
    if not all(parts):
        raise FileNotFoundError("e8sa/e8sb need test/sample5.ref and test/sample4.ref "
                                "(staged by __graft_entry__.build())")
    return np.concatenate([np.fromfile(p, dtype=np.uint8) for p in parts])


def e8s_a(n: int) -> np.ndarray:
    base = _e8_base()
    return np.tile(base, n // base.size + 1)[:n].copy()


_E8SB_LANES = 4096


def e8s_b(n: int, seed: int = 0x2545F491) -> np.ndarray:
    """Order-2 Markov chain over the bytes of sample5||sample4.  To be generated in seconds it runs
    4096 independent chains (xorshift32 per chain, seeds seed+lane) whose outputs are laid out one after
    another; the next byte is looked up in a 256-slot quantile table of the context's successor counts
    (largest-remainder rounding), i.e. probabilities at a resolution of 1/256."""
    base = _e8_base()
    ctx = (base[:-2].astype(np.int64) << 8) | base[1:-1]
    nxt = base[2:].astype(np.int64)
    counts = np.zeros((65536, 256), dtype=np.int64)
    np.add.at(counts, (ctx, nxt), 1)
    tot = counts.sum(axis=1)
    table = np.zeros((65536, 256), dtype=np.uint8)
    fallback = np.argsort(-np.bincount(base, minlength=256), kind="stable")[:1]
    used = np.nonzero(tot)[0]
    # quantile table per used context: slot k holds the successor whose cumulative share covers k/256
    for c in used:
        cnt = counts[c]
        share = cnt * 256
        q = share // tot[c]
        rem = share - q * tot[c]
        short = 256 - int(q.sum())
        if short:
            order = np.lexsort((np.arange(256), -rem))       # largest remainder, ties by byte value
            q[order[:short]] += 1
        table[c] = np.repeat(np.arange(256, dtype=np.uint8), q)
    table[tot == 0] = fallback[0]
    lanes = _E8SB_LANES
    per = (n + lanes - 1) // lanes
    out = np.empty((lanes, per), dtype=np.uint8)
    s = (np.uint32(seed) + np.arange(lanes, dtype=np.uint32) * np.uint32(0x9E3779B1)) | np.uint32(1)
    c = ctx[(np.arange(lanes, dtype=np.int64) * 7919) % ctx.size].copy()
    with np.errstate(over="ignore"):
        for k in range(per):
            s ^= s << np.uint32(13)
            s ^= s >> np.uint32(17)
            s ^= s << np.uint32(5)
            b = table[c, (s >> np.uint32(11)) & np.uint32(255)]
            out[:, k] = b
            c = ((c << 8) | b) & 0xFFFF
    return out.reshape(-1)[:n].copy()


def stream(name: str, n: int) -> np.ndarray:
    if name == "enwik":
        return synth.enwik_like(n, 2025)
    if name == "text":
        return synth.text_like(n, 2025)
    if name == "lcg":
        return synth.lcg_ascii(n, 7)
    if name == "e8sa":
        return e8s_a(n)
    if name == "e8sb":
        return e8s_b(n)
    raise KeyError(name)


def document(name: str, n: int, r: int) -> np.ndarray:
    """Multi-GPU runs (bench.py --gpus N): the job's stream is N documents of n bytes one after another, document r being
    rank r's slice.  Document 0 is stream(name, n); the others differ by seed (or are the next n bytes of the tiling / of
    the LCG sequence), so a rank can name its own bytes - and the margin in front of them - without generating the whole job."""
    if r == 0:
        return stream(name, n)
    if name == "enwik":
        return synth.enwik_like(n, 2025 + r)
    if name == "text":
        return synth.text_like(n, 2025 + r)
    if name == "lcg":
        return synth.lcg_ascii(n * (r + 1), 7)[n * r:]
    if name == "e8sa":
        base = _e8_base()
        k0 = (n * r) % base.size
        return np.tile(base, (k0 + n) // base.size + 1)[k0:k0 + n].copy()
    if name == "e8sb":
        return e8s_b(n, seed=0x2545F491 + 0x10001 * r)
    raise KeyError(name)


def world_stream(name: str, n: int, world: int) -> np.ndarray:
    return np.concatenate([document(name, n, r) for r in range(world)]) if world > 1 else stream(name, n)


def window(name: str, n: int, r: int, margin: int) -> np.ndarray:
    """Bytes [r*n - margin, (r+1)*n) of world_stream (clipped at 0): rank r's slice and the tail of what precedes it."""
    parts = [document(name, n, r)]
    need, q = min(margin, r * n), r - 1
    while need > 0:
        d = document(name, n, q)
        parts.insert(0, d[-need:] if need < d.size else d)
        need -= min(need, d.size)
        q -= 1
    return np.concatenate(parts) if len(parts) > 1 else parts[0]


def window_after(name: str, n: int, r: int, margin: int, world: int) -> np.ndarray:
    """Bytes [r*n, min(world*n, (r+1)*n + margin)) of world_stream: rank r's slice and the head of what follows it (the parallel plan
    of compressjs_amd/dist.py: a rank owns the blocks that START in its slice, the margin completes the last of them)."""
    parts = [document(name, n, r)]
    need, q = (min(margin, (world - 1 - r) * n), r + 1)
    while need > 0 and q < world:
        d = document(name, n, q)
        parts.append(d[:need] if need < d.size else d)
        need -= min(need, d.size)
        q += 1
    return np.concatenate(parts) if len(parts) > 1 else parts[0]
