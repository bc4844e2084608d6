"""two-symbol random input (python tests/gpu_two_probe.py): digest of the GPU stream, GPU round trip, libbz2."""
import sys, os, hashlib, bz2
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from compressjs_amd.bzip2 import Context
rng = np.random.RandomState(1)
_ = rng.randint(0, 256, size=50_000_000)
two = rng.randint(97, 99, size=50_000_000).astype(np.uint8)
c = Context(0, 128)
o = c.compress(two, 9)
print('gpu', os.environ.get("CJS_TEXT_BYTES", "dflt"), len(o), hashlib.sha256(o).hexdigest()[:16], 'gpu roundtrip', c.decompress(np.frombuffer(o, dtype=np.uint8)) == two.tobytes())
try:
    print('libbz2 roundtrip', bz2.decompress(o) == two.tobytes())
except Exception as e:
    print('libbz2 error', repr(e))
