"""compressjs_amd -- MI355X-native bzip2 block pipeline behind the compressjs module API.

`Bzip2.compressFile` / `BWT.bwtransform2` mirror the reference (cscott/compressjs) entry points
and run entirely as HIP kernels on gfx950 through the C ABI of libcompressjs_amd.so.
There is no CPU fallback: without the library or without a GPU these raise."""
from .bzip2 import BWT, BWTC, Bzip2, Context, HuffmanAllocator, default_context  # noqa: F401

version = "0.1.0"
