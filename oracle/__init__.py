"""CPU oracle for the bzip2 block pipeline -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product (compressjs_amd/) never does.  See bz2_oracle.c for the parity statement."""
from .oracle import *  # noqa: F401,F403
