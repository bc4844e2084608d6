#!/bin/sh
# A/B of build variants on the GPU box: sh tests/gpu_variant_probe.sh build/variants/*.so
for f in "$@"; do
  COMPRESSJS_AMD_LIB=$PWD/$f timeout 120 python bench.py --steps 4 --cpu-sample 1000000 2>/dev/null | tail -1 | python -c "
import sys,json
try:
    j=json.loads(sys.stdin.read()); print('$f', j['ms_per_step'], 'scatter', j['roofline']['avg_launch_ms'], j['config']['bit_exact_vs_oracle_prefix_and_roundtrip'])
except Exception as e:
    print('$f', 'FAILED', e)"
done
