cd ${GRAFT_REPO_ROOT:-.}
python tests/gpu_r2_probe.py gen e8sa enwik >/dev/null 2>&1
for cfg in "" "CJS_DEEP_BIG_DIV=1" "CJS_DEEP_BIG_DIV=1 CJS_TEXT_BYTES=40" "CJS_DEEP_BIG_DIV=1 CJS_TEXT_BYTES=64" "CJS_DEEP_BIG_DIV=1 CJS_TEXT_BYTES=112" "CJS_DEEP_BIG_DIV=1 CJS_TEXT_BYTES=160"; do
  env $cfg python tests/gpu_r2_probe.py run e8sa --reps 6 2>&1 | grep "^\[" | cut -c1-200
done
