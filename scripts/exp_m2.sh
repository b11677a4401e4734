#!/bin/bash
# A/B of variant libraries on mode M2 (GPU box): bash scripts/exp_m2.sh <lib.so|default> ...
cd "$(dirname "$0")/.."
for lib in "$@"; do
  if [ "$lib" = default ]; then unset LQG_HIP_LIB; else export LQG_HIP_LIB=$PWD/$lib; fi
  echo "== $lib"
  python bench_m2.py --log2-batch 17 --reps 5 2>&1 | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        j = json.loads(line); print('ms %.2f frac %.3f parity %s' % (j['ms_per_pass'], j['roofline']['frac'], max(j['parity_rel_maxnorm_vs_fp64_oracle'].values())))
    else: print(line.rstrip()[-300:])
"
done
