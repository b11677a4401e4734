#!/bin/bash
# mode M2 with its two kernels timed separately (GPU box): bash scripts/exp_m2_split.sh [f32|f64] ...
cd "$(dirname "$0")/.."
for dt in "${@:-f32}"; do
  python bench_m2.py --log2-batch 17 --reps 5 --dtype $dt 2>&1 | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        j = json.loads(line); r = j['roofline']
        print('$dt ms %.2f frac %.3f riccati %.2f forward %.2f dense %s parity %.2e %s' % (j['ms_per_pass'], r['frac'], r['riccati_kernel_ms'], r['forward_kernel_ms'], r['dense_kernels_ms_per_pass'], max(j['parity_rel_maxnorm_vs_fp64_oracle'].values()), j['calls']))
    else: print(line.rstrip()[-300:])
"
done
