#!/bin/bash
# headline with the trajectories laid [T+1][B][trial][component] (one 16-byte load per lane and step) against the default
# [trial][T+1][component][B] storage, same box, alternating
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for rep in 1 2 3; do
  for v in 0 1; do
    LQG_X4_LAYOUT=$v python3 bench.py --no-extra --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('X4_LAYOUT=$v', 'solves/s %.4g' % d['value'], 'ms/step %.3f' % d['ms_per_step'], 'forward %.3f' % r['kernel_ms'], 'riccati %.3f' % r['riccati_kernel_ms'], 'frac %.3f' % r['frac'], 'parity', d['parity']['max_rel_err_vs_fp64_oracle'], 'objective', repr(d['objective_sum']))"
  done
done
