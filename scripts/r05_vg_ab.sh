#!/bin/bash
# value+gradient config-3 leg on the default adjoint libraries and on a variant directory (same box, back to back)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = default ]; then unset LQG_PAT_DIR; else export LQG_PAT_DIR=/root/repo/variants/$v; fi
  for rep in 1 2; do
    python3 bench.py --only value_and_grad_config3 2>/dev/null | python3 -c "
import json,sys
d=json.load(sys.stdin)['value_and_grad_config3']
print('$v', round(d['ms_per_value_and_grad'],3), {k: round(x,3) for k,x in d['kernel_ms'].items()}, d.get('parity',{}).get('bars_max_rel_diff_split_vs_round1_kernels_f64'))"
  done
done
