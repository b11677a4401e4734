#!/bin/bash
# config 3 forward: k_trial_sp (scalar-cache operators, 8 workgroups per candidate) vs k_trial_lds (LDS-staged, one workgroup per candidate)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
for v in -1 0 2 4 5; do
  LQG_TRIAL_LDS=$v python3 bench.py --config 3 --steps 10 --warmup 3 > gpurun_out/r05_c3_lds$v.json 2> gpurun_out/r05_c3_lds$v.err
  python3 - gpurun_out/r05_c3_lds$v.json $v <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("TRIAL_LDS=" + sys.argv[2], "ms_per_step", round(d["ms_per_step"], 3), "phase_ms", d["phase_ms"], "checksum", repr(d["objective_checksum"]), "max", repr(d["objective_max"]))
PY
done
