#!/bin/bash
# A/B of compile-flag variants of the structure-specialised hot path on the headline workload (GPU box):
#   bash scripts/exp_flags.sh "name1:-DFOO=1 -DBAR=2" "name2:..." ...
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/flags
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  export LQG_PAT_DIR=$PWD/gpurun_out/flags/pat_$name
  export LQG_SP_FLAGS="$flags"
  python bench.py --steps 20 --warmup 3 --no-extra --no-cpu-baseline > gpurun_out/flags/bench_$name.json 2> gpurun_out/flags/bench_$name.err
  python - <<PY
import json
try:
    j=json.load(open("gpurun_out/flags/bench_$name.json")); r=j["roofline"]
    print("$name [$flags] value %.4g ms/step %.3f fwd %.3f ric %.3f parity %s" % (j["value"], j["ms_per_step"], r["kernel_ms"], r["riccati_kernel_ms"], j["parity"]["max_rel_err_vs_fp64_oracle"]))
except Exception as e:
    print("$name failed", e); print(open("gpurun_out/flags/bench_$name.err").read()[-1500:])
PY
done
