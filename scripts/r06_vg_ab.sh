#!/bin/bash
# A/B on ONE box: value+gradient at the headline shape, the round-5 reverse sweep (variants/old_tree: one system kernel, 438 VGPRs)
# against the round-6 cut (k_asp_sys_rev + k_asp_kal_rev), alternating; then a kernel trace of the new one.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r06_ab
for rep in 1 2 3; do
  for which in old new; do
    if [ $which = old ]; then (cd variants/old_tree && python bench.py --only value_and_grad_headline) > gpurun_out/r06_ab/${which}_$rep.json 2> gpurun_out/r06_ab/${which}_$rep.err
    else python bench.py --only value_and_grad_headline > gpurun_out/r06_ab/${which}_$rep.json 2> gpurun_out/r06_ab/${which}_$rep.err; fi
    python - <<PY
import json
try:
    v=json.load(open("gpurun_out/r06_ab/${which}_$rep.json"))["value_and_grad_headline"]
    print("$which $rep  %.3f ms " % v["ms_per_value_and_grad"], {k: round(x,3) for k,x in v["kernel_ms"].items()})
except Exception as e:
    print("$which $rep failed", e, open("gpurun_out/r06_ab/${which}_$rep.err").read()[-800:])
PY
  done
done
