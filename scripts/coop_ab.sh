#!/bin/bash
# lane-per-system vs cooperative kernels on BASELINE configs 2 and 4 (few systems x many trials); run on the GPU box
cd "$(dirname "$0")/.."
for c in 0 1; do
  LQG_COOP=$c timeout 600 python bench_configs.py --configs ${1:-2,4} --reps 5 > gpurun_out/cfg_coop$c.jsonl 2> gpurun_out/cfg_coop$c.err
  tail -3 gpurun_out/cfg_coop$c.err | grep -v amdgpu.ids
done
python - <<PY
import json
for c in (0,1):
    for line in open("gpurun_out/cfg_coop%d.jsonl"%c):
        j=json.loads(line)
        if "riccati_ms" in j: print("COOP=%d cfg%d %s ric %.3f fwd %.3f trial %.3f wall %.3f err %.2e %s"%(c,j["config"],j["dtype"],j["riccati_ms"],j["forward_ms"],j["trial_ms"],j["wall_ms"],j["max_rel_err_vs_fp64_oracle"],j["path"][:60]))
PY
