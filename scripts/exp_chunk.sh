#!/bin/bash
# A/B of the checkpointed-gain variants of the structure-specialised hot path (LQG_SP_CHUNK = 0: L_t streamed
# through HBM; k > 0: S checkpoints every k steps, gains recomputed in the forward kernel).  Run on the GPU box:
#   bash scripts/exp_chunk.sh "0 4 8 16" [f32|f64]
# Each variant compiles its pattern libraries into its own directory (hipcc on the box) and runs bench.py.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/chunk
DT=${2:-f32}
for ck in $1; do
  export LQG_PAT_DIR=$PWD/gpurun_out/chunk/pat_ck$ck
  export LQG_SP_FLAGS="-DLQG_SP_CHUNK=$ck"
  python bench.py --steps 20 --warmup 3 --dtype $DT --no-extra --no-cpu-baseline > gpurun_out/chunk/bench_ck${ck}_$DT.json 2> gpurun_out/chunk/bench_ck${ck}_$DT.err
  python - <<PY
import json
try:
    j=json.load(open("gpurun_out/chunk/bench_ck${ck}_$DT.json"))
    r=j["roofline"]
    print("CK=$ck $DT value %.4g solves/s  ms/step %.3f  fwd %.3f ms  ric %.3f ms  parity %s" % (j["value"], j["ms_per_step"], r["kernel_ms"], r["riccati_kernel_ms"], j.get("parity")))
except Exception as e:
    print("CK=$ck failed", e); print(open("gpurun_out/chunk/bench_ck${ck}_$DT.err").read()[-2000:])
PY
done
