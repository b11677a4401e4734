#!/bin/bash
# the time-varying log-likelihood leg under build variants of k_forward_tv_sp (compiled on the box):
#   bash scripts/r06_tv_variants.sh f64 "tag:flags" ...
cd "$(dirname "$0")/.."
DT=$1; shift
mkdir -p gpurun_out/tvvar
for spec in "$@"; do
  tag=${spec%%:*}; flags=${spec#*:}
  export LQG_PAT_DIR=$PWD/gpurun_out/tvvar/pat_$tag
  mkdir -p $LQG_PAT_DIR
  export LQG_SP_FLAGS="$flags"
  python bench.py --only timevarying_$DT > gpurun_out/tvvar/${tag}_$DT.json 2> gpurun_out/tvvar/${tag}_$DT.err
  python - <<PY
import json
try:
    v=json.load(open("gpurun_out/tvvar/${tag}_$DT.json"))["timevarying_$DT"]
    for w in ("costs_stay_psd","per_entry_jitter"):
        l=v[w]; print("$tag $DT [$flags] %-16s %.3f ms  %.2f M solves/s  hbm %.3f  phases %s err %.1e" % (w, l["ms"], l["solves_per_s"]/1e6, l["hbm_frac_algorithmic"], {k: round(x,2) for k,x in l["phase_ms"].items()}, l["max_rel_err_vs_fp64_oracle"]))
except Exception as e:
    print("$tag failed", e, open("gpurun_out/tvvar/${tag}_$DT.err").read()[-800:])
PY
done
