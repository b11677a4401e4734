#!/bin/bash
# value+gradient at the headline shape under build variants of the reverse system sweep (compiled on the box into their own
# pattern directories):  bash scripts/r06_rev_variants.sh "tag:flags" "tag:flags" ...
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/revvar
for spec in "$@"; do
  tag=${spec%%:*}; flags=${spec#*:}
  export LQG_PAT_DIR=$PWD/gpurun_out/revvar/pat_$tag
  mkdir -p $LQG_PAT_DIR
  cp -n lqg_amd/csrc/pat/pat_*.so lqg_amd/csrc/pat/pat_*.stamp $LQG_PAT_DIR/ 2>/dev/null
  export LQG_ADJ_FLAGS="$flags"
  python bench.py --only ${LEG:-value_and_grad_headline} > gpurun_out/revvar/$tag.json 2> gpurun_out/revvar/$tag.err
  python - <<PY
import json
try:
    v=json.load(open("gpurun_out/revvar/$tag.json"))[list(json.load(open("gpurun_out/revvar/$tag.json")))[0]]
    print("$tag [$flags]  %.3f ms " % v["ms_per_value_and_grad"], {k: round(x,3) for k,x in v["kernel_ms"].items()})
except Exception as e:
    print("$tag failed", e, open("gpurun_out/revvar/$tag.err").read()[-800:])
PY
  so=$(ls $LQG_PAT_DIR/padj_2616ffee38c935eb_1a3.so 2>/dev/null)
  [ -n "$so" ] && python scripts/kernel_resources.py $so 'asp_sys_rev.*1a3, 2' | cut -c1-60
done
