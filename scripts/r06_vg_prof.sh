#!/bin/bash
# per-kernel times (rocprofv3 --kernel-trace --stats) of one bench leg:  bash scripts/r06_vg_prof.sh <leg> [tag]
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
LEG=${1:-value_and_grad_headline}
TAG=${2:-r06}
rm -rf gpurun_out/prof_${TAG}_$LEG
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_$LEG -o p -- python3 bench.py --only $LEG > gpurun_out/prof_${TAG}_$LEG.log 2>&1
f=$(find gpurun_out/prof_${TAG}_$LEG -name '*kernel_stats.csv' | head -1)
cp "$f" gpurun_out/${TAG}_${LEG}_kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if "lqg" in r["Name"]]
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:12]:
    print("%7.3f ms avg x %4d calls  %s" % (float(r["AverageNs"]) * 1e-6, int(r["Calls"]), r["Name"][:110]))
PY
