#!/bin/bash
# correctness of the split sweep, then per-kernel times (rocprofv3 --kernel-trace --stats) of value + gradient on the two shapes
# usage: r05_adj_prof.sh [tag [LQG_PAT_DIR]]
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
TAG=${1:-default}
if [ -n "$2" ]; then export LQG_PAT_DIR=$2; fi
mkdir -p gpurun_out
for shape in headline config3; do
  rm -rf gpurun_out/adjprof_${TAG}_$shape
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/adjprof_${TAG}_$shape -o p -- python3 scripts/adjoint_baseline.py --sp 1 --shapes $shape --dtypes ${DTYPES:-f32} --log2-batch 18 --cands 4096 --trials 1024 --steps 5 --warmup 2 > gpurun_out/adjprof_${TAG}_$shape.log 2>&1
  grep '^{' gpurun_out/adjprof_${TAG}_$shape.log
  f=$(find gpurun_out/adjprof_${TAG}_$shape -name '*kernel_stats.csv' | head -1)
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if "lqg" in r["Name"]]
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:9]:
    print("%7.3f ms avg x %4d calls  %s" % (float(r["AverageNs"]) * 1e-6, int(r["Calls"]), r["Name"][:100]))
PY
done
