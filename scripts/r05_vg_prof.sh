#!/bin/bash
# rocprofv3 --kernel-trace --stats of the two value+gradient bench legs
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
for leg in value_and_grad_headline value_and_grad_config3; do
  rm -rf gpurun_out/vgprof_$leg
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/vgprof_$leg -o p -- python3 bench.py --only $leg > gpurun_out/vgprof_$leg.json 2> gpurun_out/vgprof_$leg.err
  python3 - gpurun_out/vgprof_$leg.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    print(k, {q: v[q] for q in ("value", "ms_per_value_and_grad", "kernel_ms", "parity") if q in v})
PY
  f=$(find gpurun_out/vgprof_$leg -name '*kernel_stats.csv' | head -1)
  python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "lqg" in r["Name"]]
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:8]:
    print("%7.3f ms avg x %4d calls  %s" % (float(r["AverageNs"]) * 1e-6, int(r["Calls"]), r["Name"][:100]))
PY
done
