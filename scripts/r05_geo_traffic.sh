#!/bin/bash
# HBM traffic (FETCH_SIZE) and time of config 3's per-trial sweep for the workgroup geometries of lqg_tuning.trial_lds
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for v in 0 5 4 2 1; do
  d=gpurun_out/geo_$v
  rm -rf $d
  LQG_TRIAL_LDS=$v rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $d -o p -- python3 bench.py --only config3 > $d.json 2>/dev/null
  python3 - $d $v <<'PY'
import csv, glob, sys, json
rows = []
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    rows += [r for r in csv.DictReader(open(f)) if "k_trial" in r["Kernel_Name"]]
vals = [float(r["Counter_Value"]) for r in rows]
big = [v for v in vals if v > 0.5 * max(vals)]
d = json.load(open(sys.argv[1] + ".json"))["config3"]
print("TRIAL_LDS=%s  fetch GB per launch %.2f (2 x FETCH_SIZE KiB)  trial ms (under PMC) %.3f" % (sys.argv[2], 2 * sum(big) / len(big) * 1024 / 1e9, d["phase_ms"]["trial"]))
PY
done
