#!/bin/bash
# SQ wait / issue counters of the per-trial sweep kernels of one BASELINE config:  bash scripts/pmc_cfg.sh <config>
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
CFG=$1
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_SMEM SQ_INSTS_VMEM_RD"; do
  tag=$(echo $grp | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/pmcc${CFG}_$tag -o p -- python3 bench_configs.py --configs $CFG --reps 2 > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
dur = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob("gpurun_out/pmcc${CFG}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_trial" in k or "k_forward" in k:
            a = acc[k[:40]][r["Counter_Name"]]
            a[0] += 1; a[1] += float(r["Counter_Value"])
for f in glob.glob("gpurun_out/pmcc${CFG}_SQ_WAVE*/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_trial" in k or "k_forward" in k:
            d = dur[k[:40]]; d[0] += 1; d[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for k, d in acc.items():
    n, t = dur[k]
    print(k, "avg us %.1f" % (t / max(n, 1)))
    for c, (m, v) in sorted(d.items()):
        print("   %-20s %.4e per call" % (c, v / m))
    if "SQ_WAIT_ANY" in d and "SQ_WAVE_CYCLES" in d:
        print("   wait fraction %.2f" % (d["SQ_WAIT_ANY"][1] / d["SQ_WAIT_ANY"][0] / (d["SQ_WAVE_CYCLES"][1] / d["SQ_WAVE_CYCLES"][0])))
PY
