#!/bin/bash
# SQ counters of the time-chunked per-trial sweep on config 2 (one counter group per pass)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VALU" "SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU"; do
  tag=$(echo $grp | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/pmct_$tag -o p -- python3 bench_configs.py --configs 2 --reps 3 > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob("gpurun_out/pmct_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_trial_ll" in k or "k_trial_zs" in k:
            a = acc[k[:24]][r["Counter_Name"]]
            a[0] += 1; a[1] += float(r["Counter_Value"])
for k, d in acc.items():
    print(k)
    for c, (n, v) in sorted(d.items()):
        print("   %-24s %.4e per call (%d calls)" % (c, v / n, n))
PY
