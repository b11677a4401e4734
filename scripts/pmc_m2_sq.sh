#!/bin/bash
# SQ wait / issue counters of the M2 kernels (python bench_m2.py, fp32, 2^17 systems)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR"; do
  tag=$(echo $grp | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/pmcm2_$tag -o p -- python3 bench_m2.py --log2-batch 17 --reps 2 > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob("gpurun_out/pmcm2_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "_tv_sp" in k:
            a = acc[k[:32]][r["Counter_Name"]]
            a[0] += 1; a[1] += float(r["Counter_Value"])
for k, d in acc.items():
    print(k)
    for c, (m, v) in sorted(d.items()):
        print("   %-26s %.4e per launch" % (c, v / m))
    w = d["SQ_WAVE_CYCLES"][1] / d["SQ_WAVE_CYCLES"][0]
    g = lambda n: d[n][1] / max(d[n][0], 1)
    print("   of wave cycles: waiting on anything %.2f, on a dependent instruction %.2f, VALU issuing %.2f" % (g("SQ_WAIT_ANY") / w, g("SQ_WAIT_INST_ANY") / w, g("SQ_ACTIVE_INST_VALU") / w))
PY
rm -rf gpurun_out/pmcm2_*
