#!/bin/bash
# SQ wait / issue counters of the headline kernels (python bench.py, fp32)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_INST_CYCLES_VMEM_RD SQ_WAVES"; do
  tag=$(echo $grp | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/pmch_$tag -o p -- python3 bench.py --no-extra --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob("gpurun_out/pmch_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_forward_sp" in k or "k_riccati_sp" in k:
            a = acc[k[:28]][r["Counter_Name"]]
            a[0] += 1; a[1] += float(r["Counter_Value"])
for k, d in acc.items():
    print(k)
    for c, (m, v) in sorted(d.items()):
        print("   %-24s %.4e per call" % (c, v / m))
    w = d["SQ_WAVE_CYCLES"][1] / d["SQ_WAVE_CYCLES"][0]
    print("   wait_any/wave_cycles %.2f   wait_inst/wave_cycles %.2f" % (d["SQ_WAIT_ANY"][1] / d["SQ_WAIT_ANY"][0] / w, d["SQ_WAIT_INST_ANY"][1] / d["SQ_WAIT_INST_ANY"][0] / w))
PY
