#!/bin/bash
# LDS / wait counters of the CUT reverse system sweep (-DLQG_ASP_SPLIT_KAL=1, compiled on the box into its own pattern directory)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
export LQG_PAT_DIR=$PWD/gpurun_out/revvar/pat_splitpmc
mkdir -p $LQG_PAT_DIR
cp -n lqg_amd/csrc/pat/pat_*.so lqg_amd/csrc/pat/pat_*.stamp $LQG_PAT_DIR/ 2>/dev/null
export LQG_ADJ_FLAGS="-DLQG_ASP_SPLIT_KAL=1 $1"
python bench.py --only value_and_grad_headline > gpurun_out/revvar/splitpmc.json 2> gpurun_out/revvar/splitpmc.err   # compile + warm
for grp in "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY" "SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"; do
  tag=$(echo $grp | tr ' ' '_')
  rm -rf gpurun_out/pmcs_$tag
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/pmcs_$tag -o p -- python3 bench.py --only value_and_grad_headline > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob("gpurun_out/pmcs_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_asp_sys_rev<float" in k or "k_asp_kal_rev<float" in k or "k_asp_sys_rev_fused<float" in k:
            a = acc[k[:60]][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k, d in acc.items():
    print(k)
    for c, (m, v) in sorted(d.items()):
        print("   %-26s %.4e per call (%d)" % (c, v / m, m))
PY
