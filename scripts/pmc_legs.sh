#!/bin/bash
# Per-leg PMC passes of the bench (run on the GPU box):  bash scripts/pmc_legs.sh <tag> [leg ...]
# For every leg: `python3 bench.py --only <leg>` under rocprofv3 with ONE counter per pass (FETCH_SIZE and WRITE_SIZE do not fit
# one pass; --pmc is only ever combined with --kernel-trace), plus one --kernel-trace --stats pass of the default headline.
# scripts/pmc_records.py <tag> turns the tables into gpurun_out/<tag>_pmc.json (install as profiles/r06_pmc.json).
cd "$(dirname "$0")/.."
TAG=$1; shift
LEGS=${@:-"headline_f32 headline_f64 m2_f32 timevarying_f64 config3 config5_one_system config4_sharded dense_generic_f32 dense_generic_f64 specialised_joint_n6 value_and_grad_headline value_and_grad_config3"}
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
mkdir -p gpurun_out
for leg in $LEGS; do
  for c in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU GRBM_GUI_ACTIVE; do
    d=gpurun_out/pmc_${TAG}_${leg}_$c
    rm -rf $d
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -o p -- python3 bench.py --only $leg --steps 3 --warmup 1 --no-cpu-baseline > $d.json 2> $d.err
    # keep the merge small: only the counter and kernel-trace tables travel back
    find $d -type f ! -name "*counter_collection.csv" ! -name "*kernel_trace.csv" -delete 2>/dev/null
  done
done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_kt -o p -- python3 bench.py --no-extra --no-cpu-baseline --steps 50 --warmup 10 > gpurun_out/prof_${TAG}_kt.json 2> gpurun_out/prof_${TAG}_kt.err
find gpurun_out/prof_${TAG}_kt -type f ! -name "*stats*.csv" -delete 2>/dev/null
python3 scripts/pmc_records.py $TAG $LEGS
