#!/bin/bash
# rocprofv3 passes over `python bench.py` (headline workload) on the GPU box: kernel trace + stats, then one PMC counter
# per pass (FETCH_SIZE and WRITE_SIZE do not fit one pass; never combined with a trace domain other than kernel-trace).
#   bash scripts/profile_headline.sh <tag> [f32|f64] [extra bench args]
# Outputs under gpurun_out/prof_<tag>_*; (rounds 1-3; round 4 collects per-leg records with scripts/pmc_legs.sh + scripts/pmc_records.py)
cd "$(dirname "$0")/.."
TAG=$1; DT=${2:-f32}; shift; shift
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
ARGS="bench.py --dtype $DT --no-extra --no-cpu-baseline $@"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_kt -o p -- python3 $ARGS --steps 10 --warmup 2 > gpurun_out/prof_${TAG}_kt.json 2> gpurun_out/prof_${TAG}_kt.err
for c in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/prof_${TAG}_$c -o p -- python3 $ARGS --steps 3 --warmup 1 > gpurun_out/prof_${TAG}_$c.json 2> gpurun_out/prof_${TAG}_$c.err
done
find gpurun_out/prof_${TAG}_* -name "*.csv" | head -20
# keep the merge small: drop everything but the stats + counter tables
find gpurun_out/prof_${TAG}_* -type f ! -name "*stats*.csv" ! -name "*counter_collection.csv" ! -name "*.json" ! -name "*.err" -delete
