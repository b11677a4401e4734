#!/bin/bash
# rocprofv3 passes over bench_m2.py (mode M2 of SURVEY.md 8d: time-varying specs in, everything materialised out)
#   bash scripts/profile_m2.sh <tag> <f32|f64> <log2_batch>
cd "$(dirname "$0")/.."
TAG=$1; DT=$2; LB=$3
export TMPDIR=/tmp
python3 bench_m2.py --dtype $DT --log2-batch $LB --reps 5 > gpurun_out/m2_${TAG}.json 2> gpurun_out/m2_${TAG}.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_kt -o p -- python3 bench_m2.py --dtype $DT --log2-batch $LB --reps 5 > gpurun_out/prof_${TAG}_kt.json 2> gpurun_out/prof_${TAG}_kt.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/prof_${TAG}_$c -o p -- python3 bench_m2.py --dtype $DT --log2-batch $LB --reps 2 > gpurun_out/prof_${TAG}_$c.json 2> gpurun_out/prof_${TAG}_$c.err
done
find gpurun_out/prof_${TAG}_* -type f ! -name "*stats*.csv" ! -name "*counter_collection.csv" ! -name "*.json" ! -name "*.err" -delete
cat gpurun_out/m2_${TAG}.json | cut -c1-700
