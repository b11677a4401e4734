#!/bin/bash
# Every committed record of the round-5 build in ONE GPU-box call:  bash scripts/r05_records.sh
#   1. PMC passes of every bench leg (scripts/pmc_legs.sh: one counter per pass) -> gpurun_out/r05_pmc.json, installed as
#      profiles/r05_pmc.json on the box so that step 2 reports `roofline.traffic` / VALU fractions / sustained clock from THIS build
#   2. the driver's bench line (python bench.py) -> gpurun_out/r05_bench_f32.json
#   3. rocprofv3 --kernel-trace --stats of the headline -> gpurun_out/r05_kernel_stats.csv; of the value+gradient legs ->
#      gpurun_out/r05_vg_kernel_stats_*.csv
#   4. all five BASELINE configs -> gpurun_out/r05_bench_configs.jsonl
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
bash scripts/pmc_legs.sh r05 > gpurun_out/r05_pmc_legs.log 2>&1
tail -3 gpurun_out/r05_pmc_legs.log
cp gpurun_out/r05_pmc.json profiles/r05_pmc.json
python3 bench.py > gpurun_out/r05_bench_f32.json 2> gpurun_out/r05_bench_f32.err
tail -c 600 gpurun_out/r05_bench_f32.json; echo
rm -rf gpurun_out/prof_r05_kt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r05_kt -o p -- python3 bench.py --no-extra --no-cpu-baseline --steps 50 --warmup 10 > gpurun_out/r05_bench_under_rocprof.json 2> /dev/null
cp $(find gpurun_out/prof_r05_kt -name '*kernel_stats.csv' | head -1) gpurun_out/r05_kernel_stats.csv
for leg in value_and_grad_headline value_and_grad_config3; do
  rm -rf gpurun_out/prof_r05_$leg
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r05_$leg -o p -- python3 bench.py --only $leg > gpurun_out/r05_$leg.json 2> /dev/null
  cp $(find gpurun_out/prof_r05_$leg -name '*kernel_stats.csv' | head -1) gpurun_out/r05_vg_kernel_stats_$leg.csv
done
python3 bench_configs.py > gpurun_out/r05_bench_configs.jsonl 2> gpurun_out/r05_bench_configs.err
find gpurun_out/prof_r05_kt gpurun_out/prof_r05_value_and_grad_headline gpurun_out/prof_r05_value_and_grad_config3 -type f ! -name "*stats*.csv" -delete 2>/dev/null
head -c 400 gpurun_out/r05_kernel_stats.csv
