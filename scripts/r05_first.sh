#!/bin/bash
# round 5, first GPU call: (1) one RCCL-backed bench line (torchrun, 1 rank), (2) SQ / GRBM counters of the headline kernels,
# (3) the round-4 adjoint on the headline and config-3 shapes (the baseline the rebuilt sweep is measured against)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --no-extra --no-cpu-baseline --steps 20 --warmup 3 > gpurun_out/r05_rccl_n1.json 2> gpurun_out/r05_rccl_n1.err
tail -c 600 gpurun_out/r05_rccl_n1.json
python scripts/adjoint_baseline.py --log2-batch 16 > gpurun_out/r05_adjoint_baseline.jsonl 2> gpurun_out/r05_adjoint_baseline.err
cat gpurun_out/r05_adjoint_baseline.jsonl
tail -3 gpurun_out/r05_adjoint_baseline.err
bash scripts/pmc_headline_sq.sh gpurun_out/r05_headline_sq.txt
