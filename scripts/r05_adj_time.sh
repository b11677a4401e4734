#!/bin/bash
# value + gradient on the review's two shapes: round-1 lane kernels (reduced sizes: they keep 22 reals per lane-step) vs the split sweep
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r05_adjoint_time.jsonl
: > $O
python scripts/adjoint_baseline.py --sp 0 --log2-batch 16 --cands 4096 --trials 16 >> $O 2>> gpurun_out/r05_adjoint_time.err
python scripts/adjoint_baseline.py --sp 1 --log2-batch 16 --cands 4096 --trials 16 >> $O 2>> gpurun_out/r05_adjoint_time.err
python scripts/adjoint_baseline.py --sp 1 --log2-batch 18 --cands 4096 --trials 1024 >> $O 2>> gpurun_out/r05_adjoint_time.err
cat $O; tail -5 gpurun_out/r05_adjoint_time.err
