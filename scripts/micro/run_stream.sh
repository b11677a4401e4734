#!/bin/bash
# scripts/micro/run_stream.sh — sweep of the M2 access-pattern microbenchmark (on the GPU box)
cd "$(dirname "$0")"
for lay in 0 1; do for blk in 64 256; do for pf in 0 1; do
  ./stream_soa 17 200 236 150 $lay $blk 0 $pf
done; done; done
# occupancy limit: 40 KB of LDS per 64-thread block -> 4 blocks (1 wave/SIMD) per CU; 20 KB -> 8 (2 waves/SIMD)
for lds in 40000 20000 10000; do ./stream_soa 17 200 236 150 0 64 $lds 0; ./stream_soa 17 200 236 150 1 64 $lds 0; done
for lay in 0 1; do ./stream_soa 17 500 88 28 $lay 64 0 0; ./stream_soa 17 500 64 64 $lay 64 0 0; ./stream_soa 20 500 16 16 $lay 64 0 0; done
