// stream_soa.hip — what HBM rate does the M2 ACCESS PATTERN admit, whatever the arithmetic?  (round-3 experiment)
// One system per lane; per step every lane reads EI reals and writes EO reals.  Layouts:
//   soa  : [T][element][B]            (system fastest: a wave touches 256 contiguous bytes per (t, element); the rows of one
//                                      step are B*4 bytes apart — what bench_m2.py hands to lqg_solve_materialised)
//   tile : [B/64][T][element][64]     (a wave streams ONE contiguous run for the whole sweep)
// usage: stream_soa <log2B> <T> <EI> <EO> <layout 0|1> <block 64|256> <waves-per-simd-limit via LDS> <prefetch 0|1> [row padding in reals]
//   (row padding: the rows of the soa layout are B + pad reals apart -- does a stride that is not a power of two admit more?)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int EI, int EO, bool TILE, bool PF>
__global__ void k_stream(const float* __restrict__ in, float* __restrict__ out, long B, int T, int lds_pad, long ld) {
  extern __shared__ float pad[];
  const long s = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (s >= B) return;
  if (lds_pad < 0) pad[threadIdx.x] = 0.f;
  const long tile = s >> 6, lane = s & 63;
  auto iaddr = [&](int t, int e) -> long { return TILE ? ((tile * T + t) * EI + e) * 64 + lane : ((long)t * EI + e) * ld + s; };
  auto oaddr = [&](int t, int e) -> long { return TILE ? ((tile * T + t) * EO + e) * 64 + lane : ((long)t * EO + e) * ld + s; };
  float acc = 0.f;
  float nx[EI];
  if (PF) {
#pragma unroll
    for (int e = 0; e < EI; ++e) nx[e] = in[iaddr(0, e)];
  }
  for (int t = 0; t < T; ++t) {
    float v[EI];
    if (PF) {
#pragma unroll
      for (int e = 0; e < EI; ++e) v[e] = nx[e];
      const int tn = t + 1 < T ? t + 1 : t;
#pragma unroll
      for (int e = 0; e < EI; ++e) nx[e] = in[iaddr(tn, e)];
    } else {
#pragma unroll
      for (int e = 0; e < EI; ++e) {
#if defined(NT_LOAD)
        v[e] = __builtin_nontemporal_load(&in[iaddr(t, e)]);
#else
        v[e] = in[iaddr(t, e)];
#endif
      }
    }
    // a dependent chain over the step's inputs (stands for the recursion: outputs need every input)
#pragma unroll
    for (int e = 0; e < EI; ++e) acc = acc * 0.999f + v[e];
#pragma unroll
    for (int e = 0; e < EO; ++e) {
#if defined(NT_STORE)
      __builtin_nontemporal_store(acc + (float)e, &out[oaddr(t, e)]);     // -DNT_STORE: do streaming hints raise the ceiling?
#else
      out[oaddr(t, e)] = acc + (float)e;
#endif
    }
  }
}

template <int EI, int EO>
void run(int log2B, int T, int layout, int block, int lds_bytes, int pf, int pad) {
  const long B = 1L << log2B, ld = B + pad;
  float *in, *out;
  hipMalloc(&in, sizeof(float) * ld * T * EI);
  hipMalloc(&out, sizeof(float) * ld * T * EO);
  hipMemset(in, 0, sizeof(float) * ld * T * EI);
  dim3 grid((unsigned)(B / block)), blk(block);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e30f;
  for (int r = 0; r < 4; ++r) {
    hipEventRecord(e0);
    if (layout == 0 && !pf) hipLaunchKernelGGL((k_stream<EI, EO, false, false>), grid, blk, lds_bytes, 0, in, out, B, T, 0, ld);
    if (layout == 1 && !pf) hipLaunchKernelGGL((k_stream<EI, EO, true, false>), grid, blk, lds_bytes, 0, in, out, B, T, 0, ld);
    if (layout == 0 && pf) hipLaunchKernelGGL((k_stream<EI, EO, false, true>), grid, blk, lds_bytes, 0, in, out, B, T, 0, ld);
    if (layout == 1 && pf) hipLaunchKernelGGL((k_stream<EI, EO, true, true>), grid, blk, lds_bytes, 0, in, out, B, T, 0, ld);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (r && ms < best) best = ms;
  }
  const double bytes = 4.0 * B * T * (EI + EO);
  printf("{\"EI\": %d, \"EO\": %d, \"log2B\": %d, \"T\": %d, \"layout\": \"%s\", \"block\": %d, \"lds_bytes\": %d, \"prefetch\": %d, \"row_pad\": %d, \"ms\": %.3f, \"GBps\": %.1f}\n",
         EI, EO, log2B, T, layout ? "tile64" : "soa", block, lds_bytes, pf, pad, best, bytes / best / 1e6);
  hipFree(in);
  hipFree(out);
}

int main(int argc, char** argv) {
  const int log2B = atoi(argv[1]), T = atoi(argv[2]), EI = atoi(argv[3]), EO = atoi(argv[4]), layout = atoi(argv[5]),
            block = atoi(argv[6]), lds = atoi(argv[7]), pf = atoi(argv[8]), pad = argc > 9 ? atoi(argv[9]) : 0;
  if (EI == 236 && EO == 150) run<236, 150>(log2B, T, layout, block, lds, pf, pad);
  else if (EI == 60 && EO == 134) run<60, 134>(log2B, T, layout, block, lds, pf, pad);   // the pattern forward kernel of M2
  else if (EI == 88 && EO == 28) run<88, 28>(log2B, T, layout, block, lds, pf, pad);
  else if (EI == 64 && EO == 64) run<64, 64>(log2B, T, layout, block, lds, pf, pad);
  else if (EI == 16 && EO == 16) run<16, 16>(log2B, T, layout, block, lds, pf, pad);
  else { printf("unsupported EI/EO\n"); return 1; }
  return 0;
}
