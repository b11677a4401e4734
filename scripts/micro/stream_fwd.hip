// stream_fwd.hip — would an LDS-DMA prefetch of HALF the per-step inputs pay for the M2 forward kernel?  (round-3 experiment)
// The forward MAT kernel (k_forward<TI = false, FUSED, MAT>) holds one wave per SIMD (409 VGPRs): per wave-step it loads 212
// rows of 256 B, runs ~3700 dependent-ish instructions, stores 134 rows — strictly one after the other.  This replays that shape:
//   mode 0: load EI rows -> wait -> CH-long arithmetic -> store EO rows                               (what the kernel does)
//   mode 1: ES of the EI rows of step t + 1 are requested by global_load_lds (no VGPR destination) right after the direct rows
//           of step t are in, so that they land in LDS while step t computes; step t + 1 reads them with ds_read
// One wave per SIMD is enforced with LDS (40 KB per 64-lane workgroup).   usage: stream_fwd <log2B> <T> <mode> <CH>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr int EI = 212, EO = 134, ES = 104;        // ES rows travel through LDS in mode 1

template <int MODE>
__global__ void __launch_bounds__(64) k_fwd(const float* __restrict__ in, float* __restrict__ out, long B, int T, int ch) {
  extern __shared__ float lds[];                   // [ES][64] + padding up to 40 KB
  const long s = blockIdx.x * 64L + threadIdx.x;
  const int lane = threadIdx.x;
  auto iaddr = [&](int t, int e) -> long { return ((long)t * EI + e) * B + s; };
  auto oaddr = [&](int t, int e) -> long { return ((long)t * EO + e) * B + s; };
  float acc0 = 0.f, acc1 = 1.f, acc2 = 2.f, acc3 = 3.f;
  if (MODE == 1) {
#pragma unroll
    for (int e = 0; e < ES; ++e)
      __builtin_amdgcn_global_load_lds(in + iaddr(0, e), (__attribute__((address_space(3))) void*)(lds + e * 64), 4, 0, 0);
  }
  for (int t = 0; t < T; ++t) {
    float v[EI];
    if (MODE == 1) {
      __builtin_amdgcn_s_waitcnt(0x0f70);             // vmcnt(0): the DMA of this step's LDS rows has landed
#pragma unroll
      for (int e = 0; e < ES; ++e) v[e] = lds[e * 64 + lane];
#pragma unroll
      for (int e = ES; e < EI; ++e) v[e] = in[iaddr(t, e)];
    } else {
#pragma unroll
      for (int e = 0; e < EI; ++e) v[e] = in[iaddr(t, e)];
    }
    float sum = 0.f;
#pragma unroll
    for (int e = 0; e < EI; ++e) sum += v[e];        // every input consumed (direct rows waited for here)
    if (MODE == 1 && t + 1 < T) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = 0; e < ES; ++e)
        __builtin_amdgcn_global_load_lds(in + iaddr(t + 1, e), (__attribute__((address_space(3))) void*)(lds + e * 64), 4, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    for (int i = 0; i < ch; ++i) {                  // the step's arithmetic: four chains of dependent multiply-adds
      acc0 = fmaf(acc0, 0.999f, sum);
      acc1 = fmaf(acc1, 0.998f, acc0);
      acc2 = fmaf(acc2, 0.997f, acc1);
      acc3 = fmaf(acc3, 0.996f, acc2);
    }
#pragma unroll
    for (int e = 0; e < EO; ++e) out[oaddr(t, e)] = acc3 + (float)e;
  }
}

int main(int argc, char** argv) {
  const int log2B = atoi(argv[1]), T = atoi(argv[2]), mode = atoi(argv[3]), ch = atoi(argv[4]);
  const long B = 1L << log2B;
  float *in, *out;
  (void)hipMalloc(&in, sizeof(float) * B * T * EI);
  (void)hipMalloc(&out, sizeof(float) * B * T * EO);
  (void)hipMemset(in, 0, sizeof(float) * B * T * EI);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  float best = 1e30f;
  for (int r = 0; r < 4; ++r) {
    (void)hipEventRecord(e0);
    if (mode == 0) hipLaunchKernelGGL(k_fwd<0>, dim3((unsigned)(B / 64)), dim3(64), 40000, 0, in, out, B, T, ch);
    else hipLaunchKernelGGL(k_fwd<1>, dim3((unsigned)(B / 64)), dim3(64), 40000, 0, in, out, B, T, ch);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (r && ms < best) best = ms;
  }
  const double bytes = 4.0 * B * T * (EI + EO);
  printf("{\"log2B\": %d, \"T\": %d, \"mode\": %d, \"chain\": %d, \"ms\": %.3f, \"GBps\": %.1f, \"us_per_wave_step\": %.2f}\n", log2B, T, mode, ch,
         best, bytes / best / 1e6, best * 1e3 / T / ((double)B / 64 / 1024));
  return 0;
}
