// lqg_rng.hpp — counter-based normal draws for System.simulate (lqg/system.py:100-105: the reference draws
// eps ~ N(0, I_x), eta ~ N(0, I_y) per (trial, step) from jax.random inside its per-trial scan).
//
// Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11): a keyed bijection of a 128-bit
// counter; no state, no sequence — a draw is a pure function of (seed, system, trial, step, block), so the trajectory of
// trial k of system s does not depend on the NUMBER of systems or trials in the call, nor on how the batch is mapped to
// lanes, kernels or GPUs (the lane kernels and the run-time-dims kernel of lqg_coop.hpp produce the same numbers), and
// nothing is materialised in HBM.  (Rounds 2-3 keyed the counter on pair = system * n_trials + trial: the draws of every
// system but the first changed with n_trials.)
//   counter = (trial, system, step, block),  key = (seed low, seed high);  trial and system indices below 2^32
//   blocks 0, 1, ... of a step feed the process noise eps (4 normals per block), blocks kEtaBlock, ... the observation noise
// Normals: Box-Muller on 32-bit uniforms (k + 1/2) 2^-32 in (0, 1), evaluated in fp32 (the reference's default precision)
// and widened for an fp64 problem: the draws of an fp64 simulation are fp32 normals (24-bit mantissa, |z| <= 6.66).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace lqg {
namespace rng {

constexpr uint32_t kEtaBlock = 0x10000u;

__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
    const uint32_t hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
    const uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
    c[0] = n0;
    c[1] = lo1;
    c[2] = n2;
    c[3] = lo0;
    k0 += W0;
    k1 += W1;
  }
}

// four standard normals of (seed, system, trial, step, block)
__device__ __forceinline__ void normal4(unsigned long long seed, long system, long trial, uint32_t step, uint32_t block,
                                        float (&z)[4]) {
  uint32_t c[4] = {(uint32_t)trial, (uint32_t)system, step, block};
  philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  constexpr float kInv32 = 2.3283064365386963e-10f;        // 2^-32
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const float u1 = ((float)c[2 * h] + 0.5f) * kInv32;     // (0, 1]: the float rounding of 2^32 - 1/2 is 2^32
    const float u2 = ((float)c[2 * h + 1] + 0.5f) * kInv32;
    const float rad = sqrtf(-2.0f * logf(fminf(u1, 0.99999994f)));
    float sn, cs;
    sincospif(2.0f * u2, &sn, &cs);
    z[2 * h] = rad * cs;
    z[2 * h + 1] = rad * sn;
  }
}

}  // namespace rng
}  // namespace lqg
