// lqg_adjoint_trial_sp.hpp — the per-TRIAL half of the round-5 reverse-mode sweep (lqg_adjoint_sp.hpp), gfx950.
//
//   (forward)         k_trial_sp<..., CKT> of lqg_kernels_sp.hpp: mean recursion + log-density over the operator stream
//                     (lqg/system.py:219-221, 244-248), keeping c_{t-1} (TrialArgs::tck) for every CKT-th row
//   k_asp_trial_rev   the mu-bar recursion backward over the same stream: per chunk the steps' (w, c) are recomputed from the
//                     chunk's checkpoint into registers, then walked backward; per step the workgroup's trials are reduced to
//                     the TRIAL SUMS of asp::Sums (wave butterfly: log2 stages in which every lane hands half of its values to
//                     its partner, so 32 sums cost ~32 adds instead of 32 x 6; then one LDS exchange between the waves).
// One lane = TPL trials of one system; one workgroup = 256 lanes = up to 1024 trials of ONE system (blockIdx.y): the step's
// operator block is wave-uniform (scalar loads), read once per workgroup and pass.
#pragma once
#include "lqg_adjoint_sp.hpp"

#ifndef LQG_ASP_TRIAL_BLOCK
#define LQG_ASP_TRIAL_BLOCK 256
#endif
#ifndef LQG_ASP_TPL
#define LQG_ASP_TPL 4
#endif
#ifndef LQG_ASP_XPREFETCH
#define LQG_ASP_XPREFETCH 1       // data rows requested one step ahead in the reverse sweep's two passes
#endif
#ifndef LQG_ASP_OPS_SCALAR
#define LQG_ASP_OPS_SCALAR 0      // 1: the reverse sweep reads the operator blocks through the scalar cache (as k_trial_sp) instead of LDS
#endif

namespace lqg {
namespace asp {

template <typename R>
struct TrialRevArgs {
  DTraj<R> x;
  const R* g;                // upstream weights, null = 1
  long g_sb, g_sn;
  R* ll;                     // value out (k_asp_trial_fwd), may be null
  long ll_sb, ll_sn;
  R* tck;                    // restart data c_{t-1} of the chunk rows [n_sys][nckt + 1][M - ND][npad] (TrialArgs::tck)
  long npad;
  int nckt;
  R* sums;                   // [parts = gridDim.x][n_sys][T][Sums::N]
  R* gsum;                   // [parts][n_sys]: sum of the upstream weights of the part's trials
  long n_sys, n_trials;
  int T;
};

// ---------------------------------------------------------------- wave butterfly
// V <= 32 values per lane -> lane l (< 32) holds the wave's total of value bitrev5(l).  Stage s pairs lane l with l ^ 2^s: each
// hands the partner the half of its values the partner keeps, so the stages cost 16 + 8 + 4 + 2 + 1 pair operations instead
// of 32 full six-stage reductions.
// Partner exchange without the LDS pipeline (round 5, after the PMC pass showed the sweep waiting, not issuing): xor 1 / 2 DPP quad
// permutes, xor 4 two banked DPP row shifts, xor 8 a DPP row rotate, xor 16 / 32 the gfx950 permlane swaps — every stage is
// VALU-only (ds_swizzle / ds_bpermute share lgkmcnt with the LDS reads of the operator blocks).
template <int S>
LQG_DEV int lane_xor_bits(int v) {
  static_assert(S >= 0 && S <= 3, "xor 1, 2, 4, 8 within a row of 16 lanes");
  if constexpr (S == 0) return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);          // quad_perm [1,0,3,2]
  else if constexpr (S == 1) return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);     // quad_perm [2,3,0,1]
  else if constexpr (S == 2) {
    int t = __builtin_amdgcn_update_dpp(v, v, 0x104, 0xf, 0x5, false);                             // row_shl:4 into banks 0, 2
    return __builtin_amdgcn_update_dpp(t, v, 0x114, 0xf, 0xa, false);                              // row_shr:4 into banks 1, 3
  } else return __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, false);                         // row_ror:8
}
template <int S>
LQG_DEV float lane_xor(float v) { return __int_as_float(lane_xor_bits<S>(__float_as_int(v))); }
template <int S>
LQG_DEV double lane_xor(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = lane_xor_bits<S>((int)(b & 0xffffffffll)), hi = lane_xor_bits<S>((int)(b >> 32));
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// lo + (partner's lo) in the lanes whose bit 4 (WIDE = 0) / bit 5 (WIDE = 1) is clear, hi + (partner's hi) in the others: one
// v_permlane16_swap / v_permlane32_swap (odd rows / the upper half of the first operand trade places with even rows / the lower
// half of the second) and one add — no selects
template <int WIDE>
LQG_DEV float swap_add(float lo, float hi) {
  if constexpr (WIDE == 0) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(lo), __float_as_uint(hi), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
  } else {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo), __float_as_uint(hi), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
}
template <int WIDE>
LQG_DEV double swap_add(double lo, double hi) {
  const unsigned long long bl = (unsigned long long)__double_as_longlong(lo), bh = (unsigned long long)__double_as_longlong(hi);
  unsigned a0, a1, b0, b1;
  if constexpr (WIDE == 0) {
    const auto r = __builtin_amdgcn_permlane16_swap((unsigned)bl, (unsigned)bh, false, false);
    const auto q = __builtin_amdgcn_permlane16_swap((unsigned)(bl >> 32), (unsigned)(bh >> 32), false, false);
    a0 = r[0]; b0 = r[1]; a1 = q[0]; b1 = q[1];
  } else {
    const auto r = __builtin_amdgcn_permlane32_swap((unsigned)bl, (unsigned)bh, false, false);
    const auto q = __builtin_amdgcn_permlane32_swap((unsigned)(bl >> 32), (unsigned)(bh >> 32), false, false);
    a0 = r[0]; b0 = r[1]; a1 = q[0]; b1 = q[1];
  }
  return __longlong_as_double((long long)(((unsigned long long)a1 << 32) | a0)) +
         __longlong_as_double((long long)(((unsigned long long)b1 << 32) | b0));
}

// stage S (partner = lane ^ 2^S) over the slots [0, 2 H): slot k + H is handed over / received.  Slots whose indices all lie
// beyond the V live values (BASE + k + H >= V) hold nothing: there both lanes just add the partner's slot k (what the `up` lanes
// then hold stands for an index >= V and is never read).
template <int S, int H, int BASE, int V, int VP, typename R>
LQG_DEV void butterfly_stage(R (&cur)[VP], const bool up) {
  LQG_UNROLL for (int k = 0; k < H; ++k) {
    if constexpr (S == 4) {
      cur[k] = (BASE + k + H >= V) ? swap_add<0>(cur[k], cur[k]) : swap_add<0>(cur[k], cur[k + H]);
    } else if (BASE + k + H >= V) {
      cur[k] = cur[k] + lane_xor<S>(cur[k]);
    } else {
      const R lo = cur[k], hi = cur[k + H];
      const R keepv = up ? hi : lo;
      const R send = up ? lo : hi;
      cur[k] = keepv + lane_xor<S>(send);
    }
  }
}
// total over the partner pair in BOTH lanes (the stages past the last halving one)
template <int S, typename R>
LQG_DEV R lane_sum(R v) {
  if constexpr (S == 4) return swap_add<0>(v, v);
  else if constexpr (S == 5) return swap_add<1>(v, v);
  else return v + lane_xor<S>(v);
}

constexpr int pow2_at_least(int v) { return v <= 8 ? 8 : (v <= 16 ? 16 : 32); }
// index whose total lane l holds after the VP-slot butterfly: the low log2(VP) lane bits, reversed
template <int VP>
LQG_DEV int butterfly_index(int l) {
  if constexpr (VP == 8) return ((l & 1) << 2) | (l & 2) | ((l & 4) >> 2);
  else if constexpr (VP == 16) return ((l & 1) << 3) | ((l & 2) << 1) | ((l & 4) >> 1) | ((l & 8) >> 3);
  else return ((l & 1) << 4) | ((l & 2) << 2) | (l & 4) | ((l & 8) >> 2) | ((l & 16) >> 4);
}

// values v[BASE .. BASE + VP) of every lane -> lane l holds the wave's total of value BASE + butterfly_index<VP>(l)
template <typename R, int V, int BASE, int VP>
LQG_DEV R wave_transpose_reduce(const R (&v)[V]) {
  const int lane = threadIdx.x & 63;
  R cur[VP];
  LQG_UNROLL for (int i = 0; i < VP; ++i) cur[i] = (BASE + i < V) ? v[BASE + i < V ? BASE + i : 0] : R(0);
  if constexpr (VP == 32) {
    butterfly_stage<0, 16, BASE, V>(cur, (lane & 1) != 0);
    butterfly_stage<1, 8, BASE, V>(cur, (lane & 2) != 0);
    butterfly_stage<2, 4, BASE, V>(cur, (lane & 4) != 0);
    butterfly_stage<3, 2, BASE, V>(cur, (lane & 8) != 0);
    butterfly_stage<4, 1, BASE, V>(cur, (lane & 16) != 0);
    return lane_sum<5>(cur[0]);
  } else if constexpr (VP == 16) {
    butterfly_stage<0, 8, BASE, V>(cur, (lane & 1) != 0);
    butterfly_stage<1, 4, BASE, V>(cur, (lane & 2) != 0);
    butterfly_stage<2, 2, BASE, V>(cur, (lane & 4) != 0);
    butterfly_stage<3, 1, BASE, V>(cur, (lane & 8) != 0);
    return lane_sum<5>(lane_sum<4>(cur[0]));
  } else {
    butterfly_stage<0, 4, BASE, V>(cur, (lane & 1) != 0);
    butterfly_stage<1, 2, BASE, V>(cur, (lane & 2) != 0);
    butterfly_stage<2, 1, BASE, V>(cur, (lane & 4) != 0);
    return lane_sum<5>(lane_sum<4>(lane_sum<3>(cur[0])));
  }
}

// reduce V per-lane values over the workgroup and store them contiguously at out[0 .. V); `lds` holds 2 x NW x VPT reals,
// `parity` alternates per call so that ONE barrier per call suffices.  Groups of 32 values, the last one of 8 / 16 / 32.
template <int V>
struct ReduceShape {
  static constexpr int NG = (V + 31) / 32;
  static constexpr int LAST = pow2_at_least(V - (NG - 1) * 32);
  static constexpr int VPT = (NG - 1) * 32 + LAST;
};
template <int GQ, typename R, int V>
LQG_DEV void block_reduce_groups(const R (&v)[V], R* buf, int lane, int wave) {
  using RS = ReduceShape<V>;
  if constexpr (GQ < RS::NG) {
    constexpr int VP = (GQ == RS::NG - 1) ? RS::LAST : 32;
    const R tot = wave_transpose_reduce<R, V, GQ * 32, VP>(v);
    if (lane < VP) buf[wave * RS::VPT + GQ * 32 + butterfly_index<VP>(lane)] = tot;
    block_reduce_groups<GQ + 1>(v, buf, lane, wave);
  }
}
// the wave's totals of V per-lane values, deposited at buf[wave][0 .. V) (natural order)
template <typename R, int V>
LQG_DEV void wave_reduce_deposit(const R (&v)[V], R* buf) {
  using RS = ReduceShape<V>;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  block_reduce_groups<0>(v, buf, lane, wave);
}

// ---------------------------------------------------------------- reverse
// FMD: mask of the stream's F block (Fj - I: the joint-dynamics mask FM plus the diagonal).  The chunk's operator blocks are
// staged in LDS once per workgroup (double-buffered: the next chunk's block is requested while this one is walked), so every
// candidate's stream crosses HBM once per pass whatever the number of its trials.
template <typename R, int M, int ND, int TPL, int CKT, Mask<M, M> FM>
__global__ void __launch_bounds__(LQG_ASP_TRIAL_BLOCK, (sizeof(R) == 4 ? 2 : 1)) k_asp_trial_rev(const R* __restrict__ ops_all, const TrialRevArgs<R> a) {
  constexpr int O = ND, RR = M - ND, BLK = LQG_ASP_TRIAL_BLOCK;
  using Ops = TrialOps<M, ND>;
  using SM = Sums<M, ND, FM>;
  constexpr auto FMD = mask_or(FM, mask_eye<M>());
  constexpr int CKN = (CKT + 1) * Ops::N;               // reals of one chunk's operator blocks + the block of the step before it
  constexpr int NW = BLK / 64, VPT = ReduceShape<SM::RAW>::VPT;
  __shared__ R lds[2 * CKT * NW * VPT];       // per-wave totals of the chunk's steps, double-buffered by chunk
  __shared__ R lops[2][CKN];
  const long sys = blockIdx.y;
  const long n0 = (long)blockIdx.x * (BLK * TPL) + threadIdx.x;
  const R* __restrict__ op = ops_all + sys * (long)(a.T + 1) * Ops::N;
  const long op_len = (long)(a.T + 1) * Ops::N;
  R* sums = a.sums + (((long)blockIdx.x * a.n_sys + sys) * a.T) * SM::N;
  // data rows: a wave-uniform row pointer (SGPRs) + a 32-bit per-lane BYTE offset of the trial — the form the hardware
  // addresses in one instruction (global_load v, v_offset, s[base]); a 64-bit per-lane pointer costs an address add per load
  const char* xbase = reinterpret_cast<const char*>(a.x.p + sys * a.x.sb);
  unsigned xo[TPL];
  auto xat = [&](int k, long t, int i) LQG_LAMBDA_INLINE -> R {
    const char* row = xbase + ((long)t * a.x.st + (long)i * a.x.sd) * (long)sizeof(R);
    return *reinterpret_cast<const R*>(row + xo[k]);
  };
  bool live[TPL];
  unsigned nn[TPL];
  R gw[TPL], pre[TPL][M], a1[TPL][O];
  LQG_UNROLL for (int k = 0; k < TPL; ++k) {
    long n = n0 + (long)k * BLK;
    live[k] = n < a.n_trials;
    n = live[k] ? n : (a.n_trials - 1);
    nn[k] = (unsigned)n;
    xo[k] = (unsigned)(n * a.x.sn * (long)sizeof(R));
    gw[k] = live[k] ? (a.g ? a.g[sys * a.g_sb + n * a.g_sn] : R(1)) : R(0);
    LQG_UNROLL for (int i = 0; i < M; ++i) pre[k][i] = R(0);
    // a_n(T) = Li_T' w_n(T): the mean entering row T from the kept c_{T-1} and the operator of step T - 1
    const R* src = a.tck + ((sys * (a.nckt + 1) + a.nckt) * RR) * a.npad + n;       // c_{T-1}
    const R* __restrict__ opT = op + (long)a.T * Ops::N;
    const R* __restrict__ opP = op + (long)(a.T - 1) * Ops::N;
    R cvp[M], dOT[O];
    LQG_UNROLL for (int i = 0; i < O; ++i) cvp[i] = xat(k, a.T - 1, i);
    LQG_UNROLL for (int i = 0; i < RR; ++i) cvp[O + i] = src[i * a.npad];
    LQG_UNROLL for (int i = 0; i < O; ++i) {
      R v = R(0);
      LQG_UNROLL for (int q = 0; q < M; ++q) if (FMD.b[i * M + q]) v += opP[Ops::F_OFF + i * M + q] * cvp[q];
      dOT[i] = v;
    }
    R w[O];
    int e = 0;
    LQG_UNROLL for (int i = 0; i < O; ++i) {
      R v = R(0);
      LQG_UNROLL for (int j = 0; j <= i; ++j)
        v += opT[Ops::L_OFF + (e++)] * ((xat(k, a.T, j) - cvp[j]) - dOT[j]);
      w[i] = v;
    }
    LQG_UNROLL for (int i = 0; i < O; ++i) {
      R v = R(0);
      LQG_UNROLL for (int q = i; q < O; ++q) v += opT[Ops::L_OFF + q * (q + 1) / 2 + i] * w[q];
      a1[k][i] = v;
    }
  }
  {   // sum of the workgroup's upstream weights (the same at every step): once
    R gl = R(0);
    LQG_UNROLL for (int k = 0; k < TPL; ++k) gl += gw[k];
    LQG_UNROLL for (int off = 32; off >= 1; off >>= 1) gl += __shfl_xor(gl, off);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = gl;
    __syncthreads();
    if (threadIdx.x == 0) {
      R tsum = R(0);
      LQG_UNROLL for (int w = 0; w < BLK / 64; ++w) tsum += lds[w];
      a.gsum[(long)blockIdx.x * a.n_sys + sys] = tsum;
    }
    __syncthreads();
  }
  // operator blocks of a chunk: thread `tid` fetches reals tid, tid + BLK, ... of the chunk
  constexpr int NLD = (CKN + BLK - 1) / BLK;
  R nx[NLD];
  LQG_UNROLL for (int q = 0; q < NLD; ++q) nx[q] = R(0);
  auto request = [&](int c) LQG_LAMBDA_INLINE {
    const long base = ((long)c * CKT - 1) * Ops::N;      // rows t0 - 1 .. t0 + CKT - 1 (the first restarts the chunk's mean state)
    LQG_UNROLL for (int q = 0; q < NLD; ++q) {
      const int i = q * BLK + (int)threadIdx.x;
      if (i < CKN && base + i >= 0 && base + i < op_len) nx[q] = op[base + i];
    }
  };
  auto publish = [&](int buf) LQG_LAMBDA_INLINE {
    LQG_UNROLL for (int q = 0; q < NLD; ++q) {
      const int i = q * BLK + (int)threadIdx.x;
      if (i < CKN) lops[buf][i] = nx[q];
    }
  };
  // the waves' totals of a chunk are summed and stored by the whole workgroup after the NEXT chunk's barrier: one barrier
  // per chunk instead of one per step
  auto flush = [&](int c, int rb) LQG_LAMBDA_INLINE {
    const R* buf = lds + (long)rb * CKT * NW * VPT;
    for (int i = threadIdx.x; i < CKT * SM::RAW; i += BLK) {
      const int j = i / SM::RAW, e = i - j * SM::RAW;
      const int t = c * CKT + j;
      if (t < a.T) {
        R tsum = buf[(j * NW) * VPT + e];
        LQG_UNROLL for (int w = 1; w < NW; ++w) tsum += buf[(j * NW + w) * VPT + e];
        sums[(long)t * SM::N + e] = tsum;
      }
    }
  };
  R wst[CKT][TPL][O], cst[CKT][TPL][RR];
  R xq[TPL][O];                                            // data rows requested one step ahead
  int rb = 0, obuf = 0;
#if !LQG_ASP_OPS_SCALAR
  request(a.nckt - 1);
#endif
  auto chunk = [&]<bool WHOLE>(const int c) LQG_LAMBDA_INLINE {        // WHOLE: CKT full steps (no per-step `t < T` tests)
    const int t0 = c * CKT;
#if LQG_ASP_OPS_SCALAR
    __syncthreads();
    if (c < a.nckt - 1) flush(c + 1, rb ^ 1);
    const R* __restrict__ lo = op + (long)t0 * Ops::N;
#else
    publish(obuf);
    __syncthreads();
    if (c < a.nckt - 1) flush(c + 1, rb ^ 1);
    if (c > 0) request(c - 1);
    const R* __restrict__ lo = lops[obuf] + Ops::N;
#endif
    // ---- recompute the chunk's (w, c)
    {
      R xprev[TPL][O], dO[TPL][O], muR[TPL][RR];
      const R* src = a.tck + ((sys * (a.nckt + 1) + c) * RR) * a.npad;
      LQG_UNROLL for (int k = 0; k < TPL; ++k) {
        LQG_UNROLL for (int i = 0; i < O; ++i) {
          xprev[k][i] = xat(k, t0 > 0 ? t0 - 1 : 0, i);
          xq[k][i] = xat(k, t0, i);
        }
      }
      if (c > 0) {
        // the mean state entering row t0 from the kept c_{t0-1}, x_{t0-1} and the operator of step t0 - 1 (k_trial_sp's update)
        const R* __restrict__ opp = lo - Ops::N;
        R Fp[M * M];
        LQG_UNROLL for (int i = 0; i < M * M; ++i) if (FMD.b[i]) Fp[i] = opp[Ops::F_OFF + i];
        LQG_UNROLL for (int k = 0; k < TPL; ++k) {
          R cvp[M];
          LQG_UNROLL for (int i = 0; i < O; ++i) cvp[i] = xprev[k][i];
          LQG_UNROLL for (int i = 0; i < RR; ++i) cvp[O + i] = src[i * a.npad + nn[k]];
          LQG_UNROLL for (int i = 0; i < M; ++i) {
            R v = R(0);
            LQG_UNROLL for (int q = 0; q < M; ++q) if (FMD.b[i * M + q]) v += Fp[i * M + q] * cvp[q];
            if (i < O) { dO[k][i < O ? i : 0] = v; }
            else muR[k][i >= O ? i - O : 0] = cvp[i] + v;
          }
        }
      } else {
        LQG_UNROLL for (int k = 0; k < TPL; ++k) {
          LQG_UNROLL for (int i = 0; i < O; ++i) dO[k][i] = R(0);
          LQG_UNROLL for (int i = 0; i < RR; ++i) muR[k][i] = R(0);
        }
      }
      LQG_UNROLL for (int j = 0; j < CKT; ++j) {
        const int t = t0 + j;
        if (WHOLE || t < a.T) {
          const R* __restrict__ opt = lo + j * Ops::N;
          R Li[O * (O + 1) / 2], U2[RR * O];
          LQG_UNROLL for (int i = 0; i < O * (O + 1) / 2; ++i) Li[i] = opt[Ops::L_OFF + i];
          LQG_UNROLL for (int i = 0; i < RR * O; ++i) U2[i] = opt[Ops::U_OFF + i];
          R Fv[M * M];
          LQG_UNROLL for (int i = 0; i < M * M; ++i) if (FMD.b[i]) Fv[i] = opt[Ops::F_OFF + i];
          LQG_UNROLL for (int k = 0; k < TPL; ++k) {
            R cv[M], w[O];
            // (data rows one step ahead in both passes: at two waves per SIMD a row requested where it is used exposes the whole
            // memory latency — the PMC pass showed this sweep issuing 0.45 of the VALU rate at its sustained clock)
#if LQG_ASP_XPREFETCH
            LQG_UNROLL for (int i = 0; i < O; ++i) cv[i] = xq[k][i];
            {   // next: step t + 1 of this pass, or — after the chunk's last executed step — this row again: the backward pass
                // starts where the recompute pass ends
              const int tn = (j + 1 < CKT && t + 1 < a.T) ? t + 1 : t;
              LQG_UNROLL for (int i = 0; i < O; ++i) xq[k][i] = xat(k, tn, i);
            }
#else
            LQG_UNROLL for (int i = 0; i < O; ++i) cv[i] = xat(k, t, i);
#endif
            int e = 0;
            LQG_UNROLL for (int i = 0; i < O; ++i) {
              R v = R(0);
              LQG_UNROLL for (int q = 0; q <= i; ++q) v += Li[e++] * ((cv[q] - xprev[k][q]) - dO[k][q]);
              w[i] = v;
              wst[j][k][i] = v;
            }
            LQG_UNROLL for (int p = 0; p < RR; ++p) {
              R v = muR[k][p];
              LQG_UNROLL for (int q = 0; q < O; ++q) v += U2[p * O + q] * w[q];
              cv[O + p] = v;
              cst[j][k][p] = v;
            }
            if (j + 1 < CKT) {                                  // (the state after the chunk's last step is not needed)
              LQG_UNROLL for (int i = 0; i < M; ++i) {
                R v = R(0);
                LQG_UNROLL for (int q = 0; q < M; ++q) if (FMD.b[i * M + q]) v += Fv[i * M + q] * cv[q];
                if (i < O) { dO[k][i < O ? i : 0] = v; }
                else muR[k][i >= O ? i - O : 0] = cv[i] + v;      // (the stream holds Fj - I)
              }
              LQG_UNROLL for (int i = 0; i < O; ++i) xprev[k][i] = cv[i];
            }
          }
        }
      }
    }
    // ---- backward
    LQG_UNROLL for (int j = CKT - 1; j >= 0; --j) {
      const int t = t0 + j;
      if (WHOLE || t < a.T) {
        const R* __restrict__ opt = lo + j * Ops::N;
        R Li[O * (O + 1) / 2], U2[RR * O];
        LQG_UNROLL for (int i = 0; i < O * (O + 1) / 2; ++i) Li[i] = opt[Ops::L_OFF + i];
        LQG_UNROLL for (int i = 0; i < RR * O; ++i) U2[i] = opt[Ops::U_OFF + i];
        R F2v[M * RR];                                              // Fj[:, o:] - I's columns
        LQG_UNROLL for (int i = 0; i < M; ++i)
          LQG_UNROLL for (int p = 0; p < RR; ++p) if (FMD.b[i * M + O + p]) F2v[i * RR + p] = opt[Ops::F_OFF + i * M + O + p];
        R acc[SM::RAW];
        LQG_UNROLL for (int i = 0; i < SM::RAW; ++i) acc[i] = R(0);
        LQG_UNROLL for (int k = 0; k < TPL; ++k) {
          R a0[O], cv[M], post[M], ch[RR];
          LQG_UNROLL for (int i = 0; i < O; ++i) {
            R v = R(0);
            LQG_UNROLL for (int q = i; q < O; ++q) v += Li[q * (q + 1) / 2 + i] * wst[j][k][q];
            a0[i] = v;
          }
#if LQG_ASP_XPREFETCH
          LQG_UNROLL for (int i = 0; i < O; ++i) cv[i] = xq[k][i];
          {
            const int tp = t > 0 ? t - 1 : 0;
            LQG_UNROLL for (int i = 0; i < O; ++i) xq[k][i] = xat(k, tp, i);
          }
#else
          LQG_UNROLL for (int i = 0; i < O; ++i) cv[i] = xat(k, t, i);
#endif
          LQG_UNROLL for (int p = 0; p < RR; ++p) cv[O + p] = cst[j][k][p];
          const R g = gw[k];
          LQG_UNROLL for (int i = 0; i < M; ++i) post[i] = pre[k][i] + (i < O ? g * a1[k][i < O ? i : 0] : R(0));
          {
            int e = 0;
            LQG_UNROLL for (int i = 0; i < O; ++i)
              LQG_UNROLL for (int q = 0; q <= i; ++q) acc[SM::W_OFF + (e++)] += g * a1[k][i] * a1[k][q];
          }
          {
            int e = 0;
            LQG_UNROLL for (int i = 0; i < M; ++i)
              LQG_UNROLL for (int q = 0; q < M; ++q)
                if (FM.b[i * M + q]) acc[SM::M_OFF + (e++)] += post[i] * cv[q];
          }
          // ch = Fj[:, o:]' post   (the stream holds Fj - I)
          LQG_UNROLL for (int p = 0; p < RR; ++p) {
            R v = post[O + p];
            LQG_UNROLL for (int i = 0; i < M; ++i)
              if (FMD.b[i * M + O + p]) v += F2v[i * RR + p] * post[i];
            ch[p] = v;
          }
          LQG_UNROLL for (int p = 0; p < RR; ++p)
            LQG_UNROLL for (int q = 0; q < O; ++q) acc[SM::C_OFF + p * O + q] += ch[p] * a0[q];
          // pre = [-Wm' ch ; ch],  Wm = U2 Li:  Wm' ch = Li' (U2' ch)
          R uc[O];
          LQG_UNROLL for (int q = 0; q < O; ++q) {
            R v = R(0);
            LQG_UNROLL for (int p = 0; p < RR; ++p) v += U2[p * O + q] * ch[p];
            uc[q] = v;
          }
          LQG_UNROLL for (int i = 0; i < O; ++i) {
            R v = R(0);
            LQG_UNROLL for (int q = i; q < O; ++q) v += Li[q * (q + 1) / 2 + i] * uc[q];
            pre[k][i] = -v;
          }
          LQG_UNROLL for (int p = 0; p < RR; ++p) pre[k][O + p] = ch[p];
          LQG_UNROLL for (int i = 0; i < O; ++i) a1[k][i] = a0[i];
        }
        wave_reduce_deposit<R, SM::RAW>(acc, lds + ((long)rb * CKT + j) * NW * VPT);
      }
    }
    obuf ^= 1;
    rb ^= 1;
  };
  for (int c = a.nckt - 1; c >= 0; --c) {
    // (whole chunks without per-step guards were measured at compile time and not kept: one basic block per chunk lets the
    // compiler hoist the chunk's 64 data loads, 408 registers spilled at two waves per SIMD)
    chunk.template operator()<false>(c);
  }
  __syncthreads();
  flush(0, rb ^ 1);
}

}  // namespace asp
}  // namespace lqg
