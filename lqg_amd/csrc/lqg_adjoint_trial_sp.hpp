// lqg_adjoint_trial_sp.hpp — the per-TRIAL half of the round-5 reverse-mode sweep (lqg_adjoint_sp.hpp), gfx950.
//
//   k_asp_trial_fwd   mean recursion + log-density over the operator stream (lqg/system.py:219-221, 244-248; the arithmetic of
//                     k_trial_sp), keeping the mean state every CKT steps
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

namespace lqg {
namespace asp {

template <typename R>
struct TrialRevArgs {
  const R* ops;              // operator stream [n_sys][T + 1][TrialOps::N]
  DTraj<R> x;
  const R* g;                // upstream weights, null = 1
  long g_sb, g_sn;
  R* ll;                     // value out (k_asp_trial_fwd), may be null
  long ll_sb, ll_sn;
  R* tck;                    // mean-state checkpoints [n_sys][nckt + 1][M][npad]
  long npad;
  int nckt;
  R* sums;                   // [parts = gridDim.x][n_sys][T][Sums::N]
  long n_sys, n_trials;
  int T;
};

// ---------------------------------------------------------------- forward
template <typename R, int M, int ND, int TPL, int CKT, Mask<M, M> FM>
__global__ void __launch_bounds__(LQG_ASP_TRIAL_BLOCK) k_asp_trial_fwd(const TrialRevArgs<R> a) {
  constexpr int O = ND, RR = M - ND, BLK = LQG_ASP_TRIAL_BLOCK;
  using Ops = TrialOps<M, ND>;
  // the FM of the sums is the mask of Fj; the operator stream holds Fj - I: its diagonal is always present
  constexpr auto FMD = mask_or(FM, mask_eye<M>());
  const long sys = blockIdx.y;
  const long n0 = (long)blockIdx.x * (BLK * TPL) + threadIdx.x;
  const R* __restrict__ op = a.ops + sys * (long)(a.T + 1) * Ops::N;
  const R* xr[TPL];
  bool live[TPL];
  R xprev[TPL][O], dO[TPL][O], muR[TPL][RR], part[TPL];
  double acc[TPL];
  long nn[TPL];
  LQG_UNROLL for (int k = 0; k < TPL; ++k) {
    long n = n0 + (long)k * BLK;
    live[k] = n < a.n_trials;
    n = live[k] ? n : (a.n_trials - 1);
    nn[k] = n;
    xr[k] = a.x.p + sys * a.x.sb + n * a.x.sn;
    LQG_UNROLL for (int i = 0; i < O; ++i) { xprev[k][i] = xr[k][i * a.x.sd]; dO[k][i] = R(0); }
    LQG_UNROLL for (int i = 0; i < RR; ++i) muR[k][i] = R(0);
    acc[k] = 0.0;
    part[k] = R(0);
  }
  const R none[1] = {R(0)};
  for (int t = 0; t <= a.T; ++t) {
    const R* __restrict__ opt = op + (long)t * Ops::N;
    if (t % CKT == 0 || t == a.T) {
      const int rec = t == a.T ? a.nckt : t / CKT;
      R* dst = a.tck + ((sys * (a.nckt + 1) + rec) * M) * a.npad;
      LQG_UNROLL for (int k = 0; k < TPL; ++k)
        if (live[k]) {
          LQG_UNROLL for (int i = 0; i < O; ++i) dst[i * a.npad + nn[k]] = dO[k][i];
          LQG_UNROLL for (int i = 0; i < RR; ++i) dst[(O + i) * a.npad + nn[k]] = muR[k][i];
        }
    }
    R Li[O * (O + 1) / 2];
    LQG_UNROLL for (int i = 0; i < O * (O + 1) / 2; ++i) Li[i] = opt[Ops::L_OFF + i];
    const R hlc = opt[Ops::H_OFF];
    LQG_UNROLL for (int k = 0; k < TPL; ++k) {
      R cv[M], w[O];
      LQG_UNROLL for (int i = 0; i < O; ++i) cv[i] = xr[k][(long)t * a.x.st + i * a.x.sd];
      R zz = R(0);
      {
        int e = 0;
        LQG_UNROLL for (int i = 0; i < O; ++i) {
          R v = R(0);
          LQG_UNROLL for (int j = 0; j <= i; ++j) v += Li[e++] * ((cv[j] - xprev[k][j]) - dO[k][j]);
          w[i] = v;
          zz += v * v;
        }
      }
      if (t > 0) part[k] += R(0.5) * zz + hlc;
      if ((t & 7) == 0 || t == a.T) { acc[k] -= (double)part[k]; part[k] = R(0); }
      if (t < a.T) {
        LQG_UNROLL for (int p = 0; p < RR; ++p) {
          R v = muR[k][p];
          LQG_UNROLL for (int j = 0; j < O; ++j) v += opt[Ops::U_OFF + p * O + j] * w[j];
          cv[O + p] = v;
        }
        R mn[M];
        trial_mean_rows<R, M, ND, FMD, false, 1, 0>(none, opt, cv, mn);
        LQG_UNROLL for (int i = 0; i < O; ++i) { dO[k][i] = mn[i]; xprev[k][i] = cv[i]; }
        LQG_UNROLL for (int p = 0; p < RR; ++p) muR[k][p] = cv[O + p] + mn[O + p];     // (the stream holds Fj - I)
      }
    }
  }
  if (a.ll) {
    LQG_UNROLL for (int k = 0; k < TPL; ++k)
      if (live[k]) a.ll[sys * a.ll_sb + nn[k] * a.ll_sn] = (R)acc[k];
  }
}

// ---------------------------------------------------------------- wave butterfly: V <= 32 values per lane -> lane l (< 32)
// holds the wave's total of value bitrev5(l)
LQG_DEV int bitrev5(int l) { return ((l & 1) << 4) | ((l & 2) << 2) | (l & 4) | ((l & 8) >> 2) | ((l & 16) >> 4); }

template <typename R, int V>
LQG_DEV R wave_transpose_reduce(const R (&v)[V], int base) {
  static_assert(V > 0, "values");
  const int lane = threadIdx.x & 63;
  R cur[32];
  LQG_UNROLL for (int i = 0; i < 32; ++i) cur[i] = R(0);
  LQG_UNROLL for (int i = 0; i < 32; ++i)
    if (base + i < V) cur[i] = v[base + i < V ? base + i : 0];
  // which of the 32 slots can be non-zero is known at compile time; stages on all-zero pairs fold away only partly — the
  // explicit bound below skips them
  constexpr int h0 = 16;
  int hbit = 0;
  (void)hbit;
  LQG_UNROLL for (int s = 0; s < 5; ++s) {
    const int h = h0 >> s;
    const bool up = (lane >> s) & 1;
    LQG_UNROLL for (int k = 0; k < 16; ++k)
      if (k < h) {
        const R lo = cur[k], hi = cur[k + h];
        const R keepv = up ? hi : lo;
        const R send = up ? lo : hi;
        cur[k] = keepv + __shfl_xor(send, 1 << s);
      }
  }
  return cur[0] + __shfl_xor(cur[0], 32);
}

// reduce V per-lane values over the workgroup and store them contiguously at out[0 .. V); `lds` holds 2 x NW x VP reals,
// `parity` alternates per call so that ONE barrier per call suffices
template <typename R, int V>
LQG_DEV void block_reduce_store(const R (&v)[V], R* lds, int parity, R* __restrict__ out) {
  constexpr int NW = LQG_ASP_TRIAL_BLOCK / 64;
  constexpr int NG = (V + 31) / 32, VP = NG * 32;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  R* buf = lds + (long)parity * NW * VP;
  LQG_UNROLL for (int gq = 0; gq < NG; ++gq) {
    const R tot = wave_transpose_reduce<R, V>(v, gq * 32);
    if (lane < 32) buf[wave * VP + gq * 32 + bitrev5(lane)] = tot;
  }
  __syncthreads();
  if ((int)threadIdx.x < V) {
    R tsum = buf[threadIdx.x];
    LQG_UNROLL for (int w = 1; w < NW; ++w) tsum += buf[w * VP + threadIdx.x];
    out[threadIdx.x] = tsum;
  }
}

// ---------------------------------------------------------------- reverse
template <typename R, int M, int ND, int TPL, int CKT, Mask<M, M> FM>
__global__ void __launch_bounds__(LQG_ASP_TRIAL_BLOCK) k_asp_trial_rev(const TrialRevArgs<R> a) {
  constexpr int O = ND, RR = M - ND, BLK = LQG_ASP_TRIAL_BLOCK;
  using Ops = TrialOps<M, ND>;
  using SM = Sums<M, ND, FM>;
  constexpr auto FMD = mask_or(FM, mask_eye<M>());
  constexpr int NG = (SM::RAW + 31) / 32;
  __shared__ R lds[2 * (BLK / 64) * NG * 32];
  const long sys = blockIdx.y;
  const long n0 = (long)blockIdx.x * (BLK * TPL) + threadIdx.x;
  const R* __restrict__ op = a.ops + sys * (long)(a.T + 1) * Ops::N;
  R* sums = a.sums + (((long)blockIdx.x * a.n_sys + sys) * a.T) * SM::N;
  const R* xr[TPL];
  bool live[TPL];
  long nn[TPL];
  R gw[TPL], pre[TPL][M], a1[TPL][O];
  const R none[1] = {R(0)};
  LQG_UNROLL for (int k = 0; k < TPL; ++k) {
    long n = n0 + (long)k * BLK;
    live[k] = n < a.n_trials;
    n = live[k] ? n : (a.n_trials - 1);
    nn[k] = n;
    xr[k] = a.x.p + sys * a.x.sb + n * a.x.sn;
    gw[k] = live[k] ? (a.g ? a.g[sys * a.g_sb + n * a.g_sn] : R(1)) : R(0);
    LQG_UNROLL for (int i = 0; i < M; ++i) pre[k][i] = R(0);
    // a_n(T) = Li_T' w_n(T) from the final mean state
    const R* src = a.tck + ((sys * (a.nckt + 1) + a.nckt) * M) * a.npad + n;
    const R* __restrict__ opT = op + (long)a.T * Ops::N;
    R w[O];
    int e = 0;
    LQG_UNROLL for (int i = 0; i < O; ++i) {
      R v = R(0);
      LQG_UNROLL for (int j = 0; j <= i; ++j)
        v += opT[Ops::L_OFF + (e++)] * ((xr[k][(long)a.T * a.x.st + j * a.x.sd] - xr[k][(long)(a.T - 1) * a.x.st + j * a.x.sd]) - src[j * a.npad]);
      w[i] = v;
    }
    LQG_UNROLL for (int i = 0; i < O; ++i) {
      R v = R(0);
      LQG_UNROLL for (int q = i; q < O; ++q) v += opT[Ops::L_OFF + q * (q + 1) / 2 + i] * w[q];
      a1[k][i] = v;
    }
  }
  R wst[CKT][TPL][O], cst[CKT][TPL][RR];
  int parity = 0;
  for (int c = a.nckt - 1; c >= 0; --c) {
    const int t0 = c * CKT;
    // ---- recompute the chunk's (w, c)
    {
      R xprev[TPL][O], dO[TPL][O], muR[TPL][RR];
      const R* src = a.tck + ((sys * (a.nckt + 1) + c) * M) * a.npad;
      LQG_UNROLL for (int k = 0; k < TPL; ++k) {
        LQG_UNROLL for (int i = 0; i < O; ++i) {
          dO[k][i] = src[i * a.npad + nn[k]];
          xprev[k][i] = xr[k][(long)(t0 > 0 ? t0 - 1 : 0) * a.x.st + i * a.x.sd];
        }
        LQG_UNROLL for (int i = 0; i < RR; ++i) muR[k][i] = src[(O + i) * a.npad + nn[k]];
      }
      LQG_UNROLL for (int j = 0; j < CKT; ++j) {
        const int t = t0 + j;
        if (t < a.T) {
          const R* __restrict__ opt = op + (long)t * Ops::N;
          LQG_UNROLL for (int k = 0; k < TPL; ++k) {
            R cv[M], w[O];
            LQG_UNROLL for (int i = 0; i < O; ++i) cv[i] = xr[k][(long)t * a.x.st + i * a.x.sd];
            int e = 0;
            LQG_UNROLL for (int i = 0; i < O; ++i) {
              R v = R(0);
              LQG_UNROLL for (int q = 0; q <= i; ++q) v += opt[Ops::L_OFF + (e++)] * ((cv[q] - xprev[k][q]) - dO[k][q]);
              w[i] = v;
              wst[j][k][i] = v;
            }
            LQG_UNROLL for (int p = 0; p < RR; ++p) {
              R v = muR[k][p];
              LQG_UNROLL for (int q = 0; q < O; ++q) v += opt[Ops::U_OFF + p * O + q] * w[q];
              cv[O + p] = v;
              cst[j][k][p] = v;
            }
            R mn[M];
            trial_mean_rows<R, M, ND, FMD, false, 1, 0>(none, opt, cv, mn);
            LQG_UNROLL for (int i = 0; i < O; ++i) { dO[k][i] = mn[i]; xprev[k][i] = cv[i]; }
            LQG_UNROLL for (int p = 0; p < RR; ++p) muR[k][p] = cv[O + p] + mn[O + p];
          }
        }
      }
    }
    // ---- backward
    LQG_UNROLL for (int j = CKT - 1; j >= 0; --j) {
      const int t = t0 + j;
      if (t < a.T) {
        const R* __restrict__ opt = op + (long)t * Ops::N;
        R acc[SM::RAW];
        LQG_UNROLL for (int i = 0; i < SM::RAW; ++i) acc[i] = R(0);
        LQG_UNROLL for (int k = 0; k < TPL; ++k) {
          R a0[O], cv[M], post[M], ch[RR];
          LQG_UNROLL for (int i = 0; i < O; ++i) {
            R v = R(0);
            LQG_UNROLL for (int q = i; q < O; ++q) v += opt[Ops::L_OFF + q * (q + 1) / 2 + i] * wst[j][k][q];
            a0[i] = v;
          }
          LQG_UNROLL for (int i = 0; i < O; ++i) cv[i] = xr[k][(long)t * a.x.st + i * a.x.sd];
          LQG_UNROLL for (int p = 0; p < RR; ++p) cv[O + p] = cst[j][k][p];
          const R g = gw[k];
          LQG_UNROLL for (int i = 0; i < M; ++i) post[i] = pre[k][i] + (i < O ? g * a1[k][i < O ? i : 0] : R(0));
          acc[SM::G_OFF] += g;
          {
            int e = 0;
            LQG_UNROLL for (int i = 0; i < O; ++i)
              LQG_UNROLL for (int q = 0; q <= i; ++q) acc[SM::W_OFF + (e++)] += g * a1[k][i] * a1[k][q];
          }
          LQG_UNROLL for (int i = 0; i < M; ++i)
            LQG_UNROLL for (int q = 0; q < M; ++q)
              if (FM.b[i * M + q]) acc[SM::mc(i, q)] += post[i] * cv[q];
          // ch = Fj[:, o:]' post   (the stream holds Fj - I)
          LQG_UNROLL for (int p = 0; p < RR; ++p) {
            R v = post[O + p];
            LQG_UNROLL for (int i = 0; i < M; ++i)
              if (FMD.b[i * M + O + p]) v += opt[Ops::F_OFF + i * M + O + p] * post[i];
            ch[p] = v;
          }
          LQG_UNROLL for (int p = 0; p < RR; ++p)
            LQG_UNROLL for (int q = 0; q < O; ++q) acc[SM::C_OFF + p * O + q] += ch[p] * a0[q];
          // pre = [-Wm' ch ; ch],  Wm = U2 Li:  Wm' ch = Li' (U2' ch)
          R uc[O];
          LQG_UNROLL for (int q = 0; q < O; ++q) {
            R v = R(0);
            LQG_UNROLL for (int p = 0; p < RR; ++p) v += opt[Ops::U_OFF + p * O + q] * ch[p];
            uc[q] = v;
          }
          LQG_UNROLL for (int i = 0; i < O; ++i) {
            R v = R(0);
            LQG_UNROLL for (int q = i; q < O; ++q) v += opt[Ops::L_OFF + q * (q + 1) / 2 + i] * uc[q];
            pre[k][i] = -v;
          }
          LQG_UNROLL for (int p = 0; p < RR; ++p) pre[k][O + p] = ch[p];
          LQG_UNROLL for (int i = 0; i < O; ++i) a1[k][i] = a0[i];
        }
        block_reduce_store<R, SM::RAW>(acc, lds, parity, sums + (long)t * SM::N);
        parity ^= 1;
      }
    }
  }
}

}  // namespace asp
}  // namespace lqg
