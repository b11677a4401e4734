// lqg_adjoint_sp.hpp — reverse-mode sweep of the LQG log-likelihood on the FORWARD path's own split (round 5), gfx950.
//
// What the reference obtains from jax.grad / jax.value_and_grad of System.log_likelihood through all three scans
// (lqg/optim.py:142-147 `jit(grad(fun))`, lqg/infer/utils.py:14-18 NUTS, lqg/infer/mle.py:14-25 SVI-Adam).  The round-1 sweep
// (lqg_adjoint.hpp) maps one (system, trial) PAIR to a lane and so repeats every matrix adjoint once per trial.  Here, as in the
// forward path (lqg_kernels_sp.hpp):
//   per SYSTEM, one lane   Riccati, Kalman, joint system, moment recursion and ALL their adjoints — the matrix part of the path is
//                          data-independent and the adjoint recursion is linear in (mu-bar, Sigma-bar), so Sigma-bar = sum over
//                          the trials runs once;
//   per TRIAL, one lane    the mean recursion forward, the mu-bar recursion backward over the same operator stream; per step the
//                          trials of a system contribute only their TRIAL SUMS (Sums below), reduced across the workgroup;
//   1 or 2 trials          swept in the system's lane (the headline shape: 2^18 candidates x one trajectory each).
// Structure (PAT masks) is resolved at compile time through lqg_sparse.hpp: the adjoint of a structurally zero entry is never
// formed.  Nothing per step is parked in HBM except checkpoints every CK steps (P, Sigma, trial means; S for the Riccati
// recursion); the chunk is recomputed in registers, then walked backward.  Time-invariant specs, no affine cost terms (what the
// pattern libraries serve); everything else keeps the round-1 kernels.  CPU restatement, same order and intermediate
// quantities: oracle/lqg_adjoint_split_np.py.
//
// Bars of the hoisted products (FAa = Fa Aa, FAd = Fd Ad, DB = Fd Bd - Fa Ba, N2 = Fd N1, N3 = Fd N1 Fd' + WWd) are accumulated
// over time and chained to the stored matrices ONCE after the sweep.
#pragma once
#include <type_traits>
#include <utility>

#include "lqg_adjoint.hpp"
#include "lqg_kernels_sp.hpp"

#ifndef LQG_ASP_STACK0
#define LQG_ASP_STACK0 0          // 1: the state before a chunk's first step stays in registers too; 0: re-read from its checkpoint (measured: 4.14 vs 4.32 ms)
#endif
#ifndef LQG_ASP_FWD_WAVES_F32
#define LQG_ASP_FWD_WAVES_F32 4   // waves per SIMD the fp32 forward system sweep of a small joint dimension (x + b <= 5) is held to: it sits at
#endif                            // 128 +- 3 VGPRs, and 131 cost a whole wave of occupancy (1.49 -> 2.11 ms per 2^18 systems, same-box A/B, round 6)

namespace lqg {
namespace asp {

// ---------------------------------------------------------------- small additions to the masked-matrix algebra
template <int R0, int C0, int NR, int NC, int M, int N>
constexpr Mask<NR, NC> mask_blk(const Mask<M, N>& a) {
  Mask<NR, NC> r{};
  for (int i = 0; i < NR; ++i)
    for (int j = 0; j < NC; ++j) r.b[i * NC + j] = a(R0 + i, C0 + j);
  return r;
}
template <int R0, int C0, int NR, int NC, typename R, int M, int N, Mask<M, N> MK>
LQG_DEV auto blk(const Mat<R, M, N, MK>& a) {
  constexpr auto MR = mask_blk<R0, C0, NR, NC>(MK);
  Mat<R, NR, NC, MR> r;
  LQG_UNROLL for (int i = 0; i < NR; ++i)
    LQG_UNROLL for (int j = 0; j < NC; ++j) if (MR(i, j)) r.v[i * NC + j] = a.v[(R0 + i) * N + C0 + j];
  return r;
}
template <typename R, int M, int N, Mask<M, N> MK>
LQG_DEV auto scaled(const Mat<R, M, N, MK>& a, R s) {
  Mat<R, M, N, MK> r;
  LQG_UNROLL for (int i = 0; i < M * N; ++i) if (MK.b[i]) r.v[i] = s * a.v[i];
  return r;
}
// acc += s * x on the accumulator's FIXED mask (entries of x outside it are adjoints of structural zeros: dropped)
template <typename R, int M, int N, Mask<M, N> MA, Mask<M, N> MX>
LQG_DEV void accum(Mat<R, M, N, MA>& acc, const Mat<R, M, N, MX>& x, R s = R(1)) {
  LQG_UNROLL for (int i = 0; i < M * N; ++i) if (MA.b[i] && MX.b[i]) acc.v[i] += s * x.v[i];
}
template <typename R, int M, int N, Mask<M, N> MK>
LQG_DEV void set_zero(Mat<R, M, N, MK>& a) {
  LQG_UNROLL for (int i = 0; i < M * N; ++i) if (MK.b[i]) a.v[i] = R(0);
}
template <typename R, int N, Mask<N, N> MK>
LQG_DEV auto sym_part(const Mat<R, N, N, MK>& a) {
  constexpr auto MR = mask_or(MK, mask_t(MK));
  Mat<R, N, N, MR> r;
  LQG_UNROLL for (int i = 0; i < N; ++i)
    LQG_UNROLL for (int j = i; j < N; ++j)
      if (MR(i, j)) {
        R v = R(0.5) * ((MK(i, j) ? a.v[i * N + j] : R(0)) + (MK(j, i) ? a.v[j * N + i] : R(0)));
        r.v[i * N + j] = v;
        r.v[j * N + i] = v;
      }
  return r;
}
template <typename R, int M, int N, Mask<M, N> MK>
LQG_DEV void store_col(R* __restrict__ p, long ld, const Mat<R, M, N, MK>& a) {       // dense [element][lane] image
  LQG_UNROLL for (int i = 0; i < M * N; ++i) p[i * ld] = MK.b[i] ? a.v[i] : R(0);
}
template <typename R, int N>
LQG_DEV void store_tri_arr(R* __restrict__ p, long ld, const R (&A)[N * N]) {
  int k = 0;
  LQG_UNROLL for (int i = 0; i < N; ++i)
    LQG_UNROLL for (int j = 0; j <= i; ++j) p[(k++) * ld] = A[i * N + j];
}
template <typename R, int N>
LQG_DEV void load_tri_arr(const R* __restrict__ p, long ld, R (&A)[N * N]) {
  int k = 0;
  LQG_UNROLL for (int i = 0; i < N; ++i)
    LQG_UNROLL for (int j = 0; j <= i; ++j) {
      R v = p[(k++) * ld];
      A[i * N + j] = v;
      A[j * N + i] = v;
    }
}

// ---------------------------------------------------------------- the trial sums of one (system, step)
// W2 (lower triangle, by rows) = sum_n g_n a_n(t+1) a_n(t+1)' | CA[RR, O] = sum_n ch_n a_n(t)' |
// MC on the mask FM of the joint dynamics (row-major order of its set entries) = sum_n post_n c_n'
// (sum_n g_n, the same at every step, travels once per system: AspArgs::gsum)
template <int M, int ND, Mask<M, M> FM>
struct Sums {
  static constexpr int O = ND, RR = M - ND;
  static constexpr int W_OFF = 0, NW = O * (O + 1) / 2;
  static constexpr int C_OFF = W_OFF + NW;
  static constexpr int M_OFF = C_OFF + RR * O;
  static constexpr int NMC = FM.count();
  static constexpr int RAW = M_OFF + NMC;
  static constexpr int N = (RAW + 3) / 4 * 4;
  static constexpr int mc(int i, int j) {
    int c = 0;
    for (int e = 0; e < i * M + j; ++e) c += FM.b[e] ? 1 : 0;
    return M_OFF + c;
  }
};

// ---------------------------------------------------------------- arguments
template <typename R>
struct AspArgs {
  ForwardArgs<R> f;          // specs, Sigma0, x (fused), ll (fused), ops (stream)
  RiccatiArgs<R> rc;         // actor cost matrices, S checkpoints (rc.Ls), eps
  const R* g;                // upstream weights [s * g_sb + n * g_sn], null = 1
  long g_sb, g_sn;
  long ll_sn;
  R* ck;                     // system checkpoints [nck + 1][CKW][ldb]   (CKW: see sys kernels)
  R* sums;                   // trial sums [parts][n_sys][T][Sums::N]    (stream path)
  const R* gsum;             // [parts][n_sys] sums of the upstream weights     (stream path)
  int parts;
  R* Lbar;                   // [T][NU * NB][ldb]
  R* Kbar;                   // [T][NB * NY][ldb]: K-bar_t without its P-bar term, the set entries of K's mask first (k_asp_sys_rev -> k_asp_kal_rev)
  R* out;                    // gradient [Layout::TOTAL][ld]
  long ld;
};

// ---------------------------------------------------------------- per-system constants with their structural masks
template <typename R, int NX, int NB, int NU, int NY, typename PAT>
struct SysConst {
  Mat<R, NB, NB, PAT::Aa> Aa;
  Mat<R, NB, NU, PAT::Ba> Ba;
  Mat<R, NY, NB, PAT::Fa> Fa;
  Mat<R, NB, NB, PAT::VVa> VVa;
  Mat<R, NY, NY, PAT::WWa> WWa;
  Mat<R, NX, NX, PAT::Ad> Ad;
  Mat<R, NX, NU, PAT::Bd> Bd;
  Mat<R, NX, NX, PAT::N1> N1;
  Mat<R, NY, NX, PAT::Fd> Fd;
  Mat<R, NY, NY, PAT::WWd> WWd;
  Mat<R, NY, NB, mask_and(mask_mul(PAT::Fa, PAT::Aa), PAT::FAa)> FAa;
  Mat<R, NY, NX, mask_and(mask_mul(PAT::Fd, PAT::Ad), PAT::FAd)> FAd;
  Mat<R, NY, NU, mask_and(mask_or(mask_mul(PAT::Fd, PAT::Bd), mask_mul(PAT::Fa, PAT::Ba)), PAT::DB)> DB;
  Mat<R, NY, NX, mask_and(mask_mul(PAT::Fd, PAT::N1), PAT::N2)> N2;
  Mat<R, NY, NY, PAT::N3> N3;
  LQG_DEV void load(const ForwardArgs<R>& a, long s) {
    Aa = load_masked<R, NB, NB, PAT::Aa>(a.aA.p + s * a.aA.sb, a.aA.sr, a.aA.sc);
    Ba = load_masked<R, NB, NU, PAT::Ba>(a.aB.p + s * a.aB.sb, a.aB.sr, a.aB.sc);
    Fa = load_masked<R, NY, NB, PAT::Fa>(a.aF.p + s * a.aF.sb, a.aF.sr, a.aF.sc);
    VVa = load_gram_masked<R, NB, PAT::VVa>(a.aV.p + s * a.aV.sb, a.aV.sr, a.aV.sc, a.nva);
    WWa = load_gram_masked<R, NY, PAT::WWa>(a.aW.p + s * a.aW.sb, a.aW.sr, a.aW.sc, a.nwa);
    Ad = load_masked<R, NX, NX, PAT::Ad>(a.dA.p + s * a.dA.sb, a.dA.sr, a.dA.sc);
    Bd = load_masked<R, NX, NU, PAT::Bd>(a.dB.p + s * a.dB.sb, a.dB.sr, a.dB.sc);
    N1 = load_gram_masked<R, NX, PAT::N1>(a.dV.p + s * a.dV.sb, a.dV.sr, a.dV.sc, a.nvd);
    Fd = load_masked<R, NY, NX, PAT::Fd>(a.dF.p + s * a.dF.sb, a.dF.sr, a.dF.sc);
    WWd = load_gram_masked<R, NY, PAT::WWd>(a.dW.p + s * a.dW.sb, a.dW.sr, a.dW.sc, a.nwd);
    FAa = restrict_to<PAT::FAa>(mul(Fa, Aa));
    FAd = restrict_to<PAT::FAd>(mul(Fd, Ad));
    DB = restrict_to<PAT::DB>(sub(mul(Fd, Bd), mul(Fa, Ba)));
    const auto FdN1 = mul(Fd, N1);
    N2 = restrict_to<PAT::N2>(FdN1);
    {
      const auto n3 = restrict_to<PAT::N3>(mul_nt_sym_add(FdN1, Fd, WWd));
      LQG_UNROLL for (int i = 0; i < NY * NY; ++i)
        if (PAT::N3.b[i]) N3.v[i] = decltype(n3)::mask.b[i] ? n3.v[i] : R(0);
    }
  }
};

// ---------------------------------------------------------------- one forward step of the system part
// From (P_t, Sigma_t, L_t): Kalman step kf.py:10-14, joint system system.py:167-207, conditioning on the observed block
// system.py:219-230 in Cholesky / Schur form (lqg_kernels.hpp).  Everything a consumer needs is handed to `use` (a generic
// lambda): the forward kernels advance the state and emit operators, the reverse kernel differentiates.
// Kalman step kf.py:10-14 from P_t: everything the joint system and the Kalman adjoint read
template <typename R, int NX, int NB, int NU, int NY, typename PAT, Mask<NB, NB> PM>
LQG_DEV auto kalman_part(const SysConst<R, NX, NB, NU, NY, PAT>& c, const Mat<R, NB, NB, PM>& Pm) {
  const auto AP = mul(c.Aa, Pm);
  const auto Pp = mul_nt_sym_add(AP, c.Aa, c.VVa);                      // kf.py:10
  const auto FP = mul(c.Fa, Pp);
  const auto Gi = spd_inverse_masked(mul_nt_sym_add(FP, c.Fa, c.WWa)); // kf.py:11
  const auto K = mul_tn(FP, Gi);                                       // kf.py:12
  const auto Pn = sym_sub_mul(Pp, K, FP);                              // kf.py:14
  struct Out {
    std::remove_cvref_t<decltype(AP)> AP; std::remove_cvref_t<decltype(Pp)> Pp; std::remove_cvref_t<decltype(FP)> FP;
    std::remove_cvref_t<decltype(Gi)> Gi; std::remove_cvref_t<decltype(K)> K; std::remove_cvref_t<decltype(Pn)> Pn;
  };
  return Out{AP, Pp, FP, Gi, K, Pn};
}
// the Kalman gain's type (its structural mask follows from the pattern and the mask of P)
template <typename T>
__device__ T&& dev_declval() noexcept;                   // (std::declval is a host function; never defined, unevaluated use only)
template <typename R, int NX, int NB, int NU, int NY, typename PAT, Mask<NB, NB> PM>
using KalmanGain = decltype(kalman_part(dev_declval<const SysConst<R, NX, NB, NU, NY, PAT>&>(), dev_declval<const Mat<R, NB, NB, PM>&>()).K);

// joint system system.py:167-207 and conditioning on the observed block system.py:219-230 (Cholesky / Schur form,
// lqg_kernels.hpp) from (K_t, Sigma_t, L_t); `use(BK, Fj, KN2, KN3, GG, Li, dinv, U2, C, F2, F2C)`
template <int ND, typename R, int NX, int NB, int NU, int NY, typename PAT, typename KT, typename Use>
LQG_DEV void joint_part(const SysConst<R, NX, NB, NU, NY, PAT>& c, const KT& K, R (&Sg)[(NX + NB) * (NX + NB)],
                        const Mat<R, NU, NB>& L, const bool first, Use&& use) {
  constexpr int M = NX + NB, O = ND, RR = M - ND;
  const auto BK = add(c.Ba, mul(K, c.DB));
  const auto Fj = block2x2(c.Ad, mul(c.Bd, L), mul(K, c.FAd), add(sub(c.Aa, mul(K, c.FAa)), mul(BK, L)));
  const auto KN2 = mul(K, c.N2);
  const auto KN3 = mul(K, c.N3);
  const auto GG = block2x2(c.N1, transpose(KN2), KN2, mul_nt_sym_add(KN3, K, Mat<R, NB, NB, mask_none<NB, NB>()>{}));
  if (first) to_dense(GG, Sg);                                         // Sigma_0 := G_0 G_0'   system.py:212
  R Li[O * O], U2[RR * O], dinv[O];
  {
    R Soo[O * O], Lc[O * O];
    LQG_UNROLL for (int i = 0; i < O; ++i)
      LQG_UNROLL for (int j = 0; j < O; ++j) Soo[i * O + j] = Sg[i * M + j];
    chol_lower<R, O>(Soo, Lc, dinv);
    tri_inverse_lower<R, O>(Lc, dinv, Li);
    LQG_UNROLL for (int p = 0; p < RR; ++p)
      LQG_UNROLL for (int j = 0; j < O; ++j) {
        R v = R(0);
        LQG_UNROLL for (int k = 0; k <= j; ++k) v += Sg[(O + p) * M + k] * Li[j * O + k];
        U2[p * O + j] = v;
      }
  }
  Mat<R, RR, RR> C;
  LQG_UNROLL for (int p = 0; p < RR; ++p)
    LQG_UNROLL for (int q = p; q < RR; ++q) {
      R v = Sg[(O + p) * M + O + q];
      LQG_UNROLL for (int j = 0; j < O; ++j) v -= U2[p * O + j] * U2[q * O + j];
      C.v[p * RR + q] = v;
      C.v[q * RR + p] = v;
    }
  const auto F2 = cols<O, RR>(Fj);
  const auto F2C = mul(F2, C);
  use(BK, Fj, KN2, KN3, GG, Li, dinv, U2, C, F2, F2C);
}

// one forward step of the system part from (P_t, Sigma_t, L_t): both halves; everything a consumer needs is handed to `use` (a
// generic lambda): the forward kernel advances the state and emits operators.
template <int ND, typename R, int NX, int NB, int NU, int NY, typename PAT, Mask<NB, NB> PM, typename Use>
LQG_DEV void sys_step(const SysConst<R, NX, NB, NU, NY, PAT>& c, const Mat<R, NB, NB, PM>& Pm, R (&Sg)[(NX + NB) * (NX + NB)],
                      const Mat<R, NU, NB>& L, const bool first, Use&& use) {
  const auto kp = kalman_part(c, Pm);
  joint_part<ND>(c, kp.K, Sg, L, first, [&](const auto& BK, const auto& Fj, const auto& KN2, const auto& KN3, const auto& GG,
                                            const auto& Li, const auto& dinv, const auto& U2, const auto& C, const auto& F2,
                                            const auto& F2C) LQG_LAMBDA_INLINE {
    use(kp.AP, kp.Pp, kp.FP, kp.Gi, kp.K, kp.Pn, BK, Fj, KN2, KN3, GG, Li, dinv, U2, C, F2, F2C);
  });
}

// control gains of the steps t0 .. t0 + CK - 1 recomputed backward from the checkpoint S_{t0 + CK} (k_riccati_sp<CK>)
template <typename R, int NB, int NU, int CK, bool WHOLE = false, typename MA, typename MB, typename MQ, typename MR>
LQG_DEV void refill_gains(const RiccatiArgs<R>& rc, long s, int t0, const MA& Aa, const MB& Ba, const MQ& rQ, const MR& rR,
                          R (&Lbuf)[CK][NU * NB]) {
  constexpr int NS = NB * (NB + 1) / 2;
  R Sr[NB * NB];
  const R* src = rc.Ls + (long)(t0 / CK) * NS * rc.ldb + s;
  int e = 0;
  LQG_UNROLL for (int i = 0; i < NB; ++i)
    LQG_UNROLL for (int j = i; j < NB; ++j) { const R v = src[(e++) * rc.ldb]; Sr[i * NB + j] = v; Sr[j * NB + i] = v; }
  LQG_UNROLL for (int j = CK - 1; j >= 0; --j)
    if (WHOLE || t0 + j < rc.T) riccati_step_sp<R, NB, NU>(Sr, Aa, Ba, rQ, rR, rc.eps, Lbuf[j]);
}

// the mean state of one trial in deviation form (lqg_kernels.hpp): observed mean = x_{t-1} + dO
template <typename R, int M, int ND>
struct TrialState {
  R dO[ND], muR[M - ND];
};

// one trial's forward step over the step's operators: returns w; advances the state when `advance`
template <typename R, int M, int ND, typename FJ>
LQG_DEV void trial_forward(const FJ& Fj, const R (&Li)[ND * ND], const R (&U2)[(M - ND) * ND], const R (&xt)[ND],
                           const R (&xprev)[ND], TrialState<R, M, ND>& st, R (&w)[ND], R (&cv)[M], const bool advance) {
  constexpr int O = ND, RR = M - ND;
  LQG_UNROLL for (int i = 0; i < O; ++i) {
    R v = R(0);
    LQG_UNROLL for (int j = 0; j <= i; ++j) v += Li[i * O + j] * ((xt[j] - xprev[j]) - st.dO[j]);
    w[i] = v;
  }
  LQG_UNROLL for (int j = 0; j < O; ++j) cv[j] = xt[j];
  LQG_UNROLL for (int p = 0; p < RR; ++p) {
    R v = st.muR[p];
    LQG_UNROLL for (int j = 0; j < O; ++j) v += U2[p * O + j] * w[j];
    cv[O + p] = v;
  }
  if (advance) {
    R mn[M];
    dev_matvec_row<O, 0>(Fj, cv, mn);                      // rows < O as deviation from x_t: ((Fj - I) cv)[i]
    LQG_UNROLL for (int i = 0; i < O; ++i) st.dO[i] = mn[i];
    LQG_UNROLL for (int p = 0; p < RR; ++p) st.muR[p] = mn[O + p];
  }
}

// ---------------------------------------------------------------- checkpoint record of the system kernels
// [P (lower triangle) | Sigma (lower triangle) | NTR x (dO, muR)] per (checkpoint, system); record c = state BEFORE step
// c * CK, record nck = the final state (after step T - 1)
template <int NB, int M, int NTR>
struct CkRec {
  static constexpr int P_OFF = 0, S_OFF = NB * (NB + 1) / 2, T_OFF = S_OFF + M * (M + 1) / 2, W = T_OFF + NTR * M;
};

template <typename PAT, int NX, int NB, int NU, int NY, bool DENSE_P>
struct Masks {
  static constexpr auto PM = kalman_state_mask<PAT, NB, NY, DENSE_P>();
  static constexpr auto FJ = joint_dynamics_mask<PAT, NX, NB, NU, NY, DENSE_P>();
};

// ================================================================= phase 1: forward system sweep (value, checkpoints, operators)
// NTR >= 1: the NTR trials of each system are swept in-lane (value written here); NTR == 0: the per-step trial operators go
// to the operator stream (lqg_kernels.hpp TrialOps) for k_asp_trial_fwd / k_asp_trial_rev.
template <typename R, int NX, int NB, int NU, int NY, int ND, typename PAT, int NTR, bool DENSE_P, int CK>
__global__ void __launch_bounds__(LQG_BLOCK, (sizeof(R) == 4 && NX + NB <= 5 ? LQG_ASP_FWD_WAVES_F32 : 1)) k_asp_sys_fwd(const AspArgs<R> A) {
  constexpr int M = NX + NB, O = ND, RR = M - ND;
  constexpr int NT = NTR > 0 ? NTR : 1;
  using Ops = TrialOps<M, ND>;
  using Rec = CkRec<NB, M, NTR>;
  using MK = Masks<PAT, NX, NB, NU, NY, DENSE_P>;
  const ForwardArgs<R>& a = A.f;
  const long s = blockIdx.x * (long)LQG_BLOCK + threadIdx.x;
  if (s >= a.n_sys) return;
  SysConst<R, NX, NB, NU, NY, PAT> c;
  c.load(a, s);
  const auto rQ = load_sym_masked<R, NB, PAT::Q>(A.rc.Q.p + s * A.rc.Q.sb, A.rc.Q.sr, A.rc.Q.sc);
  const auto rR = load_sym_masked<R, NU, PAT::Rr>(A.rc.Rm.p + s * A.rc.Rm.sb, A.rc.Rm.sr, A.rc.Rm.sc);
  Mat<R, NB, NB, MK::PM> Pm;
  {
    R P0[NB * NB];
    if (DENSE_P && a.Sigma0.p) load_sym<R, NB>(a.Sigma0.p + s * a.Sigma0.sb, a.Sigma0.sr, a.Sigma0.sc, P0);
    else load_gram<R, NB>(a.aV.p + s * a.aV.sb, a.aV.sr, a.aV.sc, a.nva, P0);
    LQG_UNROLL for (int i = 0; i < NB * NB; ++i) if (MK::PM.b[i]) Pm.v[i] = P0[i];
  }
  R Sg[M * M];
  LQG_UNROLL for (int i = 0; i < M * M; ++i) Sg[i] = R(0);
  TrialState<R, M, ND> st[NT];
  R xprev[NT][O];
  double acc[NT];
  const R* xp = NTR > 0 ? a.x.p + s * a.x.sb : nullptr;
  LQG_UNROLL for (int n = 0; n < NT; ++n) {
    acc[n] = 0.0;
    LQG_UNROLL for (int i = 0; i < O; ++i) { st[n].dO[i] = R(0); xprev[n][i] = NTR > 0 ? xp[n * a.x.sn + i * a.x.sd] : R(0); }
    LQG_UNROLL for (int i = 0; i < RR; ++i) st[n].muR[i] = R(0);
  }
  const R kLogNorm = R(0.5 * ND * 1.8378770664093453);
  auto keep = [&](int rec) LQG_LAMBDA_INLINE {
    R* dst = A.ck + (long)rec * Rec::W * a.ldb + s;
    R Pd[NB * NB];
    to_dense(Pm, Pd);
    store_tri_arr<R, NB>(dst + Rec::P_OFF * a.ldb, a.ldb, Pd);
    store_tri_arr<R, M>(dst + Rec::S_OFF * a.ldb, a.ldb, Sg);
    if constexpr (NTR > 0) {
      LQG_UNROLL for (int n = 0; n < NT; ++n) {
        LQG_UNROLL for (int i = 0; i < O; ++i) dst[(Rec::T_OFF + n * M + i) * a.ldb] = st[n].dO[i];
        LQG_UNROLL for (int i = 0; i < RR; ++i) dst[(Rec::T_OFF + n * M + O + i) * a.ldb] = st[n].muR[i];
      }
    }
  };
  R Lbuf[CK][NU * NB];
  // Whole chunks run WITHOUT per-step `t < T` tests (as k_forward_sp): one basic block per chunk, so
  // that the chunk's loads are scheduled ahead of the arithmetic that hides them; only the last, partial chunk is guarded.
  auto chunk = [&]<bool WHOLE>(const int t0) LQG_LAMBDA_INLINE {      // WHOLE: CK full steps, none of them step 0
    refill_gains<R, NB, NU, CK, WHOLE>(A.rc, s, t0, c.Aa, c.Ba, rQ, rR, Lbuf);
    keep(t0 / CK);
    LQG_UNROLL for (int j = 0; j < CK; ++j) {
      const int t = t0 + j;
      if (WHOLE || t < a.T) {
        Mat<R, NU, NB> L;
        LQG_UNROLL for (int e = 0; e < NU * NB; ++e) L.v[e] = Lbuf[j][e];
        sys_step<ND>(c, Pm, Sg, L, !WHOLE && t == 0, [&](const auto& AP, const auto& Pp, const auto& FP, const auto& Gi, const auto& K,
                                                const auto& Pn, const auto& BK, const auto& Fj, const auto& KN2, const auto& KN3,
                                                const auto& GG, const R (&Li)[O * O], const R (&dinv)[O], const R (&U2)[RR * O],
                                                const auto& C, const auto& F2, const auto& F2C) LQG_LAMBDA_INLINE {
          R pd = dinv[0];
          LQG_UNROLL for (int i = 1; i < O; ++i) pd *= dinv[i];
          if constexpr (NTR > 0) {
            // (k_forward_sp's scoring — one v_log_f32 per step, partial sums flushed per chunk — was measured here and not kept:
            // 1.43 -> 2.10 ms per 2^18 systems)
            const R lpd = log_<R>(pd);
            LQG_UNROLL for (int n = 0; n < NT; ++n) {
              R xt[O], w[O], cv[M];
              const R* xr = xp + n * a.x.sn + (long)t * a.x.st;
              LQG_UNROLL for (int i = 0; i < O; ++i) xt[i] = xr[i * a.x.sd];
              trial_forward<R, M, ND>(Fj, Li, U2, xt, xprev[n], st[n], w, cv, true);
              R zz = R(0);
              LQG_UNROLL for (int i = 0; i < O; ++i) zz += w[i] * w[i];
              if (WHOLE || t > 0) acc[n] += (double)(lpd - R(0.5) * zz - kLogNorm);
              LQG_UNROLL for (int i = 0; i < O; ++i) xprev[n][i] = xt[i];
            }
          } else {
            R* op = a.ops + ((long)s * (a.T + 1) + t) * Ops::N;
            const auto FjD = block2x2(restrict_to<PAT::AdmI>(minus_identity(c.Ad)), mul(c.Bd, L), mul(K, c.FAd),
                                      add(sub(minus_identity(c.Aa), mul(K, c.FAa)), mul(BK, L)));
            store_dense<0>(FjD, op + Ops::F_OFF);
            LQG_UNROLL for (int i = 0; i < RR * O; ++i) op[Ops::U_OFF + i] = U2[i];
            int e = 0;
            LQG_UNROLL for (int i = 0; i < O; ++i)
              LQG_UNROLL for (int jj = 0; jj <= i; ++jj) op[Ops::L_OFF + (e++)] = Li[i * O + jj];
            op[Ops::H_OFF] = -log_<R>(pd) + kLogNorm;
          }
          to_dense(mul_nt_sym_add(F2C, F2, GG), Sg);                    // Sigma' = F2 C F2' + G G'   system.py:223-230
          assign_state(Pm, Pn);
          (void)AP; (void)Pp; (void)FP; (void)Gi; (void)KN2; (void)KN3; (void)C;
        });
      }
    }
  };
  {
    chunk.template operator()<false>(0);                                // (step 0 initialises Sigma and is not scored)
    int t0 = CK;
    for (; t0 + CK <= a.T; t0 += CK) chunk.template operator()<true>(t0);
    if (t0 < a.T) chunk.template operator()<false>(t0);
  }
  keep((a.T + CK - 1) / CK);                                            // the final state
  // the score of the last row x_T
  {
    R Soo[O * O], Lc[O * O], dinv[O], Li[O * O];
    LQG_UNROLL for (int i = 0; i < O; ++i)
      LQG_UNROLL for (int j = 0; j < O; ++j) Soo[i * O + j] = Sg[i * M + j];
    chol_lower<R, O>(Soo, Lc, dinv);
    tri_inverse_lower<R, O>(Lc, dinv, Li);
    R pd = dinv[0];
    LQG_UNROLL for (int i = 1; i < O; ++i) pd *= dinv[i];
    if constexpr (NTR > 0) {
      const R lpd = log_<R>(pd);
      LQG_UNROLL for (int n = 0; n < NT; ++n) {
        const R* xr = xp + n * a.x.sn + (long)a.T * a.x.st;
        R zz = R(0);
        LQG_UNROLL for (int i = 0; i < O; ++i) {
          R v = R(0);
          LQG_UNROLL for (int j = 0; j <= i; ++j) v += Li[i * O + j] * ((xr[j * a.x.sd] - xprev[n][j]) - st[n].dO[j]);
          zz += v * v;
        }
        acc[n] += (double)(lpd - R(0.5) * zz - kLogNorm);
        if (a.ll) a.ll[s * a.ll_sb + n * A.ll_sn] = (R)acc[n];
      }
    } else {
      R* op = a.ops + ((long)s * (a.T + 1) + a.T) * Ops::N;
      int e = 0;
      LQG_UNROLL for (int i = 0; i < O; ++i)
        LQG_UNROLL for (int jj = 0; jj <= i; ++jj) op[Ops::L_OFF + (e++)] = Li[i * O + jj];
      op[Ops::H_OFF] = -log_<R>(pd) + kLogNorm;
    }
  }
}

// Which reverse system sweep run_asp launches.  0 (default): ONE kernel, k_asp_sys_rev_fused — round 5's, 438 VGPRs at the headline
// shape, one wave per SIMD.  1: the round-6 cut, k_asp_sys_rev + k_asp_kal_rev — 256 VGPRs, the chunk's states in LDS, two waves
// per SIMD.  Measured on one box, alternating, 2^18 headline systems x 2 in-lane trials (profiles/r06_rev_split.txt): fused 3.72 ms;
// cut 3.94 + 0.47 ms (two waves per SIMD reached, but every wave then waits 59 % of its cycles on the chunk-start loads that one
// 438-register wave overlaps with its own arithmetic); cut with a register stack at one wave 3.57 + 0.47.  The cut does not pay.
#ifndef LQG_ASP_SPLIT_KAL
#define LQG_ASP_SPLIT_KAL 0
#endif
#ifndef LQG_ASP_REV_XWINDOW
#define LQG_ASP_REV_XWINDOW 1     // the reverse system sweeps carry the in-lane trials' data rows in a rolling window (profiles/r06_rev_split.txt)
#endif
#ifndef LQG_ASP_REV_PREFETCH
#define LQG_ASP_REV_PREFETCH 0    // 1: the fused sweep requests a chunk's checkpoint one chunk ahead — measured 3.73 -> 4.03 ms (416 registers,
                                  // the requests hold 37 of them through the walk back; profiles/r06_rev_split.txt): not kept
#endif

// ================================================================= phase 2: reverse system sweep, ONE kernel (round 5; the default)
// Per chunk (last to first): the states P_t, Sigma_t (and the in-lane trials' means) of its steps are recomputed from the
// chunk's checkpoint into registers, then the steps are differentiated backward.  NTR == 0: the trial sums come from
// k_asp_trial_rev (A.sums, A.parts partial records per step); NTR >= 1: formed in-lane.
template <typename R, int NX, int NB, int NU, int NY, int ND, typename PAT, int NTR, bool DENSE_P, int CK>
__global__ void __launch_bounds__(LQG_BLOCK, 1) k_asp_sys_rev_fused(const AspArgs<R> A) {
  constexpr int M = NX + NB, O = ND, RR = M - ND;
  constexpr int NT = NTR > 0 ? NTR : 1;
  using Rec = CkRec<NB, M, NTR>;
  using MK = Masks<PAT, NX, NB, NU, NY, DENSE_P>;
  using SM = Sums<M, ND, MK::FJ>;
  using Lay = adj::Layout<NX, NB, NU, NY>;
  constexpr int NSB = NB * (NB + 1) / 2, NSM = M * (M + 1) / 2;
  const ForwardArgs<R>& a = A.f;
  const long s = blockIdx.x * (long)LQG_BLOCK + threadIdx.x;
  if (s >= a.n_sys) return;
  SysConst<R, NX, NB, NU, NY, PAT> c;
  c.load(a, s);
  const auto rQ = load_sym_masked<R, NB, PAT::Q>(A.rc.Q.p + s * A.rc.Q.sb, A.rc.Q.sr, A.rc.Q.sc);
  const auto rR = load_sym_masked<R, NU, PAT::Rr>(A.rc.Rm.p + s * A.rc.Rm.sb, A.rc.Rm.sr, A.rc.Rm.sc);
  const R* xp = NTR > 0 ? a.x.p + s * a.x.sb : nullptr;
  R gw[NT];
  LQG_UNROLL for (int n = 0; n < NT; ++n) gw[n] = (NTR > 0 && A.g) ? A.g[s * A.g_sb + n * A.g_sn] : R(1);

  // accumulated bars (time-invariant specs: one per matrix), on the masks of their primals
  Mat<R, NX, NX, PAT::Ad> bAd;  Mat<R, NX, NU, PAT::Bd> bBd;  Mat<R, NB, NB, PAT::Aa> bAa;  Mat<R, NB, NU, PAT::Ba> bBa;
  Mat<R, NY, NB, PAT::Fa> bFa;  Mat<R, NB, NB, PAT::VVa> bVVa;  Mat<R, NY, NY, PAT::WWa> bWWa;  Mat<R, NX, NX, PAT::N1> bN1;
  decltype(c.FAa) bFAa;  decltype(c.FAd) bFAd;  decltype(c.DB) bDB;  decltype(c.N2) bN2;  decltype(c.N3) bN3;
  set_zero(bAd); set_zero(bBd); set_zero(bAa); set_zero(bBa); set_zero(bFa); set_zero(bVVa); set_zero(bWWa); set_zero(bN1);
  set_zero(bFAa); set_zero(bFAd); set_zero(bDB); set_zero(bN2); set_zero(bN3);

  R Sigb[M * M];
  LQG_UNROLL for (int i = 0; i < M * M; ++i) Sigb[i] = R(0);
  Mat<R, NB, NB, MK::PM> Pb;
  set_zero(Pb);
  R pre[NT][M];
  LQG_UNROLL for (int n = 0; n < NT; ++n)
    LQG_UNROLL for (int i = 0; i < M; ++i) pre[n][i] = R(0);

  // state AFTER the step being differentiated: the observed block of Sigma_{t+1} and the trials' means at t + 1
  R SooN[O * O];
  TrialState<R, M, ND> stN[NT];
  const int nck = (a.T + CK - 1) / CK;
  {
    const R* src = A.ck + (long)nck * Rec::W * a.ldb + s;
    R Sf[M * M];
    load_tri_arr<R, M>(src + Rec::S_OFF * a.ldb, a.ldb, Sf);
    LQG_UNROLL for (int i = 0; i < O; ++i)
      LQG_UNROLL for (int j = 0; j < O; ++j) SooN[i * O + j] = Sf[i * M + j];
    LQG_UNROLL for (int n = 0; n < NT; ++n) {
      LQG_UNROLL for (int i = 0; i < O; ++i) stN[n].dO[i] = NTR > 0 ? src[(Rec::T_OFF + n * M + i) * a.ldb] : R(0);
      LQG_UNROLL for (int i = 0; i < RR; ++i) stN[n].muR[i] = NTR > 0 ? src[(Rec::T_OFF + n * M + O + i) * a.ldb] : R(0);
    }
  }

  R gs_stream = R(0);
  if constexpr (NTR == 0) {
    for (int part = 0; part < A.parts; ++part) gs_stream += A.gsum[(long)part * a.n_sys + s];
  }
#if LQG_ASP_REV_XWINDOW
  // the in-lane trials' data rows t - 1, t, t + 1 of the step being differentiated, carried backward through the whole sweep
  // (each step needs ONE row it has not seen: requested a step ahead instead of three rows at the point of use)
  [[maybe_unused]] R xwin[NT][3][O];
  if constexpr (NTR > 0) {
    LQG_UNROLL for (int n = 0; n < NT; ++n)
      LQG_UNROLL for (int q = 0; q < 3; ++q) {
        const int row = a.T - 2 + q > 0 ? a.T - 2 + q : 0;
        LQG_UNROLL for (int i = 0; i < O; ++i) xwin[n][q][i] = xp[n * a.x.sn + (long)row * a.x.st + i * a.x.sd];
      }
  }
#endif
  R Lbuf[CK][NU * NB];
  // register stack of the chunk's states (static indices only).  LQG_ASP_STACK0 = 0: the state before the chunk's FIRST step, its
  // checkpoint, is not stacked but read again when the walk back reaches it
  constexpr int S0 = 1;                                  // first stacked slot
  constexpr int CKS = CK - S0 > 0 ? CK - S0 : 1;
  R Pst[CKS][NSB], Sst[CKS][NSM];
  TrialState<R, M, ND> Tst[CKS][NT];
#if LQG_ASP_REV_PREFETCH
  // The chunk's checkpoint — the Riccati cost-to-go S and the record (P, Sigma, trial means): NS + Rec::W reals — is REQUESTED while
  // the chunk after it (in time) is still being differentiated and consumed when this chunk starts: one 438-register wave per SIMD
  // has nobody to hide the chunk-start loads behind but itself (round 6: the sweep waited 26 % of its cycles).
  constexpr int NSR = NB * (NB + 1) / 2;
  R pfS[NSR], pfR[Rec::W];
  auto prefetch = [&](const int t0n) LQG_LAMBDA_INLINE {
    const R* ssrc = A.rc.Ls + (long)(t0n / CK) * NSR * A.rc.ldb + s;
    LQG_UNROLL for (int e = 0; e < NSR; ++e) pfS[e] = ssrc[e * A.rc.ldb];
    const R* rsrc = A.ck + (long)(t0n / CK) * Rec::W * a.ldb + s;
    LQG_UNROLL for (int e = 0; e < Rec::W; ++e) pfR[e] = rsrc[e * a.ldb];
  };
  prefetch((nck - 1) * CK);
#endif
  auto chunk = [&]<bool WHOLE>(const int t0) LQG_LAMBDA_INLINE {      // WHOLE: CK full steps, none of them step 0
#if LQG_ASP_REV_PREFETCH
    {   // refill_gains from the prefetched S_{t0 + CK} (packed upper triangle by rows, k_riccati_sp<CK>)
      R Sr[NB * NB];
      int e = 0;
      LQG_UNROLL for (int i = 0; i < NB; ++i)
        LQG_UNROLL for (int j = i; j < NB; ++j) { const R v = pfS[e++]; Sr[i * NB + j] = v; Sr[j * NB + i] = v; }
      LQG_UNROLL for (int j = CK - 1; j >= 0; --j)
        if (WHOLE || t0 + j < A.rc.T) riccati_step_sp<R, NB, NU>(Sr, c.Aa, c.Ba, rQ, rR, A.rc.eps, Lbuf[j]);
    }
#else
    refill_gains<R, NB, NU, CK, WHOLE>(A.rc, s, t0, c.Aa, c.Ba, rQ, rR, Lbuf);
#endif
    // ---- recompute: states before the steps t0 .. t0 + CK - 1
    {
      Mat<R, NB, NB, MK::PM> Pm;
      R Sg[M * M];
      TrialState<R, M, ND> st[NT];
      R xprev[NT][O];
      const R* src = A.ck + (long)(t0 / CK) * Rec::W * a.ldb + s;
      {
#if LQG_ASP_REV_PREFETCH
        {
          int e = 0;
          LQG_UNROLL for (int i = 0; i < NB; ++i)
            LQG_UNROLL for (int k = 0; k <= i; ++k) {
              const R v = pfR[Rec::P_OFF + (e++)];
              if (MK::PM.b[i * NB + k]) Pm.v[i * NB + k] = v;
              if (MK::PM.b[k * NB + i]) Pm.v[k * NB + i] = v;
            }
          e = 0;
          LQG_UNROLL for (int i = 0; i < M; ++i)
            LQG_UNROLL for (int k = 0; k <= i; ++k) { const R v = pfR[Rec::S_OFF + (e++)]; Sg[i * M + k] = v; Sg[k * M + i] = v; }
        }
#else
        R Pd[NB * NB];
        load_tri_arr<R, NB>(src + Rec::P_OFF * a.ldb, a.ldb, Pd);
        LQG_UNROLL for (int i = 0; i < NB * NB; ++i) if (MK::PM.b[i]) Pm.v[i] = Pd[i];
        load_tri_arr<R, M>(src + Rec::S_OFF * a.ldb, a.ldb, Sg);
#endif
        LQG_UNROLL for (int n = 0; n < NT; ++n) {
#if LQG_ASP_REV_PREFETCH
          LQG_UNROLL for (int i = 0; i < O; ++i) st[n].dO[i] = NTR > 0 ? pfR[NTR > 0 ? Rec::T_OFF + n * M + i : 0] : R(0);
          LQG_UNROLL for (int i = 0; i < RR; ++i) st[n].muR[i] = NTR > 0 ? pfR[NTR > 0 ? Rec::T_OFF + n * M + O + i : 0] : R(0);
#else
          LQG_UNROLL for (int i = 0; i < O; ++i) st[n].dO[i] = NTR > 0 ? src[(Rec::T_OFF + n * M + i) * a.ldb] : R(0);
          LQG_UNROLL for (int i = 0; i < RR; ++i) st[n].muR[i] = NTR > 0 ? src[(Rec::T_OFF + n * M + O + i) * a.ldb] : R(0);
#endif
          if constexpr (NTR > 0) {
            const R* xr = xp + n * a.x.sn + (long)(t0 > 0 ? t0 - 1 : 0) * a.x.st;
            LQG_UNROLL for (int i = 0; i < O; ++i) xprev[n][i] = xr[i * a.x.sd];
          }
        }
      }
      LQG_UNROLL for (int j = 0; j < CK; ++j) {
        const int t = t0 + j;
        if (WHOLE || t < a.T) {
          if (j >= S0) {
            R Pd[NB * NB];
            to_dense(Pm, Pd);
            int e = 0;
            LQG_UNROLL for (int i = 0; i < NB; ++i)
              LQG_UNROLL for (int k = 0; k <= i; ++k) Pst[j >= S0 ? j - S0 : 0][e++] = Pd[i * NB + k];
            e = 0;
            LQG_UNROLL for (int i = 0; i < M; ++i)
              LQG_UNROLL for (int k = 0; k <= i; ++k) Sst[j >= S0 ? j - S0 : 0][e++] = Sg[i * M + k];
            LQG_UNROLL for (int n = 0; n < NT; ++n) Tst[j >= S0 ? j - S0 : 0][n] = st[n];
          }
          if (j + 1 < CK && (WHOLE || t + 1 < a.T)) {   // (the state after the chunk's last step is carried from the later chunk)
            Mat<R, NU, NB> L;
            LQG_UNROLL for (int e = 0; e < NU * NB; ++e) L.v[e] = Lbuf[j][e];
            sys_step<ND>(c, Pm, Sg, L, !WHOLE && t == 0, [&](const auto&, const auto&, const auto&, const auto&, const auto&, const auto& Pn,
                                                    const auto&, const auto& Fj, const auto&, const auto&, const auto& GG,
                                                    const R (&Li)[O * O], const R (&)[O], const R (&U2)[RR * O], const auto&,
                                                    const auto& F2, const auto& F2C) LQG_LAMBDA_INLINE {
              if constexpr (NTR > 0) {
                LQG_UNROLL for (int n = 0; n < NT; ++n) {
                  R xt[O], w[O], cv[M];
                  const R* xr = xp + n * a.x.sn + (long)t * a.x.st;
                  LQG_UNROLL for (int i = 0; i < O; ++i) xt[i] = xr[i * a.x.sd];
                  trial_forward<R, M, ND>(Fj, Li, U2, xt, xprev[n], st[n], w, cv, true);
                  LQG_UNROLL for (int i = 0; i < O; ++i) xprev[n][i] = xt[i];
                }
              }
              to_dense(mul_nt_sym_add(F2C, F2, GG), Sg);
              assign_state(Pm, Pn);
            });
          }
        }
      }
    }
#if LQG_ASP_REV_PREFETCH
    if (t0 > 0) prefetch(t0 - CK);                        // consumed by the next call; in flight while this chunk is walked back
#endif
    // ---- reverse over the chunk
    LQG_UNROLL for (int j = CK - 1; j >= 0; --j) {
      const int t = t0 + j;
      if (WHOLE || t < a.T) {
        Mat<R, NB, NB, MK::PM> Pm;
        R Sg[M * M];
        TrialState<R, M, ND> st0[NT];
        if (j >= S0) {
          int e = 0;
          LQG_UNROLL for (int i = 0; i < NB; ++i)
            LQG_UNROLL for (int k = 0; k <= i; ++k) {
              const R v = Pst[j >= S0 ? j - S0 : 0][e++];
              if (MK::PM.b[i * NB + k]) Pm.v[i * NB + k] = v;
              if (MK::PM.b[k * NB + i]) Pm.v[k * NB + i] = v;
            }
          e = 0;
          LQG_UNROLL for (int i = 0; i < M; ++i)
            LQG_UNROLL for (int k = 0; k <= i; ++k) { const R v = Sst[j >= S0 ? j - S0 : 0][e++]; Sg[i * M + k] = v; Sg[k * M + i] = v; }
          LQG_UNROLL for (int n = 0; n < NT; ++n) st0[n] = Tst[j >= S0 ? j - S0 : 0][n];
        } else {                                                       // the chunk's checkpoint
          const R* src = A.ck + (long)(t0 / CK) * Rec::W * a.ldb + s;
          R Pd[NB * NB];
          load_tri_arr<R, NB>(src + Rec::P_OFF * a.ldb, a.ldb, Pd);
          LQG_UNROLL for (int i = 0; i < NB * NB; ++i) if (MK::PM.b[i]) Pm.v[i] = Pd[i];
          load_tri_arr<R, M>(src + Rec::S_OFF * a.ldb, a.ldb, Sg);
          LQG_UNROLL for (int n = 0; n < NT; ++n) {
            LQG_UNROLL for (int i = 0; i < O; ++i) st0[n].dO[i] = NTR > 0 ? src[(Rec::T_OFF + n * M + i) * a.ldb] : R(0);
            LQG_UNROLL for (int i = 0; i < RR; ++i) st0[n].muR[i] = NTR > 0 ? src[(Rec::T_OFF + n * M + O + i) * a.ldb] : R(0);
          }
        }
        Mat<R, NU, NB> L;
        LQG_UNROLL for (int e = 0; e < NU * NB; ++e) L.v[e] = Lbuf[j][e];
        sys_step<ND>(c, Pm, Sg, L, !WHOLE && t == 0, [&](const auto& AP, const auto& Pp, const auto& FP, const auto& Gi, const auto& K,
                                                const auto& Pn, const auto& BK, const auto& Fj, const auto& KN2, const auto& KN3,
                                                const auto& GG, const R (&Li)[O * O], const R (&dinv)[O], const R (&U2)[RR * O],
                                                const auto& C, const auto& F2, const auto& F2C) LQG_LAMBDA_INLINE {
          (void)AP; (void)Pn; (void)KN2; (void)GG; (void)dinv; (void)C;
          // Li1 = chol(Sigma_{t+1}[:o, :o])^-1, Ni1 = Li1' Li1
          R Li1[O * O];
          {
            R Lc[O * O], d1[O];
            chol_lower<R, O>(SooN, Lc, d1);
            tri_inverse_lower<R, O>(Lc, d1, Li1);
          }
          R Wm[RR * O];                                   // Wm = U2 Li = S_ro S_oo^-1
          LQG_UNROLL for (int p = 0; p < RR; ++p)
            LQG_UNROLL for (int jj = 0; jj < O; ++jj) {
              R v = R(0);
              LQG_UNROLL for (int k = jj; k < O; ++k) v += U2[p * O + k] * Li[k * O + jj];
              Wm[p * O + jj] = v;
            }
          // ---- the trial sums of this step
          R gs = R(0), W2[O * O], CA[RR * O];
          Mat<R, M, M, MK::FJ> MC;
          if constexpr (NTR > 0) {
            LQG_UNROLL for (int i = 0; i < O * O; ++i) W2[i] = R(0);
            LQG_UNROLL for (int i = 0; i < RR * O; ++i) CA[i] = R(0);
            set_zero(MC);
            LQG_UNROLL for (int n = 0; n < NT; ++n) {
              R xm1[O], xt[O], x1[O], w0[O], w1[O], cv[M], a0[O], a1[O];
#if LQG_ASP_REV_XWINDOW
              // rows t - 1, t, t + 1 from the rolling window; the ONE new row the step before this one needs (t - 2) is requested
              // now and shifted in at the end of the step: its latency hides behind this step's arithmetic
              R xnew[O];
              {
                const R* xr = xp + n * a.x.sn + (long)(t > 1 ? t - 2 : 0) * a.x.st;
                LQG_UNROLL for (int i = 0; i < O; ++i) xnew[i] = xr[i * a.x.sd];
              }
              LQG_UNROLL for (int i = 0; i < O; ++i) { xm1[i] = xwin[n][0][i]; xt[i] = xwin[n][1][i]; x1[i] = xwin[n][2][i]; }
              LQG_UNROLL for (int i = 0; i < O; ++i) { xwin[n][2][i] = xt[i]; xwin[n][1][i] = xm1[i]; xwin[n][0][i] = xnew[i]; }
#else
              const R* xr = xp + n * a.x.sn;
              LQG_UNROLL for (int i = 0; i < O; ++i) {
                xt[i] = xr[(long)t * a.x.st + i * a.x.sd];
                x1[i] = xr[(long)(t + 1) * a.x.st + i * a.x.sd];
                xm1[i] = xr[(long)(t > 0 ? t - 1 : 0) * a.x.st + i * a.x.sd];
              }
#endif
              TrialState<R, M, ND> s0 = st0[n];
              trial_forward<R, M, ND>(Fj, Li, U2, xt, xm1, s0, w0, cv, false);
              LQG_UNROLL for (int i = 0; i < O; ++i) {
                R v = R(0);
                LQG_UNROLL for (int k = 0; k <= i; ++k) v += Li1[i * O + k] * ((x1[k] - xt[k]) - stN[n].dO[k]);
                w1[i] = v;
              }
              LQG_UNROLL for (int i = 0; i < O; ++i) {
                R v0 = R(0), v1 = R(0);
                LQG_UNROLL for (int k = i; k < O; ++k) { v0 += Li[k * O + i] * w0[k]; v1 += Li1[k * O + i] * w1[k]; }
                a0[i] = v0;
                a1[i] = v1;
              }
              const R g = gw[n];
              gs += g;
              R post[M];
              LQG_UNROLL for (int i = 0; i < M; ++i) post[i] = pre[n][i] + (i < O ? g * a1[i < O ? i : 0] : R(0));
              LQG_UNROLL for (int i = 0; i < O; ++i)
                LQG_UNROLL for (int k = 0; k < O; ++k) W2[i * O + k] += g * a1[i] * a1[k];
              LQG_UNROLL for (int i = 0; i < M; ++i)
                LQG_UNROLL for (int k = 0; k < M; ++k) if (MK::FJ.b[i * M + k]) MC.v[i * M + k] += post[i] * cv[k];
              R ch[RR];
              LQG_UNROLL for (int p = 0; p < RR; ++p) {
                R v = R(0);
                LQG_UNROLL for (int i = 0; i < M; ++i) if (std::remove_cvref_t<decltype(F2)>::mask.b[i * RR + p]) v += F2.v[i * RR + p] * post[i];
                ch[p] = v;
              }
              LQG_UNROLL for (int p = 0; p < RR; ++p)
                LQG_UNROLL for (int k = 0; k < O; ++k) CA[p * O + k] += ch[p] * a0[k];
              LQG_UNROLL for (int k = 0; k < O; ++k) {
                R v = R(0);
                LQG_UNROLL for (int p = 0; p < RR; ++p) v += Wm[p * O + k] * ch[p];
                pre[n][k] = -v;
              }
              LQG_UNROLL for (int p = 0; p < RR; ++p) pre[n][O + p] = ch[p];
              stN[n] = st0[n];
            }
          } else {
            LQG_UNROLL for (int i = 0; i < O * O; ++i) W2[i] = R(0);
            LQG_UNROLL for (int i = 0; i < RR * O; ++i) CA[i] = R(0);
            set_zero(MC);
            gs = gs_stream;
            for (int part = 0; part < A.parts; ++part) {
              const R* sm = A.sums + (((long)part * a.n_sys + s) * a.T + t) * SM::N;
              int e = 0;
              LQG_UNROLL for (int i = 0; i < O; ++i)
                LQG_UNROLL for (int k = 0; k <= i; ++k) {
                  const R v = sm[SM::W_OFF + (e++)];
                  W2[i * O + k] += v;
                  if (k != i) W2[k * O + i] += v;
                }
              LQG_UNROLL for (int i = 0; i < RR * O; ++i) CA[i] += sm[SM::C_OFF + i];
              LQG_UNROLL for (int i = 0; i < M; ++i)
                LQG_UNROLL for (int k = 0; k < M; ++k) if (MK::FJ.b[i * M + k]) MC.v[i * M + k] += sm[SM::mc(i, k)];
            }
          }
          // ---- log-density of x_{t+1}, all trials                                        system.py:244-248
          LQG_UNROLL for (int i = 0; i < O; ++i)
            LQG_UNROLL for (int k = 0; k < O; ++k) {
              R ni = R(0);
              LQG_UNROLL for (int q = (i > k ? i : k); q < O; ++q) ni += Li1[q * O + i] * Li1[q * O + k];
              Sigb[i * M + k] += R(0.5) * (W2[i * O + k] - gs * ni);
            }
          // ---- Sigma_{t+1} = F2 C F2' + GG,  mu_{t+1} = Fj c
          const auto SbM = from_dense<R, M, M>(Sigb);
          const auto FbR = mul(SbM, F2C);                                                 // [M, RR]
          Mat<R, M, M, MK::FJ> Fb;
          LQG_UNROLL for (int i = 0; i < M; ++i)
            LQG_UNROLL for (int k = 0; k < M; ++k)
              if (MK::FJ.b[i * M + k]) {
                R v = MC.v[i * M + k];
                if (k >= O) { if (decltype(FbR)::mask.b[i * RR + (k >= O ? k - O : 0)]) v += R(2) * FbR.v[i * RR + (k >= O ? k - O : 0)]; }
                Fb.v[i * M + k] = v;
              }
          auto G11 = blk<0, 0, NX, NX>(SbM);
          auto G21 = blk<NX, 0, NB, NX>(SbM);
          auto G22 = blk<NX, NX, NB, NB>(SbM);
          // ---- conditioning on x_t, in terms of Wm and a (no product of two inverses: oracle/lqg_adjoint_np.py)
          {
            const auto SF = mul(SbM, F2);
            const auto Ch = sym_part(mul_tn(F2, SF));                                      // [RR, RR]
            R Chd[RR * RR], ChW[RR * O];
            to_dense(Ch, Chd);
            LQG_UNROLL for (int p = 0; p < RR; ++p)
              LQG_UNROLL for (int k = 0; k < O; ++k) {
                R v = R(0);
                LQG_UNROLL for (int q = 0; q < RR; ++q) v += Chd[p * RR + q] * Wm[q * O + k];
                ChW[p * O + k] = v;
              }
            LQG_UNROLL for (int i = 0; i < O; ++i)
              LQG_UNROLL for (int k = i; k < O; ++k) {
                R v = R(0);
                LQG_UNROLL for (int p = 0; p < RR; ++p)
                  v += Wm[p * O + i] * (ChW[p * O + k] - CA[p * O + k]) + Wm[p * O + k] * (ChW[p * O + i] - CA[p * O + i]);
                Sigb[i * M + k] = R(0.5) * v;
                Sigb[k * M + i] = R(0.5) * v;
              }
            LQG_UNROLL for (int p = 0; p < RR; ++p)
              LQG_UNROLL for (int k = 0; k < O; ++k) {
                const R v = R(0.5) * (CA[p * O + k] - R(2) * ChW[p * O + k]);
                Sigb[(O + p) * M + k] = v;
                Sigb[k * M + O + p] = v;
              }
            LQG_UNROLL for (int p = 0; p < RR; ++p)
              LQG_UNROLL for (int q = 0; q < RR; ++q) Sigb[(O + p) * M + O + q] = Chd[p * RR + q];
          }
          if (!WHOLE && t == 0) {                                                          // Sigma_0 = G_0 G_0'
            const auto S0 = from_dense<R, M, M>(Sigb);
            accum(G11, blk<0, 0, NX, NX>(S0));
            accum(G21, blk<NX, 0, NB, NX>(S0));
            accum(G22, blk<NX, NX, NB, NB>(S0));
          }
          // ---- joint system -> Kbar, Lbar and the accumulated bars                      system.py:167-207
          const auto F11 = blk<0, 0, NX, NX>(Fb);
          const auto F12 = blk<0, NX, NX, NB>(Fb);
          const auto F21 = blk<NX, 0, NB, NX>(Fb);
          const auto F22 = blk<NX, NX, NB, NB>(Fb);
          const auto BKb = mul_nt(F22, L);                                                 // [NB, NU]
          Mat<R, NB, NY, std::remove_cvref_t<decltype(K)>::mask> Kb;
          set_zero(Kb);
          accum(Kb, mul_nt(F21, c.FAd));
          accum(Kb, mul_nt(F22, c.FAa), R(-1));
          accum(Kb, mul_nt(BKb, c.DB));
          accum(Kb, mul_nt(G21, c.N2), R(2));
          accum(Kb, mul(G22, KN3), R(2));
          {
            const auto Lb = add(mul_tn(c.Bd, F12), mul_tn(BK, F22));                       // [NU, NB]
            R* dst = A.Lbar + (long)t * (NU * NB) * a.ldb + s;
            store_col(dst, a.ldb, Lb);
          }
          accum(bAd, F11);
          accum(bBd, mul_nt(F12, L));
          accum(bFAd, mul_tn(K, F21));
          accum(bAa, F22);
          accum(bFAa, mul_tn(K, F22), R(-1));
          accum(bBa, BKb);
          accum(bDB, mul_tn(K, BKb));
          accum(bN1, G11);
          accum(bN2, mul_tn(K, G21), R(2));
          accum(bN3, mul_tn(K, mul(G22, K)));
          // ---- Kalman step                                                              kf.py:10-14
          accum(Kb, mul_nt(Pb, FP), R(-1));
          const auto KbGi = mul(Kb, Gi);                                                   // [NB, NY]
          const auto Gmb = scaled(mul(Gi, mul(FP, KbGi)), R(-1));                          // [NY, NY]
          const auto FPb = add(sub(transpose(KbGi), mul_tn(K, Pb)), mul(Gmb, c.Fa));       // [NY, NB]
          accum(bFa, mul(FPb, Pp));
          accum(bFa, mul_tn(Gmb, FP));
          accum(bWWa, Gmb);
          const auto Ppb = sym_part(add(Pb, mul_tn(c.Fa, FPb)));
          accum(bVVa, Ppb);
          const auto PA = mul(Ppb, c.Aa);
          accum(bAa, mul(PA, Pm), R(2));
          {
            const auto Pb1 = sym_part(mul_tn(c.Aa, PA));
            LQG_UNROLL for (int i = 0; i < NB * NB; ++i)
              if (MK::PM.b[i]) Pb.v[i] = decltype(Pb1)::mask.b[i] ? Pb1.v[i] : R(0);
          }
          // the state after step t - 1 is the state before step t
          LQG_UNROLL for (int i = 0; i < O; ++i)
            LQG_UNROLL for (int k = 0; k < O; ++k) SooN[i * O + k] = Sg[i * M + k];
        });
      }
    }
  };
  {
    int t0 = (nck - 1) * CK;
    if (t0 + CK > a.T && t0 > 0) { chunk.template operator()<false>(t0); t0 -= CK; }     // the last, partial chunk
    for (; t0 > 0; t0 -= CK) chunk.template operator()<true>(t0);
    chunk.template operator()<false>(0);                                // (step 0: Sigma_0 = G_0 G_0')
  }
  // ---- chain the hoisted products' bars to the stored matrices, write the gradient
  // FAa = Fa Aa, FAd = Fd Ad, DB = Fd Bd - Fa Ba, N2 = Fd N1, N3 = Fd N1 Fd' + WWd
  {
    R* o = A.out + s;
    const long ld = A.ld;
    accum(bAd, mul_tn(c.Fd, bFAd));
    accum(bBd, mul_tn(c.Fd, bDB));
    accum(bAa, mul_tn(c.Fa, bFAa));
    accum(bBa, mul_tn(c.Fa, bDB), R(-1));
    accum(bFa, mul_nt(bFAa, c.Aa));
    accum(bFa, mul_nt(bDB, c.Ba), R(-1));
    Mat<R, NY, NX, PAT::Fd> bFd;
    set_zero(bFd);
    accum(bFd, mul_nt(bFAd, c.Ad));
    accum(bFd, mul_nt(bDB, c.Bd));
    accum(bFd, mul(bN2, c.N1));
    accum(bFd, mul(sym_part(bN3), mul(c.Fd, c.N1)), R(2));
    accum(bN1, mul_tn(c.Fd, bN2));
    accum(bN1, mul_tn(c.Fd, mul(bN3, c.Fd)));
    if (!(DENSE_P && a.Sigma0.p)) accum(bVVa, Pb);                                         // default Sigma0 = V_0 V_0'  system.py:160
    // bars of fields that no parameter moves (PAT::live_*: lqg_amd/specialize.py) are written as zeros — everything that only
    // feeds them (their accumulators, the hoisted products' bars, the F11 block of MC, ...) is dead code and compiled out
    auto put = [&]<bool LIVE>(int off, const auto& m) LQG_LAMBDA_INLINE {
      if constexpr (LIVE) store_col(o + off * ld, ld, m);
      else {
        using MT = std::remove_cvref_t<decltype(m)>;
        LQG_UNROLL for (int i = 0; i < MT::rows * MT::cols; ++i) o[(off + i) * ld] = R(0);
      }
    };
    put.template operator()<PAT::live_Ad>(Lay::DA, bAd);
    put.template operator()<PAT::live_Bd>(Lay::DB, bBd);
    put.template operator()<PAT::live_Fd>(Lay::DF, bFd);
    put.template operator()<PAT::live_Vd>(Lay::DVV, bN1);
    {
      Mat<R, NY, NY, PAT::WWd> bWWd;
      set_zero(bWWd);
      accum(bWWd, bN3);
      put.template operator()<PAT::live_Wd>(Lay::DWW, bWWd);
    }
    put.template operator()<PAT::live_Aa>(Lay::AA, bAa);
    put.template operator()<PAT::live_Ba>(Lay::AB, bBa);
    put.template operator()<PAT::live_Fa>(Lay::AF, bFa);
    put.template operator()<PAT::live_Va>(Lay::AVV, bVVa);
    put.template operator()<PAT::live_Wa>(Lay::AWW, bWWa);
    store_col(o + Lay::AS0 * ld, ld, Pb);
  }
}

// ================================================================= phase 2: reverse system sweep, cut in two (round 6)
// Round 5 ran ALL system adjoints in one kernel: 438 VGPRs at the headline shape (two in-lane trials), one wave per SIMD, 0.35 of
// the VALU issue rate.  The adjoint of the Kalman-covariance recursion reads nothing of the moment recursion but K-bar_t — as the
// Riccati adjoint reads nothing but L-bar_t — so it is its own backward kernel now:
//   k_asp_sys_rev   Sigma-bar / mu-bar sweep: joint system, moment recursion, conditioning and the in-lane trials' adjoints; emits
//                   K-bar_t (the part that does not involve P-bar) and L-bar_t per step.  The chunk's recomputed states live in
//                   LDS ([slot][element][lane]: conflict-free, static offsets) instead of a register stack; the chunk's Kalman
//                   gains K_t (3 of 6 entries for the tracking models) stay in registers — P itself is no longer stacked.
//   k_asp_kal_rev   P-bar sweep: consumes K-bar_t, recomputes the chunk's P_t from the same checkpoints; bars of Fa, VVa, WWa,
//                   (the Kalman part of) Aa, and Sigma0.
// Per chunk (last to first): the states of its steps are recomputed from the chunk's checkpoint, then the steps are
// differentiated backward.  NTR == 0: the trial sums come from k_asp_trial_rev (A.sums, A.parts partial records per step);
// NTR >= 1: formed in-lane.

// what of the chunk's recomputed states goes to LDS: the budget is 80 reals per lane = 20 kB per 64-lane workgroup in fp32 (eight
// workgroups = two waves per SIMD fit a CU's 160 kB), 40 kB in fp64 (four workgroups: one wave per SIMD)
#ifndef LQG_ASP_REV_LDS_BUDGET
#define LQG_ASP_REV_LDS_BUDGET 80
#endif
template <int M, int NTM, int CKS>
struct RevStack {
  static constexpr int NSM = M * (M + 1) / 2;
  static constexpr int BUDGET = LQG_ASP_REV_LDS_BUDGET;
  static constexpr bool SIG = CKS * NSM <= BUDGET;                       // Sigma_t (lower triangle)
  static constexpr bool TRL = SIG && NTM > 0 && CKS * (NSM + NTM) <= BUDGET;   // the in-lane trials' mean states
  static constexpr int PER_SLOT = (SIG ? NSM : 0) + (TRL ? NTM : 0);
  static constexpr int REALS = CKS * PER_SLOT;
  static constexpr int S_OFF = 0, T_OFF = SIG ? NSM : 0;
};
#ifndef LQG_ASP_REV_WAVES_F32
#define LQG_ASP_REV_WAVES_F32 2   // waves per SIMD the fp32 Sigma-bar sweep is allocated for when its stack fits LDS
#endif
template <typename R, int M, int NTM, int CK>
constexpr int rev_waves() {
  return (sizeof(R) == 4 && RevStack<M, NTM, (CK - 1 > 0 ? CK - 1 : 1)>::SIG) ? LQG_ASP_REV_WAVES_F32 : 1;
}

template <typename R, int NX, int NB, int NU, int NY, int ND, typename PAT, int NTR, bool DENSE_P, int CK>
__global__ void __launch_bounds__(LQG_BLOCK, (rev_waves<R, NX + NB, NTR * (NX + NB), CK>())) k_asp_sys_rev(const AspArgs<R> A) {
  constexpr int M = NX + NB, O = ND, RR = M - ND;
  constexpr int NT = NTR > 0 ? NTR : 1;
  using Rec = CkRec<NB, M, NTR>;
  using MK = Masks<PAT, NX, NB, NU, NY, DENSE_P>;
  using SM = Sums<M, ND, MK::FJ>;
  using Lay = adj::Layout<NX, NB, NU, NY>;
  using KT = KalmanGain<R, NX, NB, NU, NY, PAT, MK::PM>;
  constexpr int NSM = M * (M + 1) / 2;
  static_assert(LQG_ASP_STACK0 == 0, "the state before a chunk's first step is re-read from its checkpoint");
  constexpr int S0 = 1;                                  // first stacked slot
  constexpr int CKS = CK - S0 > 0 ? CK - S0 : 1;
  using ST = RevStack<M, NTR * M, CKS>;
  __shared__ R lds_stack[ST::REALS > 0 ? ST::REALS * LQG_BLOCK : 1];
  const ForwardArgs<R>& a = A.f;
  const long s = blockIdx.x * (long)LQG_BLOCK + threadIdx.x;
  if (s >= a.n_sys) return;
  R* const lds = lds_stack + threadIdx.x;               // element e of slot q: lds[(q * PER_SLOT + e) * LQG_BLOCK]
  SysConst<R, NX, NB, NU, NY, PAT> c;
  c.load(a, s);
  const auto rQ = load_sym_masked<R, NB, PAT::Q>(A.rc.Q.p + s * A.rc.Q.sb, A.rc.Q.sr, A.rc.Q.sc);
  const auto rR = load_sym_masked<R, NU, PAT::Rr>(A.rc.Rm.p + s * A.rc.Rm.sb, A.rc.Rm.sr, A.rc.Rm.sc);
  const R* xp = NTR > 0 ? a.x.p + s * a.x.sb : nullptr;
  R gw[NT];
  LQG_UNROLL for (int n = 0; n < NT; ++n) gw[n] = (NTR > 0 && A.g) ? A.g[s * A.g_sb + n * A.g_sn] : R(1);

  // accumulated bars (time-invariant specs: one per matrix), on the masks of their primals
  Mat<R, NX, NX, PAT::Ad> bAd;  Mat<R, NX, NU, PAT::Bd> bBd;  Mat<R, NB, NB, PAT::Aa> bAa;  Mat<R, NB, NU, PAT::Ba> bBa;
  Mat<R, NY, NB, PAT::Fa> bFa;  Mat<R, NX, NX, PAT::N1> bN1;
  decltype(c.FAa) bFAa;  decltype(c.FAd) bFAd;  decltype(c.DB) bDB;  decltype(c.N2) bN2;  decltype(c.N3) bN3;
  set_zero(bAd); set_zero(bBd); set_zero(bAa); set_zero(bBa); set_zero(bFa); set_zero(bN1);
  set_zero(bFAa); set_zero(bFAd); set_zero(bDB); set_zero(bN2); set_zero(bN3);

  R Sigb[M * M];
  LQG_UNROLL for (int i = 0; i < M * M; ++i) Sigb[i] = R(0);
  R pre[NT][M];
  LQG_UNROLL for (int n = 0; n < NT; ++n)
    LQG_UNROLL for (int i = 0; i < M; ++i) pre[n][i] = R(0);

  // state AFTER the step being differentiated: the observed block of Sigma_{t+1} and the trials' means at t + 1
  R SooN[O * O];
  TrialState<R, M, ND> stN[NT];
  const int nck = (a.T + CK - 1) / CK;
  {
    const R* src = A.ck + (long)nck * Rec::W * a.ldb + s;
    R Sf[M * M];
    load_tri_arr<R, M>(src + Rec::S_OFF * a.ldb, a.ldb, Sf);
    LQG_UNROLL for (int i = 0; i < O; ++i)
      LQG_UNROLL for (int j = 0; j < O; ++j) SooN[i * O + j] = Sf[i * M + j];
    LQG_UNROLL for (int n = 0; n < NT; ++n) {
      LQG_UNROLL for (int i = 0; i < O; ++i) stN[n].dO[i] = NTR > 0 ? src[(Rec::T_OFF + n * M + i) * a.ldb] : R(0);
      LQG_UNROLL for (int i = 0; i < RR; ++i) stN[n].muR[i] = NTR > 0 ? src[(Rec::T_OFF + n * M + O + i) * a.ldb] : R(0);
    }
  }

  R gs_stream = R(0);
  if constexpr (NTR == 0) {
    for (int part = 0; part < A.parts; ++part) gs_stream += A.gsum[(long)part * a.n_sys + s];
  }
#if LQG_ASP_REV_XWINDOW
  // the in-lane trials' data rows t - 1, t, t + 1 of the step being differentiated, carried backward through the whole sweep
  // (each step needs ONE row it has not seen: requested a step ahead instead of three rows at the point of use)
  [[maybe_unused]] R xwin[NT][3][O];
  if constexpr (NTR > 0) {
    LQG_UNROLL for (int n = 0; n < NT; ++n)
      LQG_UNROLL for (int q = 0; q < 3; ++q) {
        const int row = a.T - 2 + q > 0 ? a.T - 2 + q : 0;
        LQG_UNROLL for (int i = 0; i < O; ++i) xwin[n][q][i] = xp[n * a.x.sn + (long)row * a.x.st + i * a.x.sd];
      }
  }
#endif
  R Lbuf[CK][NU * NB];
  KT Kst[CK];                                            // the chunk's Kalman gains (static indices only)
  // what of the stack does not fit LDS stays in registers (static indices only)
  R Sst[ST::SIG ? 1 : CKS][ST::SIG ? 1 : NSM];
  TrialState<R, M, ND> Tst[ST::TRL ? 1 : CKS][NT];
  auto push = [&](const int q, const R (&Sg)[M * M], const TrialState<R, M, ND> (&st)[NT]) LQG_LAMBDA_INLINE {
    int e = 0;
    LQG_UNROLL for (int i = 0; i < M; ++i)
      LQG_UNROLL for (int k = 0; k <= i; ++k) {
        if constexpr (ST::SIG) lds[(q * ST::PER_SLOT + ST::S_OFF + e) * LQG_BLOCK] = Sg[i * M + k];
        else Sst[ST::SIG ? 0 : q][ST::SIG ? 0 : e] = Sg[i * M + k];
        ++e;
      }
    if constexpr (NTR > 0) {
      LQG_UNROLL for (int n = 0; n < NT; ++n) {
        if constexpr (ST::TRL) {
          LQG_UNROLL for (int i = 0; i < O; ++i) lds[(q * ST::PER_SLOT + ST::T_OFF + n * M + i) * LQG_BLOCK] = st[n].dO[i];
          LQG_UNROLL for (int i = 0; i < RR; ++i) lds[(q * ST::PER_SLOT + ST::T_OFF + n * M + O + i) * LQG_BLOCK] = st[n].muR[i];
        } else {
          Tst[ST::TRL ? 0 : q][n] = st[n];
        }
      }
    }
  };
  auto pop = [&](const int q, R (&Sg)[M * M], TrialState<R, M, ND> (&st)[NT]) LQG_LAMBDA_INLINE {
    int e = 0;
    LQG_UNROLL for (int i = 0; i < M; ++i)
      LQG_UNROLL for (int k = 0; k <= i; ++k) {
        R v;
        if constexpr (ST::SIG) v = lds[(q * ST::PER_SLOT + ST::S_OFF + e) * LQG_BLOCK];
        else v = Sst[ST::SIG ? 0 : q][ST::SIG ? 0 : e];
        ++e;
        Sg[i * M + k] = v;
        Sg[k * M + i] = v;
      }
    LQG_UNROLL for (int n = 0; n < NT; ++n) {
      if constexpr (NTR > 0 && ST::TRL) {
        LQG_UNROLL for (int i = 0; i < O; ++i) st[n].dO[i] = lds[(q * ST::PER_SLOT + ST::T_OFF + n * M + i) * LQG_BLOCK];
        LQG_UNROLL for (int i = 0; i < RR; ++i) st[n].muR[i] = lds[(q * ST::PER_SLOT + ST::T_OFF + n * M + O + i) * LQG_BLOCK];
      } else if constexpr (NTR > 0) {
        st[n] = Tst[ST::TRL ? 0 : q][n];
      } else {
        LQG_UNROLL for (int i = 0; i < O; ++i) st[n].dO[i] = R(0);
        LQG_UNROLL for (int i = 0; i < RR; ++i) st[n].muR[i] = R(0);
      }
    }
  };
  auto load_ck = [&](const int rec, R (&Sg)[M * M], TrialState<R, M, ND> (&st)[NT]) LQG_LAMBDA_INLINE {
    const R* src = A.ck + (long)rec * Rec::W * a.ldb + s;
    load_tri_arr<R, M>(src + Rec::S_OFF * a.ldb, a.ldb, Sg);
    LQG_UNROLL for (int n = 0; n < NT; ++n) {
      LQG_UNROLL for (int i = 0; i < O; ++i) st[n].dO[i] = NTR > 0 ? src[(Rec::T_OFF + n * M + i) * a.ldb] : R(0);
      LQG_UNROLL for (int i = 0; i < RR; ++i) st[n].muR[i] = NTR > 0 ? src[(Rec::T_OFF + n * M + O + i) * a.ldb] : R(0);
    }
  };
  auto chunk = [&]<bool WHOLE>(const int t0) LQG_LAMBDA_INLINE {      // WHOLE: CK full steps, none of them step 0
    refill_gains<R, NB, NU, CK, WHOLE>(A.rc, s, t0, c.Aa, c.Ba, rQ, rR, Lbuf);
    // ---- recompute: the gains K_t of the chunk's steps and the states before the steps t0 + 1 .. t0 + CK - 1
    {
      Mat<R, NB, NB, MK::PM> Pm;
      R Sg[M * M];
      TrialState<R, M, ND> st[NT];
      R xprev[NT][O];
      {
        const R* src = A.ck + (long)(t0 / CK) * Rec::W * a.ldb + s;
        R Pd[NB * NB];
        load_tri_arr<R, NB>(src + Rec::P_OFF * a.ldb, a.ldb, Pd);
        LQG_UNROLL for (int i = 0; i < NB * NB; ++i) if (MK::PM.b[i]) Pm.v[i] = Pd[i];
        load_ck(t0 / CK, Sg, st);
        if constexpr (NTR > 0) {
          LQG_UNROLL for (int n = 0; n < NT; ++n) {
            const R* xr = xp + n * a.x.sn + (long)(t0 > 0 ? t0 - 1 : 0) * a.x.st;
            LQG_UNROLL for (int i = 0; i < O; ++i) xprev[n][i] = xr[i * a.x.sd];
          }
        }
      }
      LQG_UNROLL for (int j = 0; j < CK; ++j) {
        const int t = t0 + j;
        if (WHOLE || t < a.T) {
          if (j >= S0) push(j >= S0 ? j - S0 : 0, Sg, st);
          const auto kp = kalman_part(c, Pm);
          Kst[j] = kp.K;
          if (j + 1 < CK && (WHOLE || t + 1 < a.T)) {   // (the state after the chunk's last step is carried from the later chunk)
            Mat<R, NU, NB> L;
            LQG_UNROLL for (int e = 0; e < NU * NB; ++e) L.v[e] = Lbuf[j][e];
            joint_part<ND>(c, kp.K, Sg, L, !WHOLE && t == 0, [&](const auto&, const auto& Fj, const auto&, const auto&, const auto& GG,
                                                             const R (&Li)[O * O], const R (&)[O], const R (&U2)[RR * O], const auto&,
                                                             const auto& F2, const auto& F2C) LQG_LAMBDA_INLINE {
              if constexpr (NTR > 0) {
                LQG_UNROLL for (int n = 0; n < NT; ++n) {
                  R xt[O], w[O], cv[M];
                  const R* xr = xp + n * a.x.sn + (long)t * a.x.st;
                  LQG_UNROLL for (int i = 0; i < O; ++i) xt[i] = xr[i * a.x.sd];
                  trial_forward<R, M, ND>(Fj, Li, U2, xt, xprev[n], st[n], w, cv, true);
                  LQG_UNROLL for (int i = 0; i < O; ++i) xprev[n][i] = xt[i];
                }
              }
              to_dense(mul_nt_sym_add(F2C, F2, GG), Sg);
            });
            assign_state(Pm, kp.Pn);
          }
        }
      }
    }
    // ---- reverse over the chunk
    LQG_UNROLL for (int j = CK - 1; j >= 0; --j) {
      const int t = t0 + j;
      if (WHOLE || t < a.T) {
        R Sg[M * M];
        TrialState<R, M, ND> st0[NT];
        if (j >= S0) pop(j >= S0 ? j - S0 : 0, Sg, st0);
        else load_ck(t0 / CK, Sg, st0);                                  // the chunk's checkpoint
        const KT K = Kst[j];
        Mat<R, NU, NB> L;
        LQG_UNROLL for (int e = 0; e < NU * NB; ++e) L.v[e] = Lbuf[j][e];
        joint_part<ND>(c, K, Sg, L, !WHOLE && t == 0, [&](const auto& BK, const auto& Fj, const auto& KN2, const auto& KN3,
                                                      const auto& GG, const R (&Li)[O * O], const R (&dinv)[O], const R (&U2)[RR * O],
                                                      const auto& C, const auto& F2, const auto& F2C) LQG_LAMBDA_INLINE {
          (void)KN2; (void)GG; (void)dinv; (void)C;
          // Li1 = chol(Sigma_{t+1}[:o, :o])^-1, Ni1 = Li1' Li1
          R Li1[O * O];
          {
            R Lc[O * O], d1[O];
            chol_lower<R, O>(SooN, Lc, d1);
            tri_inverse_lower<R, O>(Lc, d1, Li1);
          }
          R Wm[RR * O];                                   // Wm = U2 Li = S_ro S_oo^-1
          LQG_UNROLL for (int p = 0; p < RR; ++p)
            LQG_UNROLL for (int jj = 0; jj < O; ++jj) {
              R v = R(0);
              LQG_UNROLL for (int k = jj; k < O; ++k) v += U2[p * O + k] * Li[k * O + jj];
              Wm[p * O + jj] = v;
            }
          // ---- the trial sums of this step
          R gs = R(0), W2[O * O], CA[RR * O];
          Mat<R, M, M, MK::FJ> MC;
          if constexpr (NTR > 0) {
            LQG_UNROLL for (int i = 0; i < O * O; ++i) W2[i] = R(0);
            LQG_UNROLL for (int i = 0; i < RR * O; ++i) CA[i] = R(0);
            set_zero(MC);
            LQG_UNROLL for (int n = 0; n < NT; ++n) {
              R xm1[O], xt[O], x1[O], w0[O], w1[O], cv[M], a0[O], a1[O];
#if LQG_ASP_REV_XWINDOW
              // rows t - 1, t, t + 1 from the rolling window; the ONE new row the step before this one needs (t - 2) is requested
              // now and shifted in at the end of the step: its latency hides behind this step's arithmetic
              R xnew[O];
              {
                const R* xr = xp + n * a.x.sn + (long)(t > 1 ? t - 2 : 0) * a.x.st;
                LQG_UNROLL for (int i = 0; i < O; ++i) xnew[i] = xr[i * a.x.sd];
              }
              LQG_UNROLL for (int i = 0; i < O; ++i) { xm1[i] = xwin[n][0][i]; xt[i] = xwin[n][1][i]; x1[i] = xwin[n][2][i]; }
              LQG_UNROLL for (int i = 0; i < O; ++i) { xwin[n][2][i] = xt[i]; xwin[n][1][i] = xm1[i]; xwin[n][0][i] = xnew[i]; }
#else
              const R* xr = xp + n * a.x.sn;
              LQG_UNROLL for (int i = 0; i < O; ++i) {
                xt[i] = xr[(long)t * a.x.st + i * a.x.sd];
                x1[i] = xr[(long)(t + 1) * a.x.st + i * a.x.sd];
                xm1[i] = xr[(long)(t > 0 ? t - 1 : 0) * a.x.st + i * a.x.sd];
              }
#endif
              TrialState<R, M, ND> s0 = st0[n];
              trial_forward<R, M, ND>(Fj, Li, U2, xt, xm1, s0, w0, cv, false);
              LQG_UNROLL for (int i = 0; i < O; ++i) {
                R v = R(0);
                LQG_UNROLL for (int k = 0; k <= i; ++k) v += Li1[i * O + k] * ((x1[k] - xt[k]) - stN[n].dO[k]);
                w1[i] = v;
              }
              LQG_UNROLL for (int i = 0; i < O; ++i) {
                R v0 = R(0), v1 = R(0);
                LQG_UNROLL for (int k = i; k < O; ++k) { v0 += Li[k * O + i] * w0[k]; v1 += Li1[k * O + i] * w1[k]; }
                a0[i] = v0;
                a1[i] = v1;
              }
              const R g = gw[n];
              gs += g;
              R post[M];
              LQG_UNROLL for (int i = 0; i < M; ++i) post[i] = pre[n][i] + (i < O ? g * a1[i < O ? i : 0] : R(0));
              LQG_UNROLL for (int i = 0; i < O; ++i)
                LQG_UNROLL for (int k = 0; k < O; ++k) W2[i * O + k] += g * a1[i] * a1[k];
              LQG_UNROLL for (int i = 0; i < M; ++i)
                LQG_UNROLL for (int k = 0; k < M; ++k) if (MK::FJ.b[i * M + k]) MC.v[i * M + k] += post[i] * cv[k];
              R ch[RR];
              LQG_UNROLL for (int p = 0; p < RR; ++p) {
                R v = R(0);
                LQG_UNROLL for (int i = 0; i < M; ++i) if (std::remove_cvref_t<decltype(F2)>::mask.b[i * RR + p]) v += F2.v[i * RR + p] * post[i];
                ch[p] = v;
              }
              LQG_UNROLL for (int p = 0; p < RR; ++p)
                LQG_UNROLL for (int k = 0; k < O; ++k) CA[p * O + k] += ch[p] * a0[k];
              LQG_UNROLL for (int k = 0; k < O; ++k) {
                R v = R(0);
                LQG_UNROLL for (int p = 0; p < RR; ++p) v += Wm[p * O + k] * ch[p];
                pre[n][k] = -v;
              }
              LQG_UNROLL for (int p = 0; p < RR; ++p) pre[n][O + p] = ch[p];
              stN[n] = st0[n];
            }
          } else {
            LQG_UNROLL for (int i = 0; i < O * O; ++i) W2[i] = R(0);
            LQG_UNROLL for (int i = 0; i < RR * O; ++i) CA[i] = R(0);
            set_zero(MC);
            gs = gs_stream;
            for (int part = 0; part < A.parts; ++part) {
              const R* sm = A.sums + (((long)part * a.n_sys + s) * a.T + t) * SM::N;
              int e = 0;
              LQG_UNROLL for (int i = 0; i < O; ++i)
                LQG_UNROLL for (int k = 0; k <= i; ++k) {
                  const R v = sm[SM::W_OFF + (e++)];
                  W2[i * O + k] += v;
                  if (k != i) W2[k * O + i] += v;
                }
              LQG_UNROLL for (int i = 0; i < RR * O; ++i) CA[i] += sm[SM::C_OFF + i];
              LQG_UNROLL for (int i = 0; i < M; ++i)
                LQG_UNROLL for (int k = 0; k < M; ++k) if (MK::FJ.b[i * M + k]) MC.v[i * M + k] += sm[SM::mc(i, k)];
            }
          }
          // ---- log-density of x_{t+1}, all trials                                        system.py:244-248
          LQG_UNROLL for (int i = 0; i < O; ++i)
            LQG_UNROLL for (int k = 0; k < O; ++k) {
              R ni = R(0);
              LQG_UNROLL for (int q = (i > k ? i : k); q < O; ++q) ni += Li1[q * O + i] * Li1[q * O + k];
              Sigb[i * M + k] += R(0.5) * (W2[i * O + k] - gs * ni);
            }
          // ---- Sigma_{t+1} = F2 C F2' + GG,  mu_{t+1} = Fj c
          const auto SbM = from_dense<R, M, M>(Sigb);
          const auto FbR = mul(SbM, F2C);                                                 // [M, RR]
          Mat<R, M, M, MK::FJ> Fb;
          LQG_UNROLL for (int i = 0; i < M; ++i)
            LQG_UNROLL for (int k = 0; k < M; ++k)
              if (MK::FJ.b[i * M + k]) {
                R v = MC.v[i * M + k];
                if (k >= O) { if (decltype(FbR)::mask.b[i * RR + (k >= O ? k - O : 0)]) v += R(2) * FbR.v[i * RR + (k >= O ? k - O : 0)]; }
                Fb.v[i * M + k] = v;
              }
          auto G11 = blk<0, 0, NX, NX>(SbM);
          auto G21 = blk<NX, 0, NB, NX>(SbM);
          auto G22 = blk<NX, NX, NB, NB>(SbM);
          // ---- conditioning on x_t, in terms of Wm and a (no product of two inverses: oracle/lqg_adjoint_np.py)
          {
            const auto SF = mul(SbM, F2);
            const auto Ch = sym_part(mul_tn(F2, SF));                                      // [RR, RR]
            R Chd[RR * RR], ChW[RR * O];
            to_dense(Ch, Chd);
            LQG_UNROLL for (int p = 0; p < RR; ++p)
              LQG_UNROLL for (int k = 0; k < O; ++k) {
                R v = R(0);
                LQG_UNROLL for (int q = 0; q < RR; ++q) v += Chd[p * RR + q] * Wm[q * O + k];
                ChW[p * O + k] = v;
              }
            LQG_UNROLL for (int i = 0; i < O; ++i)
              LQG_UNROLL for (int k = i; k < O; ++k) {
                R v = R(0);
                LQG_UNROLL for (int p = 0; p < RR; ++p)
                  v += Wm[p * O + i] * (ChW[p * O + k] - CA[p * O + k]) + Wm[p * O + k] * (ChW[p * O + i] - CA[p * O + i]);
                Sigb[i * M + k] = R(0.5) * v;
                Sigb[k * M + i] = R(0.5) * v;
              }
            LQG_UNROLL for (int p = 0; p < RR; ++p)
              LQG_UNROLL for (int k = 0; k < O; ++k) {
                const R v = R(0.5) * (CA[p * O + k] - R(2) * ChW[p * O + k]);
                Sigb[(O + p) * M + k] = v;
                Sigb[k * M + O + p] = v;
              }
            LQG_UNROLL for (int p = 0; p < RR; ++p)
              LQG_UNROLL for (int q = 0; q < RR; ++q) Sigb[(O + p) * M + O + q] = Chd[p * RR + q];
          }
          if (!WHOLE && t == 0) {                                                          // Sigma_0 = G_0 G_0'
            const auto S0m = from_dense<R, M, M>(Sigb);
            accum(G11, blk<0, 0, NX, NX>(S0m));
            accum(G21, blk<NX, 0, NB, NX>(S0m));
            accum(G22, blk<NX, NX, NB, NB>(S0m));
          }
          // ---- joint system -> Kbar, Lbar and the accumulated bars                      system.py:167-207
          const auto F11 = blk<0, 0, NX, NX>(Fb);
          const auto F12 = blk<0, NX, NX, NB>(Fb);
          const auto F21 = blk<NX, 0, NB, NX>(Fb);
          const auto F22 = blk<NX, NX, NB, NB>(Fb);
          const auto BKb = mul_nt(F22, L);                                                 // [NB, NU]
          {
            Mat<R, NB, NY, KT::mask> Kb;
            set_zero(Kb);
            accum(Kb, mul_nt(F21, c.FAd));
            accum(Kb, mul_nt(F22, c.FAa), R(-1));
            accum(Kb, mul_nt(BKb, c.DB));
            accum(Kb, mul_nt(G21, c.N2), R(2));
            accum(Kb, mul(G22, KN3), R(2));
            // K-bar_t without its P-bar term (- P-bar FP'), which k_asp_kal_rev adds: the set entries of K's mask, in row-major order
            R* dst = A.Kbar + (long)t * (NB * NY) * a.ldb + s;
            int e = 0;
            LQG_UNROLL for (int i = 0; i < NB * NY; ++i)
              if (KT::mask.b[i]) dst[(e++) * a.ldb] = Kb.v[i];
          }
          {
            const auto Lb = add(mul_tn(c.Bd, F12), mul_tn(BK, F22));                       // [NU, NB]
            R* dst = A.Lbar + (long)t * (NU * NB) * a.ldb + s;
            store_col(dst, a.ldb, Lb);
          }
          accum(bAd, F11);
          accum(bBd, mul_nt(F12, L));
          accum(bFAd, mul_tn(K, F21));
          accum(bAa, F22);
          accum(bFAa, mul_tn(K, F22), R(-1));
          accum(bBa, BKb);
          accum(bDB, mul_tn(K, BKb));
          accum(bN1, G11);
          accum(bN2, mul_tn(K, G21), R(2));
          accum(bN3, mul_tn(K, mul(G22, K)));
          // the state after step t - 1 is the state before step t
          LQG_UNROLL for (int i = 0; i < O; ++i)
            LQG_UNROLL for (int k = 0; k < O; ++k) SooN[i * O + k] = Sg[i * M + k];
        });
      }
    }
  };
  {
    int t0 = (nck - 1) * CK;
    if (t0 + CK > a.T && t0 > 0) { chunk.template operator()<false>(t0); t0 -= CK; }     // the last, partial chunk
    for (; t0 > 0; t0 -= CK) chunk.template operator()<true>(t0);
    chunk.template operator()<false>(0);                                // (step 0: Sigma_0 = G_0 G_0')
  }
  // ---- chain the hoisted products' bars to the stored matrices, write the gradient
  // FAa = Fa Aa, FAd = Fd Ad, DB = Fd Bd - Fa Ba, N2 = Fd N1, N3 = Fd N1 Fd' + WWd
  // (the bars of VVa, WWa, Sigma0 and the Kalman parts of Aa, Fa: k_asp_kal_rev, which runs next and ADDS to AA and AF)
  {
    R* o = A.out + s;
    const long ld = A.ld;
    accum(bAd, mul_tn(c.Fd, bFAd));
    accum(bBd, mul_tn(c.Fd, bDB));
    accum(bAa, mul_tn(c.Fa, bFAa));
    accum(bBa, mul_tn(c.Fa, bDB), R(-1));
    accum(bFa, mul_nt(bFAa, c.Aa));
    accum(bFa, mul_nt(bDB, c.Ba), R(-1));
    Mat<R, NY, NX, PAT::Fd> bFd;
    set_zero(bFd);
    accum(bFd, mul_nt(bFAd, c.Ad));
    accum(bFd, mul_nt(bDB, c.Bd));
    accum(bFd, mul(bN2, c.N1));
    accum(bFd, mul(sym_part(bN3), mul(c.Fd, c.N1)), R(2));
    accum(bN1, mul_tn(c.Fd, bN2));
    accum(bN1, mul_tn(c.Fd, mul(bN3, c.Fd)));
    // bars of fields that no parameter moves (PAT::live_*: lqg_amd/specialize.py) are written as zeros — everything that only
    // feeds them (their accumulators, the hoisted products' bars, the F11 block of MC, ...) is dead code and compiled out
    auto put = [&]<bool LIVE>(int off, const auto& m) LQG_LAMBDA_INLINE {
      if constexpr (LIVE) store_col(o + off * ld, ld, m);
      else {
        using MT = std::remove_cvref_t<decltype(m)>;
        LQG_UNROLL for (int i = 0; i < MT::rows * MT::cols; ++i) o[(off + i) * ld] = R(0);
      }
    };
    put.template operator()<PAT::live_Ad>(Lay::DA, bAd);
    put.template operator()<PAT::live_Bd>(Lay::DB, bBd);
    put.template operator()<PAT::live_Fd>(Lay::DF, bFd);
    put.template operator()<PAT::live_Vd>(Lay::DVV, bN1);
    {
      Mat<R, NY, NY, PAT::WWd> bWWd;
      set_zero(bWWd);
      accum(bWWd, bN3);
      put.template operator()<PAT::live_Wd>(Lay::DWW, bWWd);
    }
    put.template operator()<PAT::live_Aa>(Lay::AA, bAa);
    put.template operator()<PAT::live_Ba>(Lay::AB, bBa);
    put.template operator()<PAT::live_Fa>(Lay::AF, bFa);
  }
}

// ================================================================= phase 2b: adjoint of the Kalman-covariance recursion
// Backward in time over K-bar_t (left by k_asp_sys_rev, which must have run): the chunk's P_t are recomputed from the same
// checkpoints into registers (P alone: NB (NB + 1) / 2 reals per step).  Writes the bars of VVa, WWa and Sigma0, ADDS its parts
// of Aa and Fa to what k_asp_sys_rev wrote.                                                                     kf.py:10-14
template <typename R, int NX, int NB, int NU, int NY, int ND, typename PAT, int NTR, bool DENSE_P, int CK>
__global__ void __launch_bounds__(LQG_BLOCK, (NB <= 3 ? 2 : 1)) k_asp_kal_rev(const AspArgs<R> A) {   // (b = 4 dense: 86 spills at two waves)
  constexpr int M = NX + NB;
  using Rec = CkRec<NB, M, NTR>;
  using MK = Masks<PAT, NX, NB, NU, NY, DENSE_P>;
  using Lay = adj::Layout<NX, NB, NU, NY>;
  using KT = KalmanGain<R, NX, NB, NU, NY, PAT, MK::PM>;
  constexpr int NSB = NB * (NB + 1) / 2;
  const ForwardArgs<R>& a = A.f;
  const long s = blockIdx.x * (long)LQG_BLOCK + threadIdx.x;
  if (s >= a.n_sys) return;
  SysConst<R, NX, NB, NU, NY, PAT> c;
  c.load(a, s);
  Mat<R, NB, NB, PAT::Aa> bAa;  Mat<R, NY, NB, PAT::Fa> bFa;  Mat<R, NB, NB, PAT::VVa> bVVa;  Mat<R, NY, NY, PAT::WWa> bWWa;
  set_zero(bAa); set_zero(bFa); set_zero(bVVa); set_zero(bWWa);
  Mat<R, NB, NB, MK::PM> Pb;
  set_zero(Pb);
  constexpr int S0 = 1;
  constexpr int CKS = CK - S0 > 0 ? CK - S0 : 1;
  R Pst[CKS][NSB];
  auto load_p = [&](const int rec, Mat<R, NB, NB, MK::PM>& Pm) LQG_LAMBDA_INLINE {
    const R* src = A.ck + (long)rec * Rec::W * a.ldb + s;
    R Pd[NB * NB];
    load_tri_arr<R, NB>(src + Rec::P_OFF * a.ldb, a.ldb, Pd);
    LQG_UNROLL for (int i = 0; i < NB * NB; ++i) if (MK::PM.b[i]) Pm.v[i] = Pd[i];
  };
  auto chunk = [&]<bool WHOLE>(const int t0) LQG_LAMBDA_INLINE {
    {
      Mat<R, NB, NB, MK::PM> Pm;
      load_p(t0 / CK, Pm);
      LQG_UNROLL for (int j = 0; j + 1 < CK; ++j) {
        if (WHOLE || t0 + j + 1 < a.T) {
          const auto kp = kalman_part(c, Pm);
          assign_state(Pm, kp.Pn);
          R Pd[NB * NB];
          to_dense(Pm, Pd);
          int e = 0;
          LQG_UNROLL for (int i = 0; i < NB; ++i)
            LQG_UNROLL for (int k = 0; k <= i; ++k) Pst[j][e++] = Pd[i * NB + k];
        }
      }
    }
    LQG_UNROLL for (int j = CK - 1; j >= 0; --j) {
      const int t = t0 + j;
      if (WHOLE || t < a.T) {
        Mat<R, NB, NB, MK::PM> Pm;
        if (j >= S0) {
          int e = 0;
          LQG_UNROLL for (int i = 0; i < NB; ++i)
            LQG_UNROLL for (int k = 0; k <= i; ++k) {
              const R v = Pst[j >= S0 ? j - S0 : 0][e++];
              if (MK::PM.b[i * NB + k]) Pm.v[i * NB + k] = v;
              if (MK::PM.b[k * NB + i]) Pm.v[k * NB + i] = v;
            }
        } else {
          load_p(t0 / CK, Pm);
        }
        const auto kp = kalman_part(c, Pm);
        Mat<R, NB, NY, KT::mask> Kb;
        {
          const R* src = A.Kbar + (long)t * (NB * NY) * a.ldb + s;
          int e = 0;
          LQG_UNROLL for (int i = 0; i < NB * NY; ++i)
            if (KT::mask.b[i]) Kb.v[i] = src[(e++) * a.ldb];
        }
        accum(Kb, mul_nt(Pb, kp.FP), R(-1));
        const auto KbGi = mul(Kb, kp.Gi);                                                // [NB, NY]
        const auto Gmb = scaled(mul(kp.Gi, mul(kp.FP, KbGi)), R(-1));                    // [NY, NY]
        const auto FPb = add(sub(transpose(KbGi), mul_tn(kp.K, Pb)), mul(Gmb, c.Fa));    // [NY, NB]
        accum(bFa, mul(FPb, kp.Pp));
        accum(bFa, mul_tn(Gmb, kp.FP));
        accum(bWWa, Gmb);
        const auto Ppb = sym_part(add(Pb, mul_tn(c.Fa, FPb)));
        accum(bVVa, Ppb);
        const auto PA = mul(Ppb, c.Aa);
        accum(bAa, mul(PA, Pm), R(2));
        {
          const auto Pb1 = sym_part(mul_tn(c.Aa, PA));
          LQG_UNROLL for (int i = 0; i < NB * NB; ++i)
            if (MK::PM.b[i]) Pb.v[i] = decltype(Pb1)::mask.b[i] ? Pb1.v[i] : R(0);
        }
      }
    }
  };
  {
    const int nck = (a.T + CK - 1) / CK;
    int t0 = (nck - 1) * CK;
    if (t0 + CK > a.T && t0 > 0) { chunk.template operator()<false>(t0); t0 -= CK; }
    for (; t0 > 0; t0 -= CK) chunk.template operator()<true>(t0);
    chunk.template operator()<false>(0);
  }
  R* o = A.out + s;
  const long ld = A.ld;
  if (!(DENSE_P && a.Sigma0.p)) accum(bVVa, Pb);                                           // default Sigma0 = V_0 V_0'  system.py:160
  auto put = [&]<bool LIVE>(int off, const auto& m) LQG_LAMBDA_INLINE {
    if constexpr (LIVE) store_col(o + off * ld, ld, m);
    else {
      using MT = std::remove_cvref_t<decltype(m)>;
      LQG_UNROLL for (int i = 0; i < MT::rows * MT::cols; ++i) o[(off + i) * ld] = R(0);
    }
  };
  auto add_to = [&]<bool LIVE>(int off, const auto& m) LQG_LAMBDA_INLINE {
    using MT = std::remove_cvref_t<decltype(m)>;
    if constexpr (LIVE) {
      LQG_UNROLL for (int i = 0; i < MT::rows * MT::cols; ++i) if (MT::mask.b[i]) o[(off + i) * ld] += m.v[i];
    }
  };
  put.template operator()<PAT::live_Va>(Lay::AVV, bVVa);
  put.template operator()<PAT::live_Wa>(Lay::AWW, bWWa);
  add_to.template operator()<PAT::live_Aa>(Lay::AA, bAa);
  add_to.template operator()<PAT::live_Fa>(Lay::AF, bFa);
  store_col(o + Lay::AS0 * ld, ld, Pb);
}

// ================================================================= phase 2: adjoint of the Riccati recursion (forward in time)
// consumes Lbar_t; the chunk's S_{t+1} are recomputed backward from the checkpoint into registers.   lqr.py:16-42
template <typename R, int NB, int NU, int NX, int NY, typename PAT, int CK>
__global__ void __launch_bounds__(LQG_BLOCK, 2) k_asp_ric_rev(const AspArgs<R> A) {
  using Lay = adj::Layout<NX, NB, NU, NY>;
  constexpr int NS = NB * (NB + 1) / 2;
  const RiccatiArgs<R>& rc = A.rc;
  const long s = blockIdx.x * (long)LQG_BLOCK + threadIdx.x;
  if (s >= rc.n_sys) return;
  const auto Am = load_masked<R, NB, NB, PAT::Aa>(rc.A.p + s * rc.A.sb, rc.A.sr, rc.A.sc);
  const auto Bm = load_masked<R, NB, NU, PAT::Ba>(rc.B.p + s * rc.B.sb, rc.B.sr, rc.B.sc);
  const auto rQ = load_sym_masked<R, NB, PAT::Q>(rc.Q.p + s * rc.Q.sb, rc.Q.sr, rc.Q.sc);
  const auto rR = load_sym_masked<R, NU, PAT::Rr>(rc.Rm.p + s * rc.Rm.sb, rc.Rm.sr, rc.Rm.sc);
  R Ad[NB * NB], Bd[NB * NU], Rd[NU * NU], Pz[NU * NB];
  to_dense(Am, Ad);
  to_dense(Bm, Bd);
  to_dense(rR, Rd);
  adj::zero<R, NU * NB>(Pz);
  R bA[NB * NB], bB[NB * NU], bQ[NB * NB], bR[NU * NU], Sb[NB * NB];
  adj::zero<R, NB * NB>(bA); adj::zero<R, NB * NU>(bB); adj::zero<R, NB * NB>(bQ); adj::zero<R, NU * NU>(bR); adj::zero<R, NB * NB>(Sb);
  adj::RicStep<R, NB, NU> st;
  R Sst[CK][NS];
  auto chunk = [&]<bool WHOLE>(const int t0) LQG_LAMBDA_INLINE {
    {   // S_{t+1} of the chunk's steps, backward from the checkpoint S_{t0 + CK}
      R Sr[NB * NB], Lt[NU * NB];
      const R* src = rc.Ls + (long)(t0 / CK) * NS * rc.ldb + s;
      int e = 0;
      LQG_UNROLL for (int i = 0; i < NB; ++i)
        LQG_UNROLL for (int j = i; j < NB; ++j) { const R v = src[(e++) * rc.ldb]; Sr[i * NB + j] = v; Sr[j * NB + i] = v; }
      LQG_UNROLL for (int j = CK - 1; j >= 0; --j)
        if (WHOLE || t0 + j < rc.T) {
          int k = 0;
          LQG_UNROLL for (int i = 0; i < NB; ++i)
            LQG_UNROLL for (int q = 0; q <= i; ++q) Sst[j][k++] = Sr[i * NB + q];
          if (j > 0) riccati_step_sp<R, NB, NU>(Sr, Am, Bm, rQ, rR, rc.eps, Lt);
        }
    }
    LQG_UNROLL for (int j = 0; j < CK; ++j) {
      const int t = t0 + j;
      if (WHOLE || t < rc.T) {
        R S[NB * NB], Lb[NU * NB];
        {
          int k = 0;
          LQG_UNROLL for (int i = 0; i < NB; ++i)
            LQG_UNROLL for (int q = 0; q <= i; ++q) { const R v = Sst[j][k++]; S[i * NB + q] = v; S[q * NB + i] = v; }
        }
        const R* lsrc = A.Lbar + (long)t * (NU * NB) * rc.ldb + s;
        LQG_UNROLL for (int e = 0; e < NU * NB; ++e) Lb[e] = lsrc[e * rc.ldb];
        st.compute(S, Ad, Bd, Rd, Pz, rc.eps);
        LQG_UNROLL for (int k = 0; k < NB * NB; ++k) bQ[k] += Sb[k];
        R HLG[NU * NB], Gb[NU * NB], Hb[NU * NU], HtiLb[NU * NB], LSb[NU * NB];
        adj::copy<R, NU * NB>(st.G, HLG);
        adj::mm_acc<R, NU, NU, NB>(st.H, st.L, HLG);
        adj::mm_acc<R, NU, NB, NB>(HLG, Sb, Lb, R(2));                    // Lb += 2 (H L + G) Sb
        adj::zero<R, NU * NB>(HtiLb);
        adj::mm_acc<R, NU, NU, NB>(st.Hti, Lb, HtiLb);
        adj::zero<R, NU * NB>(LSb);
        adj::mm_acc<R, NU, NB, NB>(st.L, Sb, LSb);
        LQG_UNROLL for (int k = 0; k < NU * NB; ++k) Gb[k] = R(2) * LSb[k] - HtiLb[k];
        adj::zero<R, NU * NU>(Hb);
        adj::mmt_acc<R, NU, NB, NU>(LSb, st.L, Hb);
        adj::mmt_acc<R, NU, NB, NU>(HtiLb, st.L, Hb, R(-1));
        LQG_UNROLL for (int k = 0; k < NU * NU; ++k) bR[k] += Hb[k];
        adj::mm_acc<R, NB, NB, NB>(st.SA, Sb, bA, R(2));
        adj::mm_acc<R, NB, NU, NB>(st.SB, Gb, bA);
        adj::mmt_acc<R, NB, NB, NU>(st.SA, Gb, bB);
        R Hs[NU * NU];
        LQG_UNROLL for (int p = 0; p < NU; ++p)
          LQG_UNROLL for (int q = 0; q < NU; ++q) Hs[p * NU + q] = Hb[p * NU + q] + Hb[q * NU + p];
        adj::mm_acc<R, NB, NU, NU>(st.SB, Hs, bB);
        R X1[NB * NB], X2[NB * NU];
        adj::zero<R, NB * NB>(X1);
        adj::mm_acc<R, NB, NB, NB>(Ad, Sb, X1);
        adj::mm_acc<R, NB, NU, NB>(Bd, Gb, X1);
        adj::zero<R, NB * NU>(X2);
        adj::mm_acc<R, NB, NU, NU>(Bd, Hb, X2);
        adj::zero<R, NB * NB>(Sb);
        adj::mmt_acc<R, NB, NB, NB>(X1, Ad, Sb);
        adj::mmt_acc<R, NB, NU, NB>(X2, Bd, Sb);
        adj::symmetrise<R, NB>(Sb);
      }
    }
  };
  {
    int t0 = 0;
    for (; t0 + CK <= rc.T; t0 += CK) chunk.template operator()<true>(t0);
    if (t0 < rc.T) chunk.template operator()<false>(t0);
  }
  R* o = A.out + s;
  const long ld = A.ld;
  if constexpr (!PAT::live_Aa) adj::zero<R, NB * NB>(bA);
  if constexpr (!PAT::live_Ba) adj::zero<R, NB * NU>(bB);
  if constexpr (!PAT::live_Q) adj::zero<R, NB * NB>(bQ);
  if constexpr (!PAT::live_R) adj::zero<R, NU * NU>(bR);
  adj::store_flat<R, NB * NB>(o + Lay::AA2 * ld, ld, bA);
  adj::store_flat<R, NB * NU>(o + Lay::AB2 * ld, ld, bB);
  adj::store_flat<R, NB * NB>(o + Lay::AQ * ld, ld, bQ);
  adj::store_flat<R, NU * NU>(o + Lay::AR * ld, ld, bR);
  adj::store_flat<R, NB * NB>(o + Lay::AQF * ld, ld, Sb);
}

}  // namespace asp
}  // namespace lqg
