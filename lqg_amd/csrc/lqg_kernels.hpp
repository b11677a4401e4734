// lqg_kernels.hpp — the time sweeps of the LQG solve path as HIP kernels for gfx950 (MI355X).
//
// Mapping: ONE SYSTEM PER LANE for the per-system sweeps (Riccati backward, Kalman /
// joint-system / covariance forward), ONE TRIAL PER LANE for the per-trial mean/log-density sweep.  A wave
// therefore advances 64 independent recursions in lock step; the small matrices live in VGPRs with
// compile-time extents, every flop is a plain v_fma with no cross-lane traffic, and HBM is touched only
// for (a) the spec fields (once per lane when time-invariant), (b) the control gains L_t parked in a
// [t][element][system] scratch between the backward and the forward sweep (coalesced across lanes),
// (c) the observed trajectory, (d) the outputs.
//
// Reference parity (paths relative to the reference checkout):
//   k_riccati  <- lqg/control/lqr.py:16-42  backward()
//   k_kalman   <- lqg/belief/kf.py:6-21     forward()
//   k_forward  <- lqg/belief/kf.py:6-21 + lqg/system.py:167-235 (joint system, moment recursion)
//                 [+ lqg/system.py:237-248 log-density when FUSED]
//   k_trial    <- lqg/system.py:219-221 (mean recursion) + :244-248 (MultivariateNormal log_prob)
//   k_simulate <- lqg/system.py:106-128
//   k_gaussian_logprob <- numpyro MultivariateNormal.log_prob as used at lqg/system.py:244,248
//
// The moment recursion is evaluated in Schur-complement form: with o = observed dims and
// Sigma = [[Soo, Sor],[Sro, Srr]], Lc = chol(Soo), U2 = Sro Lc^-T, C = Srr - U2 U2^T,
//   Sigma' = F[:, o:] C F[:, o:]^T + G G^T        (== F Sigma F^T + G G^T - (F Sigma)[:, :o] Soo^-1 (Sigma F^T)[:o, :])
//   mu'    = F [x_t ; mu_r + U2 Lc^-1 (x_t - mu_o)] (== F mu + (F Sigma)[:, :o] Soo^-1 (x_t - mu_o))
// which is algebraically identical to system.py:219-230, needs ~3x fewer flops and shares one Cholesky of
// Soo between the conditioning step and the log-density of the same innovation.
#pragma once
#include "lqg_rng.hpp"
#include "lqg_small.hpp"

#ifndef LQG_BLOCK
#define LQG_BLOCK 64
#endif
// minimum waves per SIMD the register allocator must leave room for (2nd __launch_bounds__ argument)
#ifndef LQG_TRIAL_OPS_PREFETCH
#define LQG_TRIAL_OPS_PREFETCH 1
#endif
#ifndef LQG_TRIAL_PREFETCH
#define LQG_TRIAL_PREFETCH 1
#endif
#ifndef LQG_FWD_WAVES
#define LQG_FWD_WAVES 1
#endif
#ifndef LQG_RIC_WAVES
#define LQG_RIC_WAVES 1
#endif

namespace lqg {

// number of reals per (system, step) in the trial-operator stream written by k_forward, read by k_trial
template <int M, int ND>
struct TrialOps {
  static constexpr int O = ND, RR = M - ND;
  static constexpr int F_OFF = 0;                       // Fj[M,M] - I  (deviation form, see k_trial)
  static constexpr int U_OFF = M * M;                   // U2[RR,O]
  static constexpr int L_OFF = U_OFF + RR * O;          // Li lower, packed by rows
  static constexpr int H_OFF = L_OFF + O * (O + 1) / 2; // half log-det + d/2 log(2 pi)
  static constexpr int RAW = H_OFF + 1;
  static constexpr int N = (RAW + 3) / 4 * 4;
};

// ===================================================================== Riccati backward ==============
template <typename R>
struct RiccatiArgs {
  DView<R> Q, q, Qf, qf, P, Rm, r, A, B;
  DView<R> L, l, H;   // outputs, any may be null
  R* Ls;              // internal gain scratch [T][NU*NB][ldb], may be null
  long ldb;
  long n_sys;
  int T;
  R eps;
};

template <typename R, int NB, int NU, bool TI, bool AFFINE>
__global__ void __launch_bounds__(LQG_BLOCK, LQG_RIC_WAVES) k_riccati(const RiccatiArgs<R> a) {
  const long s = blockIdx.x * (long)LQG_BLOCK + threadIdx.x;
  if (s >= a.n_sys) return;

  R S[NB * NB], sv[NB];
  load_sym<R, NB>(a.Qf.p + s * a.Qf.sb, a.Qf.sr, a.Qf.sc, S);   // carry init (Qf, qf)  lqr.py:38
  LQG_UNROLL for (int i = 0; i < NB; ++i) sv[i] = R(0);
  if (AFFINE && a.qf.p) load_vec<R, NB>(a.qf.p + s * a.qf.sb, a.qf.sr, sv);

  R A[NB * NB], Bm[NB * NU], Q[NB * NB], Rm[NU * NU], P[NU * NB], q[NB], r[NU];
  auto load_step = [&](int t) {
    load_mat<R, NB, NB>(a.A.p + s * a.A.sb + t * a.A.st, a.A.sr, a.A.sc, A);
    load_mat<R, NB, NU>(a.B.p + s * a.B.sb + t * a.B.st, a.B.sr, a.B.sc, Bm);
    load_sym<R, NB>(a.Q.p + s * a.Q.sb + t * a.Q.st, a.Q.sr, a.Q.sc, Q);
    load_sym<R, NU>(a.Rm.p + s * a.Rm.sb + t * a.Rm.st, a.Rm.sr, a.Rm.sc, Rm);
    if (AFFINE) {
      LQG_UNROLL for (int i = 0; i < NU * NB; ++i) P[i] = R(0);
      LQG_UNROLL for (int i = 0; i < NB; ++i) q[i] = R(0);
      LQG_UNROLL for (int i = 0; i < NU; ++i) r[i] = R(0);
      if (a.P.p) load_mat<R, NU, NB>(a.P.p + s * a.P.sb + t * a.P.st, a.P.sr, a.P.sc, P);
      if (a.q.p) load_vec<R, NB>(a.q.p + s * a.q.sb + t * a.q.st, a.q.sr, q);
      if (a.r.p) load_vec<R, NU>(a.r.p + s * a.r.sb + t * a.r.st, a.r.sr, r);
    }
  };
  if (TI) load_step(0);

  for (int t = a.T - 1; t >= 0; --t) {   // reverse=True  lqr.py:40
    if (!TI) load_step(t);
    R SA[NB * NB], SB[NB * NU], H[NU * NU], G[NU * NB];
    mm<R, NB, NB, NB>(S, A, SA);
    mm<R, NB, NB, NU>(S, Bm, SB);
    // H = R + B^T S B (symmetric)                               lqr.py:22
    LQG_UNROLL for (int i = 0; i < NU; ++i)
      LQG_UNROLL for (int j = i; j < NU; ++j) {
        R acc = Rm[i * NU + j];
        LQG_UNROLL for (int k = 0; k < NB; ++k) acc += Bm[k * NU + i] * SB[k * NU + j];
        H[i * NU + j] = acc;
        H[j * NU + i] = acc;
      }
    // G = P + B^T S A                                            lqr.py:23
    LQG_UNROLL for (int i = 0; i < NU; ++i)
      LQG_UNROLL for (int j = 0; j < NB; ++j) {
        R acc = AFFINE ? P[i * NB + j] : R(0);
        LQG_UNROLL for (int k = 0; k < NB; ++k) acc += Bm[k * NU + i] * SA[k * NB + j];
        G[i * NB + j] = acc;
      }
    // Ht = H + max(0, eps - lambda_min(H)) I                    lqr.py:27-28
    R ev0 = min_eig_sym<R, NU>(H);
    R shift = a.eps - ev0;
    shift = (shift > R(0)) ? shift : R(0);
    R Ht[NU * NU];
    LQG_UNROLL for (int i = 0; i < NU * NU; ++i) Ht[i] = H[i];
    LQG_UNROLL for (int i = 0; i < NU; ++i) Ht[i * NU + i] += shift;
    // Ht^-1 via Cholesky (Ht is symmetric with lambda_min >= eps by construction)
    R Lc[NU * NU], dinv[NU], Li[NU * NU], Hi[NU * NU];
    chol_lower<R, NU>(Ht, Lc, dinv);
    tri_inverse_lower<R, NU>(Lc, dinv, Li);
    spd_inverse_from_tri<R, NU>(Li, Hi);
    // L = -Ht^-1 G                                               lqr.py:30
    R L[NU * NB];
    LQG_UNROLL for (int i = 0; i < NU; ++i)
      LQG_UNROLL for (int j = 0; j < NB; ++j) {
        R acc = R(0);
        LQG_UNROLL for (int k = 0; k < NU; ++k) acc -= Hi[i * NU + k] * G[k * NB + j];
        L[i * NB + j] = acc;
      }
    // W1 = H L + G  (unregularised H, lqr.py:33)
    R W1[NU * NB];
    LQG_UNROLL for (int i = 0; i < NU; ++i)
      LQG_UNROLL for (int j = 0; j < NB; ++j) {
        R acc = G[i * NB + j];
        LQG_UNROLL for (int k = 0; k < NU; ++k) acc += H[i * NU + k] * L[k * NB + j];
        W1[i * NB + j] = acc;
      }
    R g[NU], lv[NU], Hl[NU], sn[NB];
    if (AFFINE) {
      LQG_UNROLL for (int i = 0; i < NU; ++i) {                  // g = r + B^T s   lqr.py:24
        R acc = r[i];
        LQG_UNROLL for (int k = 0; k < NB; ++k) acc += Bm[k * NU + i] * sv[k];
        g[i] = acc;
      }
      LQG_UNROLL for (int i = 0; i < NU; ++i) {                  // l = -Ht^-1 g    lqr.py:31
        R acc = R(0);
        LQG_UNROLL for (int k = 0; k < NU; ++k) acc -= Hi[i * NU + k] * g[k];
        lv[i] = acc;
      }
      LQG_UNROLL for (int i = 0; i < NU; ++i) {
        R acc = g[i];
        LQG_UNROLL for (int k = 0; k < NU; ++k) acc += H[i * NU + k] * lv[k];
        Hl[i] = acc;                                             // H l + g
      }
      // s = q + A^T s + G^T l + L^T (H l + g)                    lqr.py:34
      LQG_UNROLL for (int i = 0; i < NB; ++i) {
        R acc = q[i];
        LQG_UNROLL for (int k = 0; k < NB; ++k) acc += A[k * NB + i] * sv[k];
        LQG_UNROLL for (int k = 0; k < NU; ++k) acc += G[k * NB + i] * lv[k] + L[k * NB + i] * Hl[k];
        sn[i] = acc;
      }
    }
    // S = Q + A^T S A + L^T (H L + G) + G^T L   (symmetric)      lqr.py:33
    LQG_UNROLL for (int i = 0; i < NB; ++i)
      LQG_UNROLL for (int j = i; j < NB; ++j) {
        R acc = Q[i * NB + j];
        LQG_UNROLL for (int k = 0; k < NB; ++k) acc += A[k * NB + i] * SA[k * NB + j];
        LQG_UNROLL for (int k = 0; k < NU; ++k) acc += L[k * NB + i] * W1[k * NB + j] + G[k * NB + i] * L[k * NB + j];
        S[i * NB + j] = acc;
        S[j * NB + i] = acc;
      }
    if (AFFINE) { LQG_UNROLL for (int i = 0; i < NB; ++i) sv[i] = sn[i]; }

    // emit (L, l, Ht) at index t: scan stacks reverse outputs in forward order   lqr.py:36,40
    if (a.Ls) {
      R* dst = a.Ls + (long)t * (NU * NB) * a.ldb + s;
      LQG_UNROLL for (int e = 0; e < NU * NB; ++e) dst[e * a.ldb] = L[e];
    }
    if (a.L.p) store_mat<R, NU, NB>(const_cast<R*>(a.L.p) + s * a.L.sb + t * a.L.st, a.L.sr, a.L.sc, L);
    if (a.l.p) {
      R* dst = const_cast<R*>(a.l.p) + s * a.l.sb + t * a.l.st;
      LQG_UNROLL for (int i = 0; i < NU; ++i) dst[i * a.l.sr] = AFFINE ? lv[i] : R(0);
    }
    if (a.H.p) store_mat<R, NU, NU>(const_cast<R*>(a.H.p) + s * a.H.sb + t * a.H.st, a.H.sr, a.H.sc, Ht);
  }
}

// ===================================================================== Kalman forward ================
// One Kalman step on register-resident symmetric P.  Returns K[NB,NY]; P is updated in place.
//   Pp = A P A^T + V V^T ; Gk = F Pp F^T + W W^T ; K = Pp F^T Gk^-1 ; P = Pp - K (F Pp)     kf.py:10-14
template <typename R, int NB, int NY>
LQG_DEV void kalman_step(const R (&A)[NB * NB], const R (&F)[NY * NB], const R (&VV)[NB * NB],
                         const R (&WW)[NY * NY], R (&P)[NB * NB], R (&K)[NB * NY]) {
  R AP[NB * NB], Pp[NB * NB], FP[NY * NB], Gk[NY * NY];
  mm<R, NB, NB, NB>(A, P, AP);
  mmt_sym_add<R, NB, NB>(AP, A, VV, Pp);
  mm<R, NY, NB, NB>(F, Pp, FP);
  mmt_sym_add<R, NY, NB>(FP, F, WW, Gk);
  R Lc[NY * NY], dinv[NY], Li[NY * NY], Gi[NY * NY];
  chol_lower<R, NY>(Gk, Lc, dinv);
  tri_inverse_lower<R, NY>(Lc, dinv, Li);
  spd_inverse_from_tri<R, NY>(Li, Gi);
  // K = (F Pp)^T Gk^-1   (Pp symmetric => Pp F^T = (F Pp)^T)
  LQG_UNROLL for (int i = 0; i < NB; ++i)
    LQG_UNROLL for (int j = 0; j < NY; ++j) {
      R acc = FP[i] * Gi[j];
      LQG_UNROLL for (int k = 1; k < NY; ++k) acc += FP[k * NB + i] * Gi[k * NY + j];
      K[i * NY + j] = acc;
    }
  LQG_UNROLL for (int i = 0; i < NB; ++i)
    LQG_UNROLL for (int j = i; j < NB; ++j) {
      R acc = Pp[i * NB + j];
      LQG_UNROLL for (int k = 0; k < NY; ++k) acc -= K[i * NY + k] * FP[k * NB + j];
      P[i * NB + j] = acc;
      P[j * NB + i] = acc;
    }
}

template <typename R>
struct KalmanArgs {
  DView<R> A, F, V, W, Sigma0;
  DView<R> K;  // output
  long n_sys;
  int T, nv, nw;
};

template <typename R, int NB, int NY, bool TI>
__global__ void __launch_bounds__(LQG_BLOCK) k_kalman(const KalmanArgs<R> a) {
  const long s = blockIdx.x * (long)LQG_BLOCK + threadIdx.x;
  if (s >= a.n_sys) return;
  R P[NB * NB];
  if (a.Sigma0.p) load_sym<R, NB>(a.Sigma0.p + s * a.Sigma0.sb, a.Sigma0.sr, a.Sigma0.sc, P);
  else load_gram<R, NB>(a.V.p + s * a.V.sb, a.V.sr, a.V.sc, a.nv, P);   // V[0] V[0]^T  system.py:79,160
  R A[NB * NB], F[NY * NB], VV[NB * NB], WW[NY * NY];
  auto load_step = [&](int t) {
    load_mat<R, NB, NB>(a.A.p + s * a.A.sb + t * a.A.st, a.A.sr, a.A.sc, A);
    load_mat<R, NY, NB>(a.F.p + s * a.F.sb + t * a.F.st, a.F.sr, a.F.sc, F);
    load_gram<R, NB>(a.V.p + s * a.V.sb + t * a.V.st, a.V.sr, a.V.sc, a.nv, VV);
    load_gram<R, NY>(a.W.p + s * a.W.sb + t * a.W.st, a.W.sr, a.W.sc, a.nw, WW);
  };
  if (TI) load_step(0);
  for (int t = 0; t < a.T; ++t) {
    if (!TI) load_step(t);
    R K[NB * NY];
    kalman_step<R, NB, NY>(A, F, VV, WW, P, K);
    store_mat<R, NB, NY>(const_cast<R*>(a.K.p) + s * a.K.sb + t * a.K.st, a.K.sr, a.K.sc, K);
  }
}

// ===================================================================== forward sweep =================
template <typename R>
struct ForwardArgs {
  DView<R> aA, aB, aF, aV, aW;   // actor spec
  DView<R> dA, dB, dF, dV, dW;   // dynamics spec
  DView<R> Sigma0;               // Kalman initial covariance, may be null
  const R* Ls;                   // gain scratch [T][NU*NB][ldb] from k_riccati
  long ldb;
  DTraj<R> x;                    // observed data (FUSED only): trial 0 of every system
  R* ll;                         // FUSED: ll[s * ll_sb]
  long ll_sb;
  R* ops;                        // !FUSED: trial-operator stream [n_sys][T+1][TrialOps::N], may be null
  DView<R> Sig;                  // optional Sigma output [B,T,m,m]
  DTraj<R> mu;                   // optional mu output [B,1,T,m] (FUSED only: the trial swept in-lane)
  DView<R> Kout;                 // optional Kalman gain output [B,T,b,y]
  long n_sys;
  int T, nva, nwa, nvd, nwd;
  // MIXED mode of the structure-specialised libraries (k_forward_sp<double, ..., OT = float>), both may be null: the rounding
  // residual of the operator's Fj - I block, fl32(F - fl32(F)), as [n_sys][T+1][hilo_reals] — written from the first step
  // at which |Fj - I| reaches LQG_HILO_MIN — and per system that step + 1 (0: never): the per-trial sweep applies hi + lo from
  // there on (k_trial_sp<..., HL>)
  float* ops_lo;
  int* hl;
};

// MAT = materialise (Sigma / mu / K outputs requested): kept out of the pure log-likelihood instantiation so that the
// hot kernel carries no store code, no output address arithmetic and no extra live pointers.
// OT: element type of the operator stream (!FUSED).  OT = float with R = double is the MIXED mode of an fp32 problem whose
// per-system sweeps run in fp64 (include/lqg_hip.h: LQG_F32_SYS64): the operators are rounded to fp32 ONCE, on the way out.
template <typename R, int NX, int NB, int NU, int NY, int ND, bool TI, bool FUSED, bool MAT, typename OT = R>
__global__ void __launch_bounds__(LQG_BLOCK, LQG_FWD_WAVES) k_forward(const ForwardArgs<R> a) {
  constexpr int M = NX + NB, O = ND, RR = M - ND;
  using Ops = TrialOps<M, ND>;
  const long s = blockIdx.x * (long)LQG_BLOCK + threadIdx.x;
  if (s >= a.n_sys) return;

  // ---- per-step constants (loaded once when every spec field is time-invariant)
  R Aa[NB * NB], Ba[NB * NU], Fa[NY * NB], VVa[NB * NB], WWa[NY * NY];
  R Ad[NX * NX], Bd[NX * NU], N1[NX * NX];
  R FAa[NY * NB], FAd[NY * NX], DB[NY * NU], N2[NY * NX], N3[NY * NY];
  auto load_step = [&](int t) {
    R Fd[NY * NX], WWd[NY * NY];
    load_mat<R, NB, NB>(a.aA.p + s * a.aA.sb + t * a.aA.st, a.aA.sr, a.aA.sc, Aa);
    load_mat<R, NB, NU>(a.aB.p + s * a.aB.sb + t * a.aB.st, a.aB.sr, a.aB.sc, Ba);
    load_mat<R, NY, NB>(a.aF.p + s * a.aF.sb + t * a.aF.st, a.aF.sr, a.aF.sc, Fa);
    load_gram<R, NB>(a.aV.p + s * a.aV.sb + t * a.aV.st, a.aV.sr, a.aV.sc, a.nva, VVa);
    load_gram<R, NY>(a.aW.p + s * a.aW.sb + t * a.aW.st, a.aW.sr, a.aW.sc, a.nwa, WWa);
    load_mat<R, NX, NX>(a.dA.p + s * a.dA.sb + t * a.dA.st, a.dA.sr, a.dA.sc, Ad);
    load_mat<R, NX, NU>(a.dB.p + s * a.dB.sb + t * a.dB.st, a.dB.sr, a.dB.sc, Bd);
    load_mat<R, NY, NX>(a.dF.p + s * a.dF.sb + t * a.dF.st, a.dF.sr, a.dF.sc, Fd);
    load_gram<R, NX>(a.dV.p + s * a.dV.sb + t * a.dV.st, a.dV.sr, a.dV.sc, a.nvd, N1);   // Vd Vd^T
    load_gram<R, NY>(a.dW.p + s * a.dW.sb + t * a.dW.st, a.dW.sr, a.dW.sc, a.nwd, WWd);  // Wd Wd^T
    mm<R, NY, NB, NB>(Fa, Aa, FAa);                 // Fa Aa
    mm<R, NY, NX, NX>(Fd, Ad, FAd);                 // Fd Ad
    R FBd[NY * NU], FBa[NY * NU];
    mm<R, NY, NX, NU>(Fd, Bd, FBd);
    mm<R, NY, NB, NU>(Fa, Ba, FBa);
    LQG_UNROLL for (int i = 0; i < NY * NU; ++i) DB[i] = FBd[i] - FBa[i];   // Fd Bd - Fa Ba  system.py:177-180
    mm<R, NY, NX, NX>(Fd, N1, N2);                  // Fd Vd Vd^T
    mmt_sym_add<R, NY, NX>(N2, Fd, WWd, N3);        // Fd Vd Vd^T Fd^T + Wd Wd^T
  };

  R P[NB * NB];
  if (a.Sigma0.p) load_sym<R, NB>(a.Sigma0.p + s * a.Sigma0.sb, a.Sigma0.sr, a.Sigma0.sc, P);
  else load_gram<R, NB>(a.aV.p + s * a.aV.sb, a.aV.sr, a.aV.sc, a.nva, P);               // system.py:160
  if (TI) load_step(0);

  R Sg[M * M];        // predictive covariance of (state, belief), symmetric, mirrored
  // predictive mean (FUSED), observed block in DEVIATION form: mu_o = xprev + dO with xprev the last data row.  The
  // innovation x_t - mu_o = (x_t - x_{t-1}) - dO then never subtracts two large nearly equal numbers (the data
  // difference is exact in floating point, dO is small), which removes the dominant fp32 error of the path
  // (worst case over 2^18 candidates 1.6e-6 -> 3e-7 rel on the log-likelihood).
  R xprev[O], dO[O], muR[RR];
  double acc = 0.0;
  const R* xp = nullptr;
  if (FUSED) {
    xp = a.x.p + s * a.x.sb;
    LQG_UNROLL for (int i = 0; i < O; ++i) { xprev[i] = xp[i * a.x.sd]; dO[i] = R(0); }   // mu0 = [x[0], 0...]  system.py:211
    LQG_UNROLL for (int i = 0; i < RR; ++i) muR[i] = R(0);
  }
  const R kLogNorm = R(0.5 * ND * 1.8378770664093453);

  // conditioning operators of step k from the current Sg: Li = chol(Soo)^-1, U2 = Sro Li^T, hl = sum log diag
  R Li[O * O], U2[RR * O], hl;
  auto condition = [&]() {
    R Soo[O * O], Lc[O * O], dinv[O];
    LQG_UNROLL for (int i = 0; i < O; ++i)
      LQG_UNROLL for (int j = 0; j < O; ++j) Soo[i * O + j] = Sg[i * M + j];
    chol_lower<R, O>(Soo, Lc, dinv);
    tri_inverse_lower<R, O>(Lc, dinv, Li);
    R pd = dinv[0];
    LQG_UNROLL for (int i = 1; i < O; ++i) pd *= dinv[i];
    hl = -log_<R>(pd);                                   // = sum_i log Lc[i][i]
    LQG_UNROLL for (int p = 0; p < RR; ++p)
      LQG_UNROLL for (int j = 0; j < O; ++j) {
        R v = R(0);
        LQG_UNROLL for (int k = 0; k <= j; ++k) v += Sg[(O + p) * M + k] * Li[j * O + k];
        U2[p * O + j] = v;
      }
  };
  // whitened innovation of the observation at row k, and its log-density
  R w[O], xt[O];
  auto innovate = [&](int k, bool score) {
    const R* xr = xp + (long)k * a.x.st;
    LQG_UNROLL for (int i = 0; i < O; ++i) xt[i] = xr[i * a.x.sd];
    R e[O];
    LQG_UNROLL for (int i = 0; i < O; ++i) e[i] = (xt[i] - xprev[i]) - dO[i];
    R zz = R(0);
    LQG_UNROLL for (int i = 0; i < O; ++i) {
      R v = R(0);
      LQG_UNROLL for (int j = 0; j <= i; ++j) v += Li[i * O + j] * e[j];
      w[i] = v;
      zz += v * v;
    }
    if (score) acc -= (double)(R(0.5) * zz + hl + kLogNorm);   // log N(x_k; mu_o, Soo)  system.py:244,248
  };

  for (int t = 0; t < a.T; ++t) {
    if (!TI) load_step(t);
    // ---- Kalman gain K_t                                               kf.py:10-14
    R K[NB * NY];
    kalman_step<R, NB, NY>(Aa, Fa, VVa, WWa, P, K);
    if (MAT && a.Kout.p) store_mat<R, NB, NY>(const_cast<R*>(a.Kout.p) + s * a.Kout.sb + t * a.Kout.st, a.Kout.sr, a.Kout.sc, K);
    // ---- control gain L_t from the backward sweep
    R L[NU * NB];
    {
      const R* src = a.Ls + (long)t * (NU * NB) * a.ldb + s;
      LQG_UNROLL for (int e = 0; e < NU * NB; ++e) L[e] = src[e * a.ldb];
    }
    // ---- joint dynamics Fj = [[Ad, Bd L],[K Fd Ad, Aa - K Fa Aa + (Ba + K (Fd Bd - Fa Ba)) L]]   system.py:167-187
    R Fj[M * M], BK[NB * NU];
    {
      LQG_UNROLL for (int i = 0; i < NB; ++i)
        LQG_UNROLL for (int j = 0; j < NU; ++j) {
          R v = Ba[i * NU + j];
          LQG_UNROLL for (int k = 0; k < NY; ++k) v += K[i * NY + k] * DB[k * NU + j];
          BK[i * NU + j] = v;
        }
      LQG_UNROLL for (int i = 0; i < NX; ++i) {
        LQG_UNROLL for (int j = 0; j < NX; ++j) Fj[i * M + j] = Ad[i * NX + j];
        LQG_UNROLL for (int j = 0; j < NB; ++j) {
          R v = R(0);
          LQG_UNROLL for (int k = 0; k < NU; ++k) v += Bd[i * NU + k] * L[k * NB + j];
          Fj[i * M + NX + j] = v;
        }
      }
      LQG_UNROLL for (int i = 0; i < NB; ++i) {
        LQG_UNROLL for (int j = 0; j < NX; ++j) {
          R v = R(0);
          LQG_UNROLL for (int k = 0; k < NY; ++k) v += K[i * NY + k] * FAd[k * NX + j];
          Fj[(NX + i) * M + j] = v;
        }
        LQG_UNROLL for (int j = 0; j < NB; ++j) {
          R v = Aa[i * NB + j];
          LQG_UNROLL for (int k = 0; k < NY; ++k) v -= K[i * NY + k] * FAa[k * NB + j];
          LQG_UNROLL for (int k = 0; k < NU; ++k) v += BK[i * NU + k] * L[k * NB + j];
          Fj[(NX + i) * M + NX + j] = v;
        }
      }
    }
    // ---- joint noise covariance GG = Gj Gj^T, Gj = [[Vd, 0],[K Fd Vd, K Wd]]                    system.py:190-207
    R GG[M * M];
    {
      R KN2[NB * NX], KN3[NB * NY];
      mm<R, NB, NY, NX>(K, N2, KN2);
      mm<R, NB, NY, NY>(K, N3, KN3);
      LQG_UNROLL for (int i = 0; i < NX; ++i)
        LQG_UNROLL for (int j = 0; j < NX; ++j) GG[i * M + j] = N1[i * NX + j];
      LQG_UNROLL for (int i = 0; i < NB; ++i)
        LQG_UNROLL for (int j = 0; j < NX; ++j) {
          GG[(NX + i) * M + j] = KN2[i * NX + j];
          GG[j * M + NX + i] = KN2[i * NX + j];
        }
      LQG_UNROLL for (int i = 0; i < NB; ++i)
        LQG_UNROLL for (int j = i; j < NB; ++j) {
          R v = R(0);
          LQG_UNROLL for (int k = 0; k < NY; ++k) v += KN3[i * NY + k] * K[j * NY + k];
          GG[(NX + i) * M + NX + j] = v;
          GG[(NX + j) * M + NX + i] = v;
        }
    }
    if (t == 0) { LQG_UNROLL for (int i = 0; i < M * M; ++i) Sg[i] = GG[i]; }   // Sigma0 := G[0] G[0]^T  system.py:212

    // ---- condition on x_t
    condition();
    if (FUSED) {
      innovate(t, t > 0);
      // mu' = Fj [x_t ; mu_r + U2 w]                                  system.py:219-221
      R c[RR];
      LQG_UNROLL for (int p = 0; p < RR; ++p) {
        R v = muR[p];
        LQG_UNROLL for (int j = 0; j < O; ++j) v += U2[p * O + j] * w[j];
        c[p] = v;
      }
      R mn[M];
      LQG_UNROLL for (int i = 0; i < M; ++i) {
        R v = R(0);
        LQG_UNROLL for (int j = 0; j < O; ++j) v += ((i < O && i == j) ? Fj[i * M + j] - R(1) : Fj[i * M + j]) * xt[j];
        LQG_UNROLL for (int p = 0; p < RR; ++p) v += Fj[i * M + O + p] * c[p];
        mn[i] = v;                                              // rows < O: deviation from x_t
      }
      LQG_UNROLL for (int i = 0; i < O; ++i) { dO[i] = mn[i]; xprev[i] = xt[i]; }
      LQG_UNROLL for (int p = 0; p < RR; ++p) muR[p] = mn[O + p];
      if (MAT && a.mu.p) {
        R* dst = const_cast<R*>(a.mu.p) + s * a.mu.sb + (long)t * a.mu.st;
        LQG_UNROLL for (int i = 0; i < M; ++i) dst[i * a.mu.sd] = (i < O) ? xt[i] + mn[i] : mn[i];
      }
    } else if (a.ops) {
      OT* op = reinterpret_cast<OT*>(a.ops) + ((long)s * (a.T + 1) + t) * Ops::N;
      // Fj - I.  The diagonal is ASSEMBLED as a deviation — (A_ii - 1) first (exact for A_ii in [1/2, 2]), then the small
      // terms — instead of subtracting 1 from the rounded entry: fl(F_ii) carries an absolute error of eps/2, i.e. a relative
      // error of eps / |F_ii - 1| on the deviation, and it multiplies the mean state at EVERY step with the same sign.
      LQG_UNROLL for (int i = 0; i < M; ++i)
        LQG_UNROLL for (int j = 0; j < M; ++j) {
          if (i != j) { op[Ops::F_OFF + i * M + j] = (OT)Fj[i * M + j]; continue; }
          R v;
          if (i < NX) {
            v = Ad[i * NX + i] - R(1);
          } else {
            const int ib = i - NX;
            v = Aa[ib * NB + ib] - R(1);
            LQG_UNROLL for (int k = 0; k < NY; ++k) v -= K[ib * NY + k] * FAa[k * NB + ib];
            LQG_UNROLL for (int k = 0; k < NU; ++k) v += BK[ib * NU + k] * L[k * NB + ib];
          }
          op[Ops::F_OFF + i * M + i] = (OT)v;
        }
      LQG_UNROLL for (int i = 0; i < RR * O; ++i) op[Ops::U_OFF + i] = (OT)U2[i];
      {
        int e = 0;
        LQG_UNROLL for (int i = 0; i < O; ++i)
          LQG_UNROLL for (int j = 0; j <= i; ++j) op[Ops::L_OFF + (e++)] = (OT)Li[i * O + j];
      }
      op[Ops::H_OFF] = (OT)(hl + kLogNorm);
    }
    // ---- Sigma' = Fj[:, o:] C Fj[:, o:]^T + GG,  C = Srr - U2 U2^T                              system.py:223-230
    {
      R C[RR * RR];
      LQG_UNROLL for (int p = 0; p < RR; ++p)
        LQG_UNROLL for (int q = p; q < RR; ++q) {
          R v = Sg[(O + p) * M + O + q];
          LQG_UNROLL for (int j = 0; j < O; ++j) v -= U2[p * O + j] * U2[q * O + j];
          C[p * RR + q] = v;
          C[q * RR + p] = v;
        }
      R T1[M * RR];
      LQG_UNROLL for (int i = 0; i < M; ++i)
        LQG_UNROLL for (int q = 0; q < RR; ++q) {
          R v = R(0);
          LQG_UNROLL for (int p = 0; p < RR; ++p) v += Fj[i * M + O + p] * C[p * RR + q];
          T1[i * RR + q] = v;
        }
      LQG_UNROLL for (int i = 0; i < M; ++i)
        LQG_UNROLL for (int j = i; j < M; ++j) {
          R v = GG[i * M + j];
          LQG_UNROLL for (int q = 0; q < RR; ++q) v += T1[i * RR + q] * Fj[j * M + O + q];
          Sg[i * M + j] = v;
          Sg[j * M + i] = v;
        }
    }
    if (MAT && a.Sig.p) store_mat<R, M, M>(const_cast<R*>(a.Sig.p) + s * a.Sig.sb + t * a.Sig.st, a.Sig.sr, a.Sig.sc, Sg);
  }
  // ---- last row: only the density of x_T under the final predictive moments
  condition();
  if (FUSED) {
    innovate(a.T, true);
    if (a.ll) a.ll[s * a.ll_sb] = (R)acc;
  } else if (a.ops) {
    OT* op = reinterpret_cast<OT*>(a.ops) + ((long)s * (a.T + 1) + a.T) * Ops::N;
    {
      int e = 0;
      LQG_UNROLL for (int i = 0; i < O; ++i)
        LQG_UNROLL for (int j = 0; j <= i; ++j) op[Ops::L_OFF + (e++)] = (OT)Li[i * O + j];
    }
    op[Ops::H_OFF] = (OT)(hl + kLogNorm);
  }
}

// ===================================================================== per-trial sweep ===============
// reals per step of the MIXED mode's residual stream (ForwardArgs::ops_lo): the dense m x m image of the Fj - I block's residual
template <int M>
constexpr int hilo_len() { return (M * M + 3) / 4 * 4; }

template <typename R>
struct TrialArgs {
  DTraj<R> x;         // observed data
  DTraj<R> mu;        // optional output mu[B,N,T,m]
  R* ll;              // optional output
  long ll_sb, ll_sn;
  long n_trials;
  int T;
  // k_trial_sp<..., CKT > 0> (the reverse-mode sweep's forward pass, lqg_adjoint_trial_sp.hpp): c_{t-1} (the RR = M - ND
  // conditioned unobserved means of the step before) for every row t that starts a chunk of CKT, and for the last row T, kept at
  // tck[((sys * (nckt + 1) + rec) * RR + e) * npad + trial] — the mean state entering row t follows from it, x_{t-1} and the operator
  R* tck;
  long npad;
  int nckt;
  // MIXED mode (see ForwardArgs): when hl is set, a launch of k_trial_sp<..., HL> walks only the systems with (hl[sys] != 0) == HL
  const float* ops_lo;
  const int* hl;
};

// grid.x covers trials (LQG_BLOCK * TPL per block), grid.y = system.  The operator stream of the block's system is
// wave-uniform: `ops_all` is a direct `const __restrict__` kernel argument so that the compiler can prove the loads
// read-only and issue them as SCALAR loads (s_load_dwordx*, operands consumed straight from SGPRs); x is read with
// one trial per lane.
// STORE_MU is a template flag so that the log-likelihood instantiation carries no store code / output addressing.
// The data pointers advance incrementally (one 64-bit add per trial per step instead of a multiply-add chain), and
// in fp32 the per-step densities are summed in fp32 over chunks of kAccChunk steps before entering the fp64 total.
template <typename R, int M, int ND, int TPL, bool STORE_MU>
__global__ void __launch_bounds__(LQG_BLOCK) k_trial(const R* __restrict__ ops_all, const TrialArgs<R> a) {
  constexpr int O = ND, RR = M - ND;
  constexpr int kAccChunk = 8;
  using Ops = TrialOps<M, ND>;
  const long sys = blockIdx.y;
  const long n0 = (long)blockIdx.x * (LQG_BLOCK * TPL) + threadIdx.x;
  const R* __restrict__ op = ops_all + sys * (long)(a.T + 1) * Ops::N;
  const R* xr[TPL];
  bool live[TPL];
  R xprev[TPL][O], dO[TPL][O], muR[TPL][RR];   // observed mean = xprev + dO (deviation form, see k_forward)
  double acc[TPL];
  R part[TPL];
  LQG_UNROLL for (int k = 0; k < TPL; ++k) {
    long n = n0 + (long)k * LQG_BLOCK;
    live[k] = n < a.n_trials;
    n = live[k] ? n : (a.n_trials - 1);
    xr[k] = a.x.p + sys * a.x.sb + n * a.x.sn;
    LQG_UNROLL for (int i = 0; i < O; ++i) { xprev[k][i] = xr[k][i * a.x.sd]; dO[k][i] = R(0); }
    LQG_UNROLL for (int i = 0; i < RR; ++i) muR[k][i] = R(0);
    acc[k] = 0.0;
    part[k] = R(0);
  }
#if LQG_TRIAL_PREFETCH
  // software pipeline of the data stream, depth D = LQG_TRIAL_PREFETCH: while step t computes, rows t+1 .. t+D are in
  // flight (row indices clamped to T).  xq[k][0] is the row of the current step.
  constexpr int D = LQG_TRIAL_PREFETCH;
  R xq[TPL][D][O];
  LQG_UNROLL for (int k = 0; k < TPL; ++k)
    LQG_UNROLL for (int j = 0; j < D; ++j) {
      const long row = (j < a.T) ? j : a.T;
      LQG_UNROLL for (int i = 0; i < O; ++i) xq[k][j][i] = xr[k][row * a.x.st + i * a.x.sd];
    }
#endif
  // Double-buffered operator block: step t+1's block is requested (scalar loads) while step t computes.  Only when both
  // blocks fit comfortably in SGPRs (<= 32 dwords per block: n=2 models in fp32 — config 3: 5.1 -> 4.5 ms); larger blocks
  // spill and lose (m=5 in fp64: 3.2 -> 3.9 ms).
  constexpr bool PF = LQG_TRIAL_OPS_PREFETCH && (Ops::N * (int)(sizeof(R) / 4) <= 32);
  constexpr int NPF = PF ? Ops::N : 1;
  R opn[NPF], opc[NPF];
  if (PF) {
    LQG_UNROLL for (int i = 0; i < NPF; ++i) opn[i] = op[i];
  }
  for (int t = 0; t <= a.T; ++t) {
    if (PF) {
      LQG_UNROLL for (int i = 0; i < NPF; ++i) opc[i] = opn[i];
      const R* __restrict__ nx = op + ((t < a.T) ? Ops::N : 0);
      LQG_UNROLL for (int i = 0; i < NPF; ++i) opn[i] = nx[i];
    }
#define LQG_OP(i_) (PF ? opc[PF ? (i_) : 0] : op[i_])
    R Li[O * (O + 1) / 2];
    LQG_UNROLL for (int i = 0; i < O * (O + 1) / 2; ++i) Li[i] = LQG_OP(Ops::L_OFF + i);
    const R hlc = LQG_OP(Ops::H_OFF);
    const bool flush = ((t & (kAccChunk - 1)) == 0) || t == a.T;
    LQG_UNROLL for (int k = 0; k < TPL; ++k) {
      R xt[O], w[O];
#if LQG_TRIAL_PREFETCH
      LQG_UNROLL for (int i = 0; i < O; ++i) xt[i] = xq[k][0][i];
      LQG_UNROLL for (int j = 0; j + 1 < D; ++j)
        LQG_UNROLL for (int i = 0; i < O; ++i) xq[k][j][i] = xq[k][j + 1][i];
      {
        const long row = (t + D < a.T) ? (long)(t + D) : (long)a.T;
        LQG_UNROLL for (int i = 0; i < O; ++i) xq[k][D - 1][i] = xr[k][row * a.x.st + i * a.x.sd];
      }
#else
      LQG_UNROLL for (int i = 0; i < O; ++i) xt[i] = xr[k][i * a.x.sd];
      xr[k] += a.x.st;
#endif
      R zz = R(0);
      {
        int e = 0;
        LQG_UNROLL for (int i = 0; i < O; ++i) {
          R v = R(0);
          LQG_UNROLL for (int j = 0; j <= i; ++j) v += Li[e++] * ((xt[j] - xprev[k][j]) - dO[k][j]);
          w[i] = v;
          zz += v * v;
        }
      }
      if (t > 0) part[k] += R(0.5) * zz + hlc;
      if (flush) { acc[k] -= (double)part[k]; part[k] = R(0); }
      if (t < a.T) {
        R c[RR];
        LQG_UNROLL for (int p = 0; p < RR; ++p) {
          R v = muR[k][p];
          LQG_UNROLL for (int j = 0; j < O; ++j) v += LQG_OP(Ops::U_OFF + p * O + j) * w[j];
          c[p] = v;
        }
        R mn[M];
        LQG_UNROLL for (int i = 0; i < M; ++i) {
          R v = R(0);
          LQG_UNROLL for (int j = 0; j < O; ++j) v += LQG_OP(Ops::F_OFF + i * M + j) * xt[j];
          LQG_UNROLL for (int p = 0; p < RR; ++p) v += LQG_OP(Ops::F_OFF + i * M + O + p) * c[p];
          mn[i] = v;
        }
        // the stream holds Fj - I: mn is the DEVIATION of the new mean from [x_t ; c]
        LQG_UNROLL for (int i = 0; i < O; ++i) { dO[k][i] = mn[i]; xprev[k][i] = xt[i]; }
        LQG_UNROLL for (int p = 0; p < RR; ++p) muR[k][p] = c[p] + mn[O + p];
        if (STORE_MU && live[k]) {
          long n = n0 + (long)k * LQG_BLOCK;
          R* dst = const_cast<R*>(a.mu.p) + sys * a.mu.sb + n * a.mu.sn + (long)t * a.mu.st;
          LQG_UNROLL for (int i = 0; i < M; ++i) dst[i * a.mu.sd] = (i < O) ? xt[i] + mn[i] : muR[k][i - O];
        }
      }
    }
    op += Ops::N;
#undef LQG_OP
  }
  if (a.ll) {
    LQG_UNROLL for (int k = 0; k < TPL; ++k)
      if (live[k]) a.ll[sys * a.ll_sb + (n0 + (long)k * LQG_BLOCK) * a.ll_sn] = (R)acc[k];
  }
}

// ===================================================================== simulate ======================
template <typename R>
struct SimArgs {
  DView<R> aA, aB, aF;           // actor model used for the belief update
  DView<R> dA, dB, dF, dV, dW;   // true dynamics
  DView<R> L, l, K;              // gains (l may be null)
  DTraj<R> eps, eta;             // standard-normal draws; both null = drawn in-kernel (lqg_rng.hpp) from `seed`
  DView<R> x0, xh0;              // initial state / belief (may be null = 0)
  DTraj<R> xs, xh, ys, us;       // outputs (xh, ys, us may be null)
  long n_sys, n_trials;
  int T, nvd, nwd;
  unsigned long long seed;       // RNG variants only
};

// one (system, trial) per lane; trials are the fast index so x-loads of a shared system broadcast.
// RNG: the draws of system.py:100-105 are made in-kernel (counter-based Philox, lqg_rng.hpp) instead of being read from
// eps / eta — nothing but the trajectories crosses HBM.
template <typename R, int NX, int NB, int NU, int NY, bool RNG = false>
__global__ void __launch_bounds__(LQG_BLOCK) k_simulate(const SimArgs<R> a) {
  const long gid = blockIdx.x * (long)LQG_BLOCK + threadIdx.x;
  if (gid >= a.n_sys * a.n_trials) return;
  const long s = gid / a.n_trials, n = gid % a.n_trials;
  R x[NX], xh[NB];
  LQG_UNROLL for (int i = 0; i < NX; ++i) x[i] = a.x0.p ? a.x0.p[s * a.x0.sb + i * a.x0.sr] : R(0);
  LQG_UNROLL for (int i = 0; i < NB; ++i) xh[i] = a.xh0.p ? a.xh0.p[s * a.xh0.sb + i * a.xh0.sr] : R(0);
  auto out = [&](const DTraj<R>& v, int t) { return const_cast<R*>(v.p) + s * v.sb + n * v.sn + (long)t * v.st; };
  {
    R* d0 = out(a.xs, 0);
    LQG_UNROLL for (int i = 0; i < NX; ++i) d0[i * a.xs.sd] = x[i];
    if (a.xh.p) { R* d1 = out(a.xh, 0); LQG_UNROLL for (int i = 0; i < NB; ++i) d1[i * a.xh.sd] = xh[i]; }
  }
  for (int t = 0; t < a.T; ++t) {
    R L[NU * NB], K[NB * NY], u[NU];
    load_mat<R, NU, NB>(a.L.p + s * a.L.sb + t * a.L.st, a.L.sr, a.L.sc, L);
    load_mat<R, NB, NY>(a.K.p + s * a.K.sb + t * a.K.st, a.K.sr, a.K.sc, K);
    LQG_UNROLL for (int i = 0; i < NU; ++i) {               // u = L xhat + l     system.py:110
      R v = a.l.p ? a.l.p[s * a.l.sb + t * a.l.st + i * a.l.sr] : R(0);
      LQG_UNROLL for (int k = 0; k < NB; ++k) v += L[i * NB + k] * xh[k];
      u[i] = v;
    }
    R A[NX * NX], Bm[NX * NU], xn[NX];
    load_mat<R, NX, NX>(a.dA.p + s * a.dA.sb + t * a.dA.st, a.dA.sr, a.dA.sc, A);
    load_mat<R, NX, NU>(a.dB.p + s * a.dB.sb + t * a.dB.st, a.dB.sr, a.dB.sc, Bm);
    const R* ep = RNG ? nullptr : a.eps.p + s * a.eps.sb + n * a.eps.sn + (long)t * a.eps.st;
    const R* et = RNG ? nullptr : a.eta.p + s * a.eta.sb + n * a.eta.sn + (long)t * a.eta.st;
    LQG_UNROLL for (int i = 0; i < NX; ++i) {               // x = A x + B u + V eps   system.py:113-117
      R v = R(0);
      LQG_UNROLL for (int k = 0; k < NX; ++k) v += A[i * NX + k] * x[k];
      LQG_UNROLL for (int k = 0; k < NU; ++k) v += Bm[i * NU + k] * u[k];
      xn[i] = v;
    }
    {
      const R* Vp = a.dV.p + s * a.dV.sb + t * a.dV.st;
      if constexpr (RNG) {
        for (int k0 = 0; k0 < a.nvd; k0 += 4) {
          float z[4];
          rng::normal4(a.seed, s, n, (uint32_t)t, (uint32_t)(k0 >> 2), z);
          LQG_UNROLL for (int j = 0; j < 4; ++j)
            if (k0 + j < a.nvd) {
              LQG_UNROLL for (int i = 0; i < NX; ++i) xn[i] += Vp[i * a.dV.sr + (k0 + j) * a.dV.sc] * (R)z[j];
            }
        }
      } else {
        for (int k = 0; k < a.nvd; ++k) {
          R e = ep[k * a.eps.sd];
          LQG_UNROLL for (int i = 0; i < NX; ++i) xn[i] += Vp[i * a.dV.sr + k * a.dV.sc] * e;
        }
      }
    }
    LQG_UNROLL for (int i = 0; i < NX; ++i) x[i] = xn[i];
    R F[NY * NX], y[NY];
    load_mat<R, NY, NX>(a.dF.p + s * a.dF.sb + t * a.dF.st, a.dF.sr, a.dF.sc, F);
    LQG_UNROLL for (int i = 0; i < NY; ++i) {               // y = F x + W eta          system.py:120
      R v = R(0);
      LQG_UNROLL for (int k = 0; k < NX; ++k) v += F[i * NX + k] * x[k];
      y[i] = v;
    }
    {
      const R* Wp = a.dW.p + s * a.dW.sb + t * a.dW.st;
      if constexpr (RNG) {
        for (int k0 = 0; k0 < a.nwd; k0 += 4) {
          float z[4];
          rng::normal4(a.seed, s, n, (uint32_t)t, rng::kEtaBlock + (uint32_t)(k0 >> 2), z);
          LQG_UNROLL for (int j = 0; j < 4; ++j)
            if (k0 + j < a.nwd) {
              LQG_UNROLL for (int i = 0; i < NY; ++i) y[i] += Wp[i * a.dW.sr + (k0 + j) * a.dW.sc] * (R)z[j];
            }
        }
      } else {
        for (int k = 0; k < a.nwd; ++k) {
          R e = et[k * a.eta.sd];
          LQG_UNROLL for (int i = 0; i < NY; ++i) y[i] += Wp[i * a.dW.sr + k * a.dW.sc] * e;
        }
      }
    }
    R Aa[NB * NB], Ba[NB * NU], Fa[NY * NB], xp[NB], inn[NY];
    load_mat<R, NB, NB>(a.aA.p + s * a.aA.sb + t * a.aA.st, a.aA.sr, a.aA.sc, Aa);
    load_mat<R, NB, NU>(a.aB.p + s * a.aB.sb + t * a.aB.st, a.aB.sr, a.aB.sc, Ba);
    load_mat<R, NY, NB>(a.aF.p + s * a.aF.sb + t * a.aF.st, a.aF.sr, a.aF.sc, Fa);
    LQG_UNROLL for (int i = 0; i < NB; ++i) {               // x_pred = A xhat + B u     system.py:123
      R v = R(0);
      LQG_UNROLL for (int k = 0; k < NB; ++k) v += Aa[i * NB + k] * xh[k];
      LQG_UNROLL for (int k = 0; k < NU; ++k) v += Ba[i * NU + k] * u[k];
      xp[i] = v;
    }
    LQG_UNROLL for (int i = 0; i < NY; ++i) {
      R v = y[i];
      LQG_UNROLL for (int k = 0; k < NB; ++k) v -= Fa[i * NB + k] * xp[k];
      inn[i] = v;
    }
    LQG_UNROLL for (int i = 0; i < NB; ++i) {               // xhat = x_pred + K (y - F x_pred)  system.py:124
      R v = xp[i];
      LQG_UNROLL for (int k = 0; k < NY; ++k) v += K[i * NY + k] * inn[k];
      xh[i] = v;
    }
    R* d0 = out(a.xs, t + 1);
    LQG_UNROLL for (int i = 0; i < NX; ++i) d0[i * a.xs.sd] = x[i];
    if (a.xh.p) { R* d1 = out(a.xh, t + 1); LQG_UNROLL for (int i = 0; i < NB; ++i) d1[i * a.xh.sd] = xh[i]; }
    if (a.ys.p) { R* d2 = out(a.ys, t); LQG_UNROLL for (int i = 0; i < NY; ++i) d2[i * a.ys.sd] = y[i]; }
    if (a.us.p) { R* d3 = out(a.us, t); LQG_UNROLL for (int i = 0; i < NU; ++i) d3[i * a.us.sd] = u[i]; }
  }
}

// ===================================================================== Gaussian log-density ==========
template <typename R>
struct LogprobArgs {
  DTraj<R> value, mu;
  DView<R> Sig;
  R* out;
  long out_sb, out_sn, n_sys, n_trials;
  int T;
};

template <typename R, int KD>
__global__ void __launch_bounds__(LQG_BLOCK) k_gaussian_logprob(const LogprobArgs<R> a) {
  const long n = blockIdx.x * (long)LQG_BLOCK + threadIdx.x;
  const long s = blockIdx.y;
  if (n >= a.n_trials) return;
  double acc = 0.0;
  const R kLogNorm = R(0.5 * KD * 1.8378770664093453);
  for (int t = 0; t < a.T; ++t) {
    R Sk[KD * KD], Lc[KD * KD], dinv[KD];
    const R* sp = a.Sig.p + s * a.Sig.sb + t * a.Sig.st;
    LQG_UNROLL for (int i = 0; i < KD; ++i)
      LQG_UNROLL for (int j = 0; j <= i; ++j) {
        Sk[i * KD + j] = sp[i * a.Sig.sr + j * a.Sig.sc];
        Sk[j * KD + i] = Sk[i * KD + j];
      }
    chol_lower<R, KD>(Sk, Lc, dinv);
    const R* vp = a.value.p + s * a.value.sb + n * a.value.sn + (long)t * a.value.st;
    const R* mp = a.mu.p + s * a.mu.sb + n * a.mu.sn + (long)t * a.mu.st;
    R z[KD], zz = R(0), pd = R(1);
    LQG_UNROLL for (int i = 0; i < KD; ++i) {
      R v = vp[i * a.value.sd] - mp[i * a.mu.sd];
      LQG_UNROLL for (int j = 0; j < i; ++j) v -= Lc[i * KD + j] * z[j];
      z[i] = v * dinv[i];
      zz += z[i] * z[i];
      pd *= dinv[i];
    }
    acc -= (double)(R(0.5) * zz - log_<R>(pd) + kLogNorm);
  }
  a.out[s * a.out_sb + n * a.out_sn] = (R)acc;
}

// ===================================================================== sum over trials ===============
// out[b] = sum_n ll[b, n] in fp64 with a FIXED reduction tree (bitwise reproducible run to run, no atomics):
// stage 1: grid (chunks, B), each block reduces one chunk of kSumChunk trials into part[b][chunk];
// stage 2: grid (B), one block reduces the chunk partials of its system.
constexpr int kSumChunk = 4096;

template <typename T>
LQG_DEV double block_sum_256(double acc, double* smem) {
  LQG_UNROLL for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63) == 0) smem[threadIdx.x >> 6] = acc;
  __syncthreads();
  return (smem[0] + smem[1]) + (smem[2] + smem[3]);
}

template <typename R>
__global__ void __launch_bounds__(256) k_sum_trials(const R* ll, long n_trials, long sb, long sn, double* part,
                                                    long n_chunks) {
  __shared__ double smem[4];
  const long b = blockIdx.y, c = blockIdx.x;
  const long lo = c * kSumChunk, hi = (lo + kSumChunk < n_trials) ? lo + kSumChunk : n_trials;
  double acc = 0.0;
  for (long n = lo + threadIdx.x; n < hi; n += 256) acc += (double)ll[b * sb + n * sn];
  const double tot = block_sum_256<R>(acc, smem);
  if (threadIdx.x == 0) part[b * n_chunks + c] = tot;
}

template <int UNUSED = 0>   // template only so that the definition may live in this header (one instance per TU)
__global__ void __launch_bounds__(256) k_sum_partials(const double* part, long n_chunks, double* out) {
  __shared__ double smem[4];
  const long b = blockIdx.x;
  double acc = 0.0;
  for (long c = threadIdx.x; c < n_chunks; c += 256) acc += part[b * n_chunks + c];
  const double tot = block_sum_256<double>(acc, smem);
  if (threadIdx.x == 0) out[b] = tot;
}

}  // namespace lqg
