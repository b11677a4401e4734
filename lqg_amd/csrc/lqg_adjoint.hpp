// lqg_adjoint.hpp — reverse-mode sweep of the LQG log-likelihood as HIP kernels for gfx950 (MI355X):
// d(sum_n g[n] ll[n]) / d(spec matrices) through all three scans of the path, what the reference obtains from
// jax.grad of System.log_likelihood (lqg/optim.py:142-147, lqg/infer/utils.py:18,37-39, notebooks/Tutorial.ipynb
// cell 40).  SURVEY.md §8(f) rank 1.  CPU restatement with the same structure: oracle/lqg_adjoint_np.py.
//
// Mapping: ONE (system, trial) PAIR PER LANE, matrices in registers, exactly as the forward kernels.  Four sweeps,
// talking through a [t][element][lane] scratch (coalesced across lanes):
//   k_adj_riccati      t = T-1..0   lqr.py:16-42         keeps S_{t+1} and L_t
//   k_adj_forward      t = 0..T-1   kf.py:6-21 + system.py:167-248   keeps the state BEFORE each step (P_t, Sigma_t,
//                                   mu_t) and returns the log-likelihood value
//   k_adj_reverse      t = T-1..0   recomputes the step from the kept state, then the adjoints of the log-density,
//                                   the moment recursion, the joint system (-> Lbar_t over L_t's slot, Kbar_t) and
//                                   the Kalman step; accumulates the bars of the ten time-invariant spec matrices
//   k_adj_riccati_rev  t = 0..T-1   adjoint of the Riccati recursion (it ran backward), consumes Lbar_t
// Noise enters through the Gram matrices VV = V V', WW = W W' (the host chains VVbar, WWbar to V, W).  Adjoints of
// the symmetric carries (Sigma, P, S) are symmetrised every step: the antisymmetric part is invisible to symmetric
// perturbations but grows geometrically and destroys the result by cancellation otherwise.  The eigenvalue-floor
// shift (lqr.py:27-28) is held constant.  q, r, qf, P get no gradient (the likelihood ignores the affine gain l).
// TI = true: time-invariant specs, loaded once, ONE accumulated bar per matrix (slab 0 of the output).  TI = false: specs
// loaded every step, one bar per matrix PER STEP (slab t); Qf and Sigma0 bars always go to slab 0.
#pragma once
#include "lqg_small.hpp"

#ifndef LQG_BLOCK
#define LQG_BLOCK 64
#endif

namespace lqg {
namespace adj {

// ---- small dense helpers (row-major, compile-time extents, accumulate forms) ---------------------------------
template <typename R, int N>
LQG_DEV void zero(R (&C)[N]) {
  LQG_UNROLL for (int i = 0; i < N; ++i) C[i] = R(0);
}
template <typename R, int N>
LQG_DEV void copy(const R (&A)[N], R (&C)[N]) {
  LQG_UNROLL for (int i = 0; i < N; ++i) C[i] = A[i];
}
// C[M,N] += alpha A[M,K] B[K,N]
template <typename R, int M, int K, int N>
LQG_DEV void mm_acc(const R (&A)[M * K], const R (&B)[K * N], R (&C)[M * N], R alpha = R(1)) {
  LQG_UNROLL for (int i = 0; i < M; ++i)
    LQG_UNROLL for (int j = 0; j < N; ++j) {
      R acc = R(0);
      LQG_UNROLL for (int k = 0; k < K; ++k) acc += A[i * K + k] * B[k * N + j];
      C[i * N + j] += alpha * acc;
    }
}
// C[M,N] += alpha A[K,M]^T B[K,N]
template <typename R, int M, int K, int N>
LQG_DEV void mtm_acc(const R (&A)[K * M], const R (&B)[K * N], R (&C)[M * N], R alpha = R(1)) {
  LQG_UNROLL for (int i = 0; i < M; ++i)
    LQG_UNROLL for (int j = 0; j < N; ++j) {
      R acc = R(0);
      LQG_UNROLL for (int k = 0; k < K; ++k) acc += A[k * M + i] * B[k * N + j];
      C[i * N + j] += alpha * acc;
    }
}
// C[M,N] += alpha A[M,K] B[N,K]^T
template <typename R, int M, int K, int N>
LQG_DEV void mmt_acc(const R (&A)[M * K], const R (&B)[N * K], R (&C)[M * N], R alpha = R(1)) {
  LQG_UNROLL for (int i = 0; i < M; ++i)
    LQG_UNROLL for (int j = 0; j < N; ++j) {
      R acc = R(0);
      LQG_UNROLL for (int k = 0; k < K; ++k) acc += A[i * K + k] * B[j * K + k];
      C[i * N + j] += alpha * acc;
    }
}
template <typename R, int N>
LQG_DEV void symmetrise(R (&A)[N * N]) {
  LQG_UNROLL for (int i = 0; i < N; ++i)
    LQG_UNROLL for (int j = i + 1; j < N; ++j) {
      R v = R(0.5) * (A[i * N + j] + A[j * N + i]);
      A[i * N + j] = v;
      A[j * N + i] = v;
    }
}
// inverse of a symmetric positive-definite matrix; returns sum_i log(1 / Lc_ii) = -0.5 log det A
template <typename R, int N>
LQG_DEV R spd_inverse(const R (&A)[N * N], R (&Ainv)[N * N]) {
  R Lc[N * N], dinv[N], Li[N * N];
  chol_lower<R, N>(A, Lc, dinv);
  tri_inverse_lower<R, N>(Lc, dinv, Li);
  spd_inverse_from_tri<R, N>(Li, Ainv);
  R s = R(0);
  LQG_UNROLL for (int i = 0; i < N; ++i) s += log_<R>(dinv[i]);
  return s;
}
// block copy out of / accumulate into a row-major matrix with LD columns
template <typename R, int LD, int ROWS, int COLS, int NSRC>
LQG_DEV void get_block(const R (&S)[NSRC], int r0, int c0, R (&D)[ROWS * COLS]) {
  LQG_UNROLL for (int i = 0; i < ROWS; ++i)
    LQG_UNROLL for (int j = 0; j < COLS; ++j) D[i * COLS + j] = S[(r0 + i) * LD + c0 + j];
}
template <typename R, int LD, int ROWS, int COLS, int NDST>
LQG_DEV void set_block(const R (&S)[ROWS * COLS], int r0, int c0, R (&D)[NDST]) {
  LQG_UNROLL for (int i = 0; i < ROWS; ++i)
    LQG_UNROLL for (int j = 0; j < COLS; ++j) D[(r0 + i) * LD + c0 + j] = S[i * COLS + j];
}
// packed lower-triangle store / load of a symmetric matrix at p[k * ld]
template <typename R, int N>
LQG_DEV void store_tri(R* __restrict__ p, long ld, const R (&A)[N * N]) {
  int k = 0;
  LQG_UNROLL for (int i = 0; i < N; ++i)
    LQG_UNROLL for (int j = 0; j <= i; ++j) p[(k++) * ld] = A[i * N + j];
}
template <typename R, int N>
LQG_DEV void load_tri(const R* __restrict__ p, long ld, R (&A)[N * N]) {
  int k = 0;
  LQG_UNROLL for (int i = 0; i < N; ++i)
    LQG_UNROLL for (int j = 0; j <= i; ++j) {
      R v = p[(k++) * ld];
      A[i * N + j] = v;
      A[j * N + i] = v;
    }
}
template <typename R, int N>
LQG_DEV void store_flat(R* __restrict__ p, long ld, const R (&A)[N]) {
  LQG_UNROLL for (int i = 0; i < N; ++i) p[i * ld] = A[i];
}
template <typename R, int N>
LQG_DEV void load_flat(const R* __restrict__ p, long ld, R (&A)[N]) {
  LQG_UNROLL for (int i = 0; i < N; ++i) A[i] = p[i * ld];
}

// ---- layout of the scratch and of the gradient output ----------------------------------------------------------
template <int NX, int NB, int NU, int NY>
struct Layout {
  static constexpr int M = NX + NB;
  static constexpr int TRI_B = NB * (NB + 1) / 2, TRI_M = M * (M + 1) / 2;
  // per (step, lane) scratch reals
  static constexpr int S_OFF = 0, L_OFF = S_OFF + TRI_B, P_OFF = L_OFF + NU * NB, SIG_OFF = P_OFF + TRI_B,
                       MU_OFF = SIG_OFF + TRI_M, STEP = MU_OFF + M;
  // gradient output elements (each a full row-major matrix), [element][lane]
  static constexpr int DA = 0, DB = DA + NX * NX, DF = DB + NX * NU, DVV = DF + NY * NX, DWW = DVV + NX * NX,
                       AA = DWW + NY * NY, AB = AA + NB * NB, AF = AB + NB * NU, AVV = AF + NY * NB,
                       AWW = AVV + NB * NB, AQ = AWW + NY * NY, AR = AQ + NB * NB, AQF = AR + NU * NU,
                       AS0 = AQF + NB * NB, AA2 = AS0 + NB * NB, AB2 = AA2 + NB * NB, TOTAL = AB2 + NB * NU;
};

template <typename R>
struct AdjArgs {
  DView<R> Q, Qf, P, Rm, A, B, F, V, W;   // actor
  DView<R> dA, dB, dF, dV, dW;            // dynamics
  DView<R> Sigma0;                        // may be null (default V V')
  DTraj<R> x;
  const R* g;                             // upstream weights d(objective)/d ll[s, n]; null = 1
  long g_sb, g_sn;
  R* ll;                                  // value out, may be null
  long ll_sb, ll_sn;
  R* ws;                                  // [T][Layout::STEP][ld]
  R* out;                                 // [Layout::TOTAL][ld]
  long ld, n_lanes, n_trials;
  int T, nva, nwa, nvd, nwd;
  R eps;
};

// ---- the time-invariant spec of one lane ------------------------------------------------------------------------
template <typename R, int NX, int NB, int NU, int NY>
struct Spec {
  R Ad[NX * NX], Bd[NX * NU], Fd[NY * NX], VVd[NX * NX], WWd[NY * NY];
  R Aa[NB * NB], Ba[NB * NU], Fa[NY * NB], VVa[NB * NB], WWa[NY * NY];
  LQG_DEV void load(const AdjArgs<R>& a, long s, long t = 0) {
    load_mat<R, NX, NX>(a.dA.p + s * a.dA.sb + t * a.dA.st, a.dA.sr, a.dA.sc, Ad);
    load_mat<R, NX, NU>(a.dB.p + s * a.dB.sb + t * a.dB.st, a.dB.sr, a.dB.sc, Bd);
    load_mat<R, NY, NX>(a.dF.p + s * a.dF.sb + t * a.dF.st, a.dF.sr, a.dF.sc, Fd);
    load_gram<R, NX>(a.dV.p + s * a.dV.sb + t * a.dV.st, a.dV.sr, a.dV.sc, a.nvd, VVd);
    load_gram<R, NY>(a.dW.p + s * a.dW.sb + t * a.dW.st, a.dW.sr, a.dW.sc, a.nwd, WWd);
    load_mat<R, NB, NB>(a.A.p + s * a.A.sb + t * a.A.st, a.A.sr, a.A.sc, Aa);
    load_mat<R, NB, NU>(a.B.p + s * a.B.sb + t * a.B.st, a.B.sr, a.B.sc, Ba);
    load_mat<R, NY, NB>(a.F.p + s * a.F.sb + t * a.F.st, a.F.sr, a.F.sc, Fa);
    load_gram<R, NB>(a.V.p + s * a.V.sb + t * a.V.st, a.V.sr, a.V.sc, a.nva, VVa);
    load_gram<R, NY>(a.W.p + s * a.W.sb + t * a.W.st, a.W.sr, a.W.sc, a.nwa, WWa);
  }
};

// ---- one Riccati step from S = S_{t+1}                                                      lqr.py:22-33 -------
template <typename R, int NB, int NU>
struct RicStep {
  R H[NU * NU], G[NU * NB], Hti[NU * NU], L[NU * NB], SA[NB * NB], SB[NB * NU];
  LQG_DEV void compute(const R (&S)[NB * NB], const R (&A)[NB * NB], const R (&B)[NB * NU], const R (&Rm)[NU * NU],
                       const R (&P)[NU * NB], R eps) {
    zero<R, NB * NB>(SA);
    zero<R, NB * NU>(SB);
    mm_acc<R, NB, NB, NB>(S, A, SA);
    mm_acc<R, NB, NB, NU>(S, B, SB);
    copy<R, NU * NU>(Rm, H);
    mtm_acc<R, NU, NB, NU>(B, SB, H);                 // H = R + B' S B
    symmetrise<R, NU>(H);
    copy<R, NU * NB>(P, G);
    mtm_acc<R, NU, NB, NB>(B, SA, G);                 // G = P + B' S A
    R shift = eps - min_eig_sym<R, NU>(H);
    shift = (shift > R(0)) ? shift : R(0);
    R Ht[NU * NU];
    copy<R, NU * NU>(H, Ht);
    LQG_UNROLL for (int i = 0; i < NU; ++i) Ht[i * NU + i] += shift;
    spd_inverse<R, NU>(Ht, Hti);
    zero<R, NU * NB>(L);
    mm_acc<R, NU, NU, NB>(Hti, G, L, R(-1));          // L = -Ht^-1 G
  }
  // S_t = Q + A' S A + L' H L + L' G + G' L
  LQG_DEV void next(const R (&Q)[NB * NB], const R (&A)[NB * NB], R (&S)[NB * NB]) {
    R HL[NU * NB];
    copy<R, NU * NB>(G, HL);
    mm_acc<R, NU, NU, NB>(H, L, HL);                  // H L + G
    copy<R, NB * NB>(Q, S);
    mtm_acc<R, NB, NB, NB>(A, SA, S);
    mtm_acc<R, NB, NU, NB>(L, HL, S);                 // L'(H L + G)
    mtm_acc<R, NB, NU, NB>(G, L, S);                  // G' L
    symmetrise<R, NB>(S);
  }
};

template <typename R, int NX, int NB, int NU, int NY, int ND, bool TI>
__global__ void __launch_bounds__(LQG_BLOCK, 1) k_adj_riccati(const AdjArgs<R> a) {
  using Lay = Layout<NX, NB, NU, NY>;
  const long i = blockIdx.x * (long)LQG_BLOCK + threadIdx.x;
  if (i >= a.n_lanes) return;
  const long s = i / a.n_trials;
  R A[NB * NB], B[NB * NU], Q[NB * NB], Rm[NU * NU], P[NU * NB], S[NB * NB];
  auto load_step = [&](long t) {
    load_mat<R, NB, NB>(a.A.p + s * a.A.sb + t * a.A.st, a.A.sr, a.A.sc, A);
    load_mat<R, NB, NU>(a.B.p + s * a.B.sb + t * a.B.st, a.B.sr, a.B.sc, B);
    load_sym<R, NB>(a.Q.p + s * a.Q.sb + t * a.Q.st, a.Q.sr, a.Q.sc, Q);
    load_sym<R, NU>(a.Rm.p + s * a.Rm.sb + t * a.Rm.st, a.Rm.sr, a.Rm.sc, Rm);
    zero<R, NU * NB>(P);
    if (a.P.p) load_mat<R, NU, NB>(a.P.p + s * a.P.sb + t * a.P.st, a.P.sr, a.P.sc, P);
  };
  if (TI) load_step(0);
  load_sym<R, NB>(a.Qf.p + s * a.Qf.sb, a.Qf.sr, a.Qf.sc, S);
  RicStep<R, NB, NU> st;
  for (int t = a.T - 1; t >= 0; --t) {
    if (!TI) load_step(t);
    R* w = a.ws + ((long)t * Lay::STEP) * a.ld + i;
    store_tri<R, NB>(w + Lay::S_OFF * a.ld, a.ld, S);
    st.compute(S, A, B, Rm, P, a.eps);
    store_flat<R, NU * NB>(w + Lay::L_OFF * a.ld, a.ld, st.L);
    st.next(Q, A, S);
  }
}

// ---- one forward step from the state before it (shared by the forward and the reverse sweep) -------------------
template <typename R, int NX, int NB, int NU, int NY, int ND>
struct FwdStep {
  static constexpr int M = NX + NB;
  R Pp[NB * NB], Gi[NY * NY], K[NB * NY], Y[NB * NX], D[NY * NU], Z[NY * NB], YVV[NB * NX], KWW[NB * NY];
  static constexpr int RR = M - ND;                   // unobserved part of the joint state
  R F[M * M], GG[M * M], Fr[M * RR], Wm[RR * ND], N[ND * ND], r[ND], av[ND], c[M], Crr[RR * RR], FCr[M * RR], FPp[NY * NB];
  R mu1[M], Sig1[M * M];

  // Kalman step + joint system (independent of the moment state)
  LQG_DEV void system(const Spec<R, NX, NB, NU, NY>& sp, const R (&P0)[NB * NB], const R (&L)[NU * NB]) {
    R AP[NB * NB];
    zero<R, NB * NB>(AP);
    mm_acc<R, NB, NB, NB>(sp.Aa, P0, AP);
    copy<R, NB * NB>(sp.VVa, Pp);
    mmt_acc<R, NB, NB, NB>(AP, sp.Aa, Pp);            // Pp = A P A' + V V'          kf.py:10
    symmetrise<R, NB>(Pp);
    zero<R, NY * NB>(FPp);
    mm_acc<R, NY, NB, NB>(sp.Fa, Pp, FPp);
    R Gm[NY * NY];
    copy<R, NY * NY>(sp.WWa, Gm);
    mmt_acc<R, NY, NB, NY>(FPp, sp.Fa, Gm);           // F Pp F' + W W'              kf.py:11
    symmetrise<R, NY>(Gm);
    spd_inverse<R, NY>(Gm, Gi);
    zero<R, NB * NY>(K);
    mtm_acc<R, NB, NY, NY>(FPp, Gi, K);               // K = Pp F' G^-1              kf.py:12
    zero<R, NB * NX>(Y);
    mm_acc<R, NB, NY, NX>(K, sp.Fd, Y);
    zero<R, NY * NU>(D);
    mm_acc<R, NY, NX, NU>(sp.Fd, sp.Bd, D);
    mm_acc<R, NY, NB, NU>(sp.Fa, sp.Ba, D, R(-1));    // D = Fd Bd - Fa Ba
    zero<R, NY * NB>(Z);
    mm_acc<R, NY, NU, NB>(D, L, Z);
    mm_acc<R, NY, NB, NB>(sp.Fa, sp.Aa, Z, R(-1));    // Z = D L - Fa Aa
    R blk12[NX * NB], blk21[NB * NX], blk22[NB * NB];
    zero<R, NX * NB>(blk12);
    mm_acc<R, NX, NU, NB>(sp.Bd, L, blk12);
    zero<R, NB * NX>(blk21);
    mm_acc<R, NB, NX, NX>(Y, sp.Ad, blk21);
    copy<R, NB * NB>(sp.Aa, blk22);
    mm_acc<R, NB, NU, NB>(sp.Ba, L, blk22);
    mm_acc<R, NB, NY, NB>(K, Z, blk22);
    set_block<R, M, NX, NX>(sp.Ad, 0, 0, F);          // system.py:167-181
    set_block<R, M, NX, NB>(blk12, 0, NX, F);
    set_block<R, M, NB, NX>(blk21, NX, 0, F);
    set_block<R, M, NB, NB>(blk22, NX, NX, F);
    zero<R, NB * NX>(YVV);
    mm_acc<R, NB, NX, NX>(Y, sp.VVd, YVV);
    zero<R, NB * NY>(KWW);
    mm_acc<R, NB, NY, NY>(K, sp.WWd, KWW);
    R g22[NB * NB];
    zero<R, NB * NB>(g22);
    mmt_acc<R, NB, NX, NB>(YVV, Y, g22);
    mmt_acc<R, NB, NY, NB>(KWW, K, g22);
    symmetrise<R, NB>(g22);
    set_block<R, M, NX, NX>(sp.VVd, 0, 0, GG);        // G_j G_j'                    system.py:194-202
    set_block<R, M, NB, NX>(YVV, NX, 0, GG);
    LQG_UNROLL for (int p = 0; p < NX; ++p)
      LQG_UNROLL for (int q = 0; q < NB; ++q) GG[p * M + NX + q] = YVV[q * NX + p];
    set_block<R, M, NB, NB>(g22, NX, NX, GG);
  }
  // P after the update: Pp - K F Pp                                                  kf.py:14
  LQG_DEV void kalman_update(R (&P1)[NB * NB]) const {
    copy<R, NB * NB>(Pp, P1);
    mm_acc<R, NB, NY, NB>(K, FPp, P1, R(-1));
    symmetrise<R, NB>(P1);
  }
  // conditioning on x_t and propagation                                              system.py:219-230
  // The observed rows of the conditional mean / covariance are x_t and 0 exactly, so only the unobserved block is
  // formed, through Wm = S_ro S_oo^-1 (no Soo - Soo Soo^-1 Soo cancellation: the point-mass model has cond(S_oo) ~ 1e8).
  // DEVIATION FORM (as the forward kernels): the observed entries of `mu` hold dO = mu_o - x_{t-1}, the
  // predicted mean relative to the PREVIOUS data row (xprev), so that the innovation is (x_t - x_{t-1}) - dO — the data
  // difference is exact in floating point and nothing large is cancelled; mu1's observed entries come out relative to
  // x_t.  Derivatives are unaffected (d mu_o / d dO = 1): only the evaluation of r changes.  Without it the fp32
  // gradients of the fully observed point-mass model (cond(S_oo) ~ 5e8) were useless.
  template <bool FULL>
  LQG_DEV void moments(const R (&Sig)[M * M], const R (&mu)[M], const R (&xt)[ND], const R (&xprev)[ND]) {
    R Soo[ND * ND], Sro[RR * ND], Srr[RR * RR];
    get_block<R, M, ND, ND>(Sig, 0, 0, Soo);
    get_block<R, M, RR, ND>(Sig, ND, 0, Sro);
    get_block<R, M, RR, RR>(Sig, ND, ND, Srr);
    get_block<R, M, M, RR>(F, 0, ND, Fr);
    spd_inverse<R, ND>(Soo, N);
    LQG_UNROLL for (int k = 0; k < ND; ++k) r[k] = (xt[k] - xprev[k]) - mu[k];
    LQG_UNROLL for (int k = 0; k < ND; ++k) {
      R acc = R(0);
      LQG_UNROLL for (int j = 0; j < ND; ++j) acc += N[k * ND + j] * r[j];
      av[k] = acc;
    }
    zero<R, RR * ND>(Wm);
    mm_acc<R, RR, ND, ND>(Sro, N, Wm);
    LQG_UNROLL for (int k = 0; k < ND; ++k) c[k] = xt[k];
    LQG_UNROLL for (int k = 0; k < RR; ++k) {
      R acc = mu[ND + k];
      LQG_UNROLL for (int j = 0; j < ND; ++j) acc += Wm[k * ND + j] * r[j];
      c[ND + k] = acc;
    }
    copy<R, RR * RR>(Srr, Crr);
    mmt_acc<R, RR, ND, RR>(Wm, Sro, Crr, R(-1));      // C_rr = S_rr - Wm S_or
    symmetrise<R, RR>(Crr);
    zero<R, M * RR>(FCr);
    mm_acc<R, M, RR, RR>(Fr, Crr, FCr);
    LQG_UNROLL for (int k = 0; k < M; ++k) {
      R acc = R(0);
      LQG_UNROLL for (int j = 0; j < M; ++j) acc += ((k < ND && k == j) ? F[k * M + j] - R(1) : F[k * M + j]) * c[j];
      mu1[k] = acc;                                   // observed rows: relative to x_t
    }
    if (FULL) {
      copy<R, M * M>(GG, Sig1);
      mmt_acc<R, M, RR, M>(FCr, Fr, Sig1);            // Sig1 = F C F' + G G'
      symmetrise<R, M>(Sig1);
    } else {                                          // only the observed block is needed by the reverse sweep
      LQG_UNROLL for (int p = 0; p < ND; ++p)
        LQG_UNROLL for (int q = 0; q < ND; ++q) {
          R acc = GG[p * M + q];
          LQG_UNROLL for (int k = 0; k < RR; ++k) acc += FCr[p * RR + k] * Fr[q * RR + k];
          Sig1[p * M + q] = acc;
        }
    }
  }
};

template <typename R, int NX, int NB, int NU, int NY, int ND, bool TI>
__global__ void __launch_bounds__(LQG_BLOCK, 1) k_adj_forward(const AdjArgs<R> a) {
  using Lay = Layout<NX, NB, NU, NY>;
  constexpr int M = NX + NB;
  const long i = blockIdx.x * (long)LQG_BLOCK + threadIdx.x;
  if (i >= a.n_lanes) return;
  const long s = i / a.n_trials, n = i % a.n_trials;
  Spec<R, NX, NB, NU, NY> sp;
  sp.load(a, s);
  R P[NB * NB], Sig[M * M], mu[M], L[NU * NB], xt[ND];
  if (a.Sigma0.p) load_sym<R, NB>(a.Sigma0.p + s * a.Sigma0.sb, a.Sigma0.sr, a.Sigma0.sc, P);
  else copy<R, NB * NB>(sp.VVa, P);
  const R* xp = a.x.p + s * a.x.sb + n * a.x.sn;
  R xprev[ND];
  zero<R, M>(mu);                                                    // mu0 = [x[0], 0...] (system.py:211): dO = 0
  LQG_UNROLL for (int k = 0; k < ND; ++k) xt[k] = xp[k * a.x.sd];
  LQG_UNROLL for (int k = 0; k < ND; ++k) xprev[k] = xt[k];
  double ll = 0.0;
  FwdStep<R, NX, NB, NU, NY, ND> f;
  for (int t = 0; t < a.T; ++t) {
    if (!TI && t > 0) sp.load(a, s, t);
    R* w = a.ws + ((long)t * Lay::STEP) * a.ld + i;
    load_flat<R, NU * NB>(w + Lay::L_OFF * a.ld, a.ld, L);
    f.system(sp, P, L);
    if (t == 0) copy<R, M * M>(f.GG, Sig);                           // system.py:212
    store_tri<R, NB>(w + Lay::P_OFF * a.ld, a.ld, P);
    store_tri<R, M>(w + Lay::SIG_OFF * a.ld, a.ld, Sig);
    store_flat<R, M>(w + Lay::MU_OFF * a.ld, a.ld, mu);
    f.template moments<true>(Sig, mu, xt, xprev);
    f.kalman_update(P);
    copy<R, M * M>(f.Sig1, Sig);
    copy<R, M>(f.mu1, mu);
    LQG_UNROLL for (int k = 0; k < ND; ++k) xprev[k] = xt[k];
    LQG_UNROLL for (int k = 0; k < ND; ++k) xt[k] = xp[(long)(t + 1) * a.x.st + k * a.x.sd];
    R Soo[ND * ND], Ni[ND * ND], e[ND];
    get_block<R, M, ND, ND>(Sig, 0, 0, Soo);
    R nhl = spd_inverse<R, ND>(Soo, Ni);                             // -0.5 log det
    R q = R(0);
    LQG_UNROLL for (int k = 0; k < ND; ++k) e[k] = (xt[k] - xprev[k]) - mu[k];
    LQG_UNROLL for (int p = 0; p < ND; ++p)
      LQG_UNROLL for (int k = 0; k < ND; ++k) q += e[p] * Ni[p * ND + k] * e[k];
    ll += (double)(nhl - R(0.5) * q) - 0.5 * ND * 1.8378770664093453;  // system.py:244-248
  }
  if (a.ll) a.ll[s * a.ll_sb + n * a.ll_sn] = (R)ll;
}

template <typename R, int NX, int NB, int NU, int NY, int ND, bool TI>
__global__ void __launch_bounds__(LQG_BLOCK, 1) k_adj_reverse(const AdjArgs<R> a) {
  using Lay = Layout<NX, NB, NU, NY>;
  constexpr int M = NX + NB;
  const long i = blockIdx.x * (long)LQG_BLOCK + threadIdx.x;
  if (i >= a.n_lanes) return;
  const long s = i / a.n_trials, n = i % a.n_trials;
  Spec<R, NX, NB, NU, NY> sp;
  sp.load(a, s);
  const R g = a.g ? a.g[s * a.g_sb + n * a.g_sn] : R(1);
  const R* xp = a.x.p + s * a.x.sb + n * a.x.sn;
  // accumulated bars of the ten time-invariant matrices
  R bdA[NX * NX], bdB[NX * NU], bdF[NY * NX], bdVV[NX * NX], bdWW[NY * NY];
  R baA[NB * NB], baB[NB * NU], baF[NY * NB], baVV[NB * NB], baWW[NY * NY];
  zero<R, NX * NX>(bdA); zero<R, NX * NU>(bdB); zero<R, NY * NX>(bdF); zero<R, NX * NX>(bdVV); zero<R, NY * NY>(bdWW);
  zero<R, NB * NB>(baA); zero<R, NB * NU>(baB); zero<R, NY * NB>(baF); zero<R, NB * NB>(baVV); zero<R, NY * NY>(baWW);
  R mub[M], Sigb[M * M], Pb[NB * NB];
  zero<R, M>(mub); zero<R, M * M>(Sigb); zero<R, NB * NB>(Pb);
  FwdStep<R, NX, NB, NU, NY, ND> f;
  auto store_bars = [&](long slab) {
    R* o = a.out + slab * (long)Lay::TOTAL * a.ld + i;
    store_flat<R, NX * NX>(o + Lay::DA * a.ld, a.ld, bdA);
    store_flat<R, NX * NU>(o + Lay::DB * a.ld, a.ld, bdB);
    store_flat<R, NY * NX>(o + Lay::DF * a.ld, a.ld, bdF);
    store_flat<R, NX * NX>(o + Lay::DVV * a.ld, a.ld, bdVV);
    store_flat<R, NY * NY>(o + Lay::DWW * a.ld, a.ld, bdWW);
    store_flat<R, NB * NB>(o + Lay::AA * a.ld, a.ld, baA);
    store_flat<R, NB * NU>(o + Lay::AB * a.ld, a.ld, baB);
    store_flat<R, NY * NB>(o + Lay::AF * a.ld, a.ld, baF);
    store_flat<R, NB * NB>(o + Lay::AVV * a.ld, a.ld, baVV);
    store_flat<R, NY * NY>(o + Lay::AWW * a.ld, a.ld, baWW);
  };
  for (int t = a.T - 1; t >= 0; --t) {
    if (!TI) {
      sp.load(a, s, t);
      zero<R, NX * NX>(bdA); zero<R, NX * NU>(bdB); zero<R, NY * NX>(bdF); zero<R, NX * NX>(bdVV); zero<R, NY * NY>(bdWW);
      zero<R, NB * NB>(baA); zero<R, NB * NU>(baB); zero<R, NY * NB>(baF); zero<R, NB * NB>(baVV); zero<R, NY * NY>(baWW);
    }
    R* w = a.ws + ((long)t * Lay::STEP) * a.ld + i;
    R P0[NB * NB], Sig[M * M], mu[M], L[NU * NB], xt[ND], x1[ND];
    load_flat<R, NU * NB>(w + Lay::L_OFF * a.ld, a.ld, L);
    load_tri<R, NB>(w + Lay::P_OFF * a.ld, a.ld, P0);
    load_tri<R, M>(w + Lay::SIG_OFF * a.ld, a.ld, Sig);
    load_flat<R, M>(w + Lay::MU_OFF * a.ld, a.ld, mu);
    R xm1[ND];
    LQG_UNROLL for (int k = 0; k < ND; ++k) xt[k] = xp[(long)t * a.x.st + k * a.x.sd];
    LQG_UNROLL for (int k = 0; k < ND; ++k) x1[k] = xp[(long)(t + 1) * a.x.st + k * a.x.sd];
    LQG_UNROLL for (int k = 0; k < ND; ++k) xm1[k] = xp[(long)(t > 0 ? t - 1 : 0) * a.x.st + k * a.x.sd];
    f.system(sp, P0, L);
    f.template moments<false>(Sig, mu, xt, xm1);
    // ---- log-density of x[t+1]                                                     system.py:244-248
    {
      R Soo[ND * ND], Ni[ND * ND], wv[ND];
      get_block<R, M, ND, ND>(f.Sig1, 0, 0, Soo);
      spd_inverse<R, ND>(Soo, Ni);
      LQG_UNROLL for (int p = 0; p < ND; ++p) {
        R acc = R(0);
        LQG_UNROLL for (int k = 0; k < ND; ++k) acc += Ni[p * ND + k] * ((x1[k] - xt[k]) - f.mu1[k]);
        wv[p] = acc;
      }
      LQG_UNROLL for (int p = 0; p < ND; ++p) mub[p] += g * wv[p];
      LQG_UNROLL for (int p = 0; p < ND; ++p)
        LQG_UNROLL for (int q = 0; q < ND; ++q) Sigb[p * M + q] += R(0.5) * g * (wv[p] * wv[q] - Ni[p * ND + q]);
    }
    // ---- Sig1 = F C F' + GG ; mu1 = F c
    constexpr int RR = M - ND;
    R Fb[M * M], GGb[M * M];
    {
      R FbR[M * RR];
      zero<R, M * RR>(FbR);
      mm_acc<R, M, M, RR>(Sigb, f.FCr, FbR, R(2));
      LQG_UNROLL for (int p = 0; p < M; ++p) {
        LQG_UNROLL for (int q = 0; q < ND; ++q) Fb[p * M + q] = mub[p] * f.c[q];
        LQG_UNROLL for (int q = 0; q < RR; ++q) Fb[p * M + ND + q] = FbR[p * RR + q] + mub[p] * f.c[ND + q];
      }
    }
    copy<R, M * M>(Sigb, GGb);
    // ---- adjoint of the conditioning, in terms of Wm = S_ro S_oo^-1 and a = S_oo^-1 r (oracle/lqg_adjoint_np.py)
    {
      R SF[M * RR], Ch[RR * RR], ch[RR], Wtc[ND], ChW[RR * ND], Sro[RR * ND], Soo[ND * ND];
      zero<R, M * RR>(SF);
      mm_acc<R, M, M, RR>(Sigb, f.Fr, SF);
      zero<R, RR * RR>(Ch);
      mtm_acc<R, RR, M, RR>(f.Fr, SF, Ch);
      symmetrise<R, RR>(Ch);
      LQG_UNROLL for (int p = 0; p < RR; ++p) {
        R acc = R(0);
        LQG_UNROLL for (int k = 0; k < M; ++k) acc += f.Fr[k * RR + p] * mub[k];
        ch[p] = acc;
      }
      LQG_UNROLL for (int p = 0; p < ND; ++p) {
        R acc = R(0);
        LQG_UNROLL for (int k = 0; k < RR; ++k) acc += f.Wm[k * ND + p] * ch[k];
        Wtc[p] = acc;
      }
      zero<R, RR * ND>(ChW);
      mm_acc<R, RR, RR, ND>(Ch, f.Wm, ChW);
      LQG_UNROLL for (int p = 0; p < RR; ++p)
        LQG_UNROLL for (int q = 0; q < ND; ++q) Sro[p * ND + q] = ch[p] * f.av[q] - R(2) * ChW[p * ND + q];
      zero<R, ND * ND>(Soo);
      mtm_acc<R, ND, RR, ND>(f.Wm, ChW, Soo);
      LQG_UNROLL for (int p = 0; p < ND; ++p)
        LQG_UNROLL for (int q = 0; q < ND; ++q) Soo[p * ND + q] -= Wtc[p] * f.av[q];
      symmetrise<R, ND>(Soo);
      LQG_UNROLL for (int p = 0; p < ND; ++p) mub[p] = -Wtc[p];
      LQG_UNROLL for (int p = 0; p < RR; ++p) mub[ND + p] = ch[p];
      LQG_UNROLL for (int p = 0; p < ND; ++p)
        LQG_UNROLL for (int q = 0; q < ND; ++q) Sigb[p * M + q] = Soo[p * ND + q];
      LQG_UNROLL for (int p = 0; p < RR; ++p)
        LQG_UNROLL for (int q = 0; q < ND; ++q) {
          Sigb[(ND + p) * M + q] = R(0.5) * Sro[p * ND + q];
          Sigb[q * M + ND + p] = R(0.5) * Sro[p * ND + q];
        }
      LQG_UNROLL for (int p = 0; p < RR; ++p)
        LQG_UNROLL for (int q = 0; q < RR; ++q) Sigb[(ND + p) * M + ND + q] = Ch[p * RR + q];
    }
    if (t == 0) {
      LQG_UNROLL for (int k = 0; k < M * M; ++k) GGb[k] += Sigb[k];      // Sigma_0 = G_0 G_0'
    }
    // ---- joint system -> spec bars, Lbar, Kbar                                     system.py:167-202
    R F11[NX * NX], F12[NX * NB], F21[NB * NX], F22[NB * NB], G11[NX * NX], G21[NB * NX], G22[NB * NB];
    get_block<R, M, NX, NX>(Fb, 0, 0, F11);
    get_block<R, M, NX, NB>(Fb, 0, NX, F12);
    get_block<R, M, NB, NX>(Fb, NX, 0, F21);
    get_block<R, M, NB, NB>(Fb, NX, NX, F22);
    get_block<R, M, NX, NX>(GGb, 0, 0, G11);
    get_block<R, M, NB, NX>(GGb, NX, 0, G21);
    get_block<R, M, NB, NB>(GGb, NX, NX, G22);
    R Yb[NB * NX], KtF22[NY * NB], Kb[NB * NY], Db[NY * NU], Lbar[NU * NB];
    zero<R, NB * NX>(Yb);
    mmt_acc<R, NB, NX, NX>(F21, sp.Ad, Yb);
    mm_acc<R, NB, NX, NX>(G21, sp.VVd, Yb, R(2));
    mm_acc<R, NB, NB, NX>(G22, f.YVV, Yb, R(2));
    zero<R, NY * NB>(KtF22);
    mtm_acc<R, NY, NB, NB>(f.K, F22, KtF22);
    zero<R, NB * NY>(Kb);
    mmt_acc<R, NB, NX, NY>(Yb, sp.Fd, Kb);
    mmt_acc<R, NB, NB, NY>(F22, f.Z, Kb);
    mm_acc<R, NB, NB, NY>(G22, f.KWW, Kb, R(2));
    zero<R, NY * NU>(Db);
    mmt_acc<R, NY, NB, NU>(KtF22, L, Db);
    // dynamics
    LQG_UNROLL for (int k = 0; k < NX * NX; ++k) bdA[k] += F11[k];
    mtm_acc<R, NX, NB, NX>(f.Y, F21, bdA);
    mmt_acc<R, NX, NB, NU>(F12, L, bdB);
    mtm_acc<R, NX, NY, NU>(sp.Fd, Db, bdB);
    mtm_acc<R, NY, NB, NX>(f.K, Yb, bdF);
    mmt_acc<R, NY, NU, NX>(Db, sp.Bd, bdF);
    LQG_UNROLL for (int k = 0; k < NX * NX; ++k) bdVV[k] += G11[k];
    mtm_acc<R, NX, NB, NX>(f.Y, G21, bdVV, R(2));
    {
      R G22Y[NB * NX], G22K[NB * NY];
      zero<R, NB * NX>(G22Y);
      mm_acc<R, NB, NB, NX>(G22, f.Y, G22Y);
      mtm_acc<R, NX, NB, NX>(f.Y, G22Y, bdVV);
      zero<R, NB * NY>(G22K);
      mm_acc<R, NB, NB, NY>(G22, f.K, G22K);
      mtm_acc<R, NY, NB, NY>(f.K, G22K, bdWW);
    }
    // actor (joint-system part)
    LQG_UNROLL for (int k = 0; k < NB * NB; ++k) baA[k] += F22[k];
    mtm_acc<R, NB, NY, NB>(sp.Fa, KtF22, baA, R(-1));
    mmt_acc<R, NB, NB, NU>(F22, L, baB);
    mtm_acc<R, NB, NY, NU>(sp.Fa, Db, baB, R(-1));
    mmt_acc<R, NY, NB, NB>(KtF22, sp.Aa, baF, R(-1));
    mmt_acc<R, NY, NU, NB>(Db, sp.Ba, baF, R(-1));
    zero<R, NU * NB>(Lbar);
    mtm_acc<R, NU, NX, NB>(sp.Bd, F12, Lbar);
    mtm_acc<R, NU, NB, NB>(sp.Ba, F22, Lbar);
    mtm_acc<R, NU, NY, NB>(f.D, KtF22, Lbar);
    store_flat<R, NU * NB>(w + Lay::L_OFF * a.ld, a.ld, Lbar);         // over L_t's slot, for k_adj_riccati_rev
    // ---- Kalman step adjoint                                                       kf.py:10-14
    {
      mmt_acc<R, NB, NB, NY>(Pb, f.FPp, Kb, R(-1));                    // Kb -= Pb (F Pp)'
      R KF[NB * NB], Ppb[NB * NB], KbGi[NB * NY], Wm[NB * NY], T1[NY * NY], Gmb[NY * NY];
      zero<R, NB * NB>(KF);
      mm_acc<R, NB, NY, NB>(f.K, sp.Fa, KF);
      copy<R, NB * NB>(Pb, Ppb);
      mtm_acc<R, NB, NB, NB>(KF, Pb, Ppb, R(-1));
      zero<R, NB * NY>(KbGi);
      mm_acc<R, NB, NY, NY>(Kb, f.Gi, KbGi);
      mm_acc<R, NB, NY, NB>(KbGi, sp.Fa, Ppb);
      copy<R, NB * NY>(KbGi, Wm);
      mm_acc<R, NB, NB, NY>(Pb, f.K, Wm, R(-1));
      mtm_acc<R, NY, NB, NB>(Wm, f.Pp, baF);                           // aF += (Kb Gi - Pb K)' Pp
      zero<R, NY * NY>(T1);
      mm_acc<R, NY, NB, NY>(f.FPp, KbGi, T1);
      zero<R, NY * NY>(Gmb);
      mm_acc<R, NY, NY, NY>(f.Gi, T1, Gmb, R(-1));
      R FtG[NB * NY];
      zero<R, NB * NY>(FtG);
      mtm_acc<R, NB, NY, NY>(sp.Fa, Gmb, FtG);
      mm_acc<R, NB, NY, NB>(FtG, sp.Fa, Ppb);
      symmetrise<R, NB>(Ppb);
      R Gs[NY * NY];
      LQG_UNROLL for (int p = 0; p < NY; ++p)
        LQG_UNROLL for (int q = 0; q < NY; ++q) Gs[p * NY + q] = Gmb[p * NY + q] + Gmb[q * NY + p];
      mm_acc<R, NY, NY, NB>(Gs, f.FPp, baF);
      LQG_UNROLL for (int k = 0; k < NY * NY; ++k) baWW[k] += Gmb[k];
      LQG_UNROLL for (int k = 0; k < NB * NB; ++k) baVV[k] += Ppb[k];
      R PA[NB * NB];
      zero<R, NB * NB>(PA);
      mm_acc<R, NB, NB, NB>(Ppb, sp.Aa, PA);
      mm_acc<R, NB, NB, NB>(PA, P0, baA, R(2));
      zero<R, NB * NB>(Pb);
      mtm_acc<R, NB, NB, NB>(sp.Aa, PA, Pb);
      symmetrise<R, NB>(Pb);
    }
    if (t == 0 && !a.Sigma0.p) {                                       // default Sigma0 = V_0 V_0'  system.py:160
      LQG_UNROLL for (int k = 0; k < NB * NB; ++k) baVV[k] += Pb[k];
    }
    if (!TI) store_bars(t);
  }
  if (TI) store_bars(0);
  store_flat<R, NB * NB>(a.out + i + Lay::AS0 * a.ld, a.ld, Pb);
}

template <typename R, int NX, int NB, int NU, int NY, int ND, bool TI>
__global__ void __launch_bounds__(LQG_BLOCK, 1) k_adj_riccati_rev(const AdjArgs<R> a) {
  using Lay = Layout<NX, NB, NU, NY>;
  const long i = blockIdx.x * (long)LQG_BLOCK + threadIdx.x;
  if (i >= a.n_lanes) return;
  const long s = i / a.n_trials;
  R A[NB * NB], B[NB * NU], Rm[NU * NU], P[NU * NB];
  auto load_step = [&](long t) {
    load_mat<R, NB, NB>(a.A.p + s * a.A.sb + t * a.A.st, a.A.sr, a.A.sc, A);
    load_mat<R, NB, NU>(a.B.p + s * a.B.sb + t * a.B.st, a.B.sr, a.B.sc, B);
    load_sym<R, NU>(a.Rm.p + s * a.Rm.sb + t * a.Rm.st, a.Rm.sr, a.Rm.sc, Rm);
    zero<R, NU * NB>(P);
    if (a.P.p) load_mat<R, NU, NB>(a.P.p + s * a.P.sb + t * a.P.st, a.P.sr, a.P.sc, P);
  };
  if (TI) load_step(0);
  R bA[NB * NB], bB[NB * NU], bQ[NB * NB], bR[NU * NU], Sb[NB * NB];
  zero<R, NB * NB>(bA); zero<R, NB * NU>(bB); zero<R, NB * NB>(bQ); zero<R, NU * NU>(bR); zero<R, NB * NB>(Sb);
  auto store_bars = [&](long slab) {
    R* o = a.out + slab * (long)Lay::TOTAL * a.ld + i;
    store_flat<R, NB * NB>(o + Lay::AA2 * a.ld, a.ld, bA);
    store_flat<R, NB * NU>(o + Lay::AB2 * a.ld, a.ld, bB);
    store_flat<R, NB * NB>(o + Lay::AQ * a.ld, a.ld, bQ);
    store_flat<R, NU * NU>(o + Lay::AR * a.ld, a.ld, bR);
  };
  RicStep<R, NB, NU> st;
  for (int t = 0; t < a.T; ++t) {
    if (!TI) {
      load_step(t);
      zero<R, NB * NB>(bA); zero<R, NB * NU>(bB); zero<R, NB * NB>(bQ); zero<R, NU * NU>(bR);
    }
    const R* w = a.ws + ((long)t * Lay::STEP) * a.ld + i;
    R S[NB * NB], Lb[NU * NB];
    load_tri<R, NB>(w + Lay::S_OFF * a.ld, a.ld, S);
    load_flat<R, NU * NB>(w + Lay::L_OFF * a.ld, a.ld, Lb);
    st.compute(S, A, B, Rm, P, a.eps);
    LQG_UNROLL for (int k = 0; k < NB * NB; ++k) bQ[k] += Sb[k];
    R HLG[NU * NB], Gb[NU * NB], Hb[NU * NU], HtiLb[NU * NB], LSb[NU * NB];
    copy<R, NU * NB>(st.G, HLG);
    mm_acc<R, NU, NU, NB>(st.H, st.L, HLG);
    mm_acc<R, NU, NB, NB>(HLG, Sb, Lb, R(2));                          // Lb += 2 (H L + G) Sb
    zero<R, NU * NB>(HtiLb);
    mm_acc<R, NU, NU, NB>(st.Hti, Lb, HtiLb);
    zero<R, NU * NB>(LSb);
    mm_acc<R, NU, NB, NB>(st.L, Sb, LSb);
    LQG_UNROLL for (int k = 0; k < NU * NB; ++k) Gb[k] = R(2) * LSb[k] - HtiLb[k];
    zero<R, NU * NU>(Hb);
    mmt_acc<R, NU, NB, NU>(LSb, st.L, Hb);
    mmt_acc<R, NU, NB, NU>(HtiLb, st.L, Hb, R(-1));
    LQG_UNROLL for (int k = 0; k < NU * NU; ++k) bR[k] += Hb[k];
    mm_acc<R, NB, NB, NB>(st.SA, Sb, bA, R(2));
    mm_acc<R, NB, NU, NB>(st.SB, Gb, bA);
    mmt_acc<R, NB, NB, NU>(st.SA, Gb, bB);
    R Hs[NU * NU];
    LQG_UNROLL for (int p = 0; p < NU; ++p)
      LQG_UNROLL for (int q = 0; q < NU; ++q) Hs[p * NU + q] = Hb[p * NU + q] + Hb[q * NU + p];
    mm_acc<R, NB, NU, NU>(st.SB, Hs, bB);
    // Sb <- sym(A Sb A' + B Gb A' + B Hb B')
    R X1[NB * NB], X2[NB * NU];
    zero<R, NB * NB>(X1);
    mm_acc<R, NB, NB, NB>(A, Sb, X1);
    mm_acc<R, NB, NU, NB>(B, Gb, X1);                                  // (A Sb + B Gb)
    zero<R, NB * NU>(X2);
    mm_acc<R, NB, NU, NU>(B, Hb, X2);
    zero<R, NB * NB>(Sb);
    mmt_acc<R, NB, NB, NB>(X1, A, Sb);
    mmt_acc<R, NB, NU, NB>(X2, B, Sb);
    symmetrise<R, NB>(Sb);
    if (!TI) store_bars(t);
  }
  if (TI) store_bars(0);
  store_flat<R, NB * NB>(a.out + i + Lay::AQF * a.ld, a.ld, Sb);
}

}  // namespace adj
}  // namespace lqg
