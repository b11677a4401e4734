// lqg_coop.hpp — COOPERATIVE system sweeps: one workgroup per system, matrices staged in LDS, run-time dimensions.
//
// The lane-per-system kernels (lqg_kernels.hpp, lqg_kernels_sp.hpp) put one whole system into one lane's registers:
// unbeatable when there are >= 10^4 systems to fill the chip, but (i) a handful of systems (one parameter vector with
// many trials: every NUTS / Adam step of lqg/infer/utils.py:18,37-39, lqg/infer/mle.py:17-23; BASELINE configs 2 and 4)
// leaves one lane walking the whole recursion at one instruction per 4 cycles, and (ii) a joint dimension beyond ~20
// (the reference's DelayedSubjectiveActor, lqg/tracking/delay.py:44-51: x=26, b=39) does not fit a lane at all.
// Here the lanes of a workgroup share ONE system: every matrix product of a step is spread over the lanes (one output
// element per lane, operands read from LDS), independent products share a stage, and a stage ends with one workgroup
// barrier (a single-wave workgroup needs none: LDS operations of a wave complete in order).  The latency of a step is
// then (number of dependent stages) x (one LDS round trip + K FMAs) instead of (number of scalar FMAs) x 4 cycles:
// Riccati 4 stages per step, forward sweep 5 (the Kalman recursion of step t+1 is software-pipelined under the joint
// system / Sigma recursion of step t).  Dimensions are run-time arguments: ONE compiled kernel serves every model shape
// (no per-shape instantiation, no on-demand compile), time-varying specs and affine cost terms included.  When the
// working set exceeds LDS (m = 65 in fp64) the same code runs with its arena in global memory (L2-resident).
//
// Mathematics and operation order follow k_riccati / k_forward (Schur-form moment recursion, deviation-form operators,
// Cholesky-based inverses), so results agree with the lane kernels to rounding.  Reference parity:
// lqg/control/lqr.py:16-42, lqg/belief/kf.py:6-21, lqg/system.py:142-248.
#pragma once
#include "lqg_kernels.hpp"

namespace lqg {
namespace coop {

constexpr int kMaxSmall = 6;    // u, y, d <= 6: their u x u / y x y / d x d factorizations run in registers

template <typename R>
struct Args {
  DView<R> aQ, aq, aQf, aqf, aP, aR, ar, aA, aB, aF, aV, aW;   // actor spec (q, qf, P, r may be null = zero)
  DView<R> dA, dB, dF, dV, dW;                                  // dynamics spec
  DView<R> Sigma0;                                              // Kalman initial covariance, may be null
  DView<R> L, l, H, K, Sig;                                     // optional outputs (lqr.py:42, kf.py:21, system.py:235)
  R* Ls;            // gain scratch [n_sys][T][u*b]
  R* ops;           // trial-operator stream [n_sys][T+1][nops] (layout of TrialOps), may be null
  R* arena;         // GLOBAL variant: [n_sys][arena_reals] working set
  long arena_reals;
  long n_sys;
  int T, x, b, u, y, d, nva, nwa, nvd, nwd, nops;
  int ti;           // 1: every spec field time-invariant
  R eps;
};

// (row, col) of this thread's first element of an r x c result; later elements (r*c > BLOCK) by division
template <int BLOCK>
struct Shape {
  int rows, cols, n, i0, j0;
  LQG_DEV Shape(int r, int c) : rows(r), cols(c), n(r * c) {
    i0 = (int)threadIdx.x / (c > 0 ? c : 1);
    j0 = (int)threadIdx.x - i0 * c;
  }
  template <typename F>
  LQG_DEV void each(F f) const {
    if ((int)threadIdx.x < n) f(i0, j0);
    for (int e = (int)threadIdx.x + BLOCK; e < n; e += BLOCK) {
      const int i = e / cols;
      f(i, e - i * cols);
    }
  }
};

template <int BLOCK>
LQG_DEV void stage_end() {
  __syncthreads();   // workgroup barrier + LDS/global visibility inside the workgroup
}

// ---- strided global -> arena loads (cooperative) ---------------------------------------------------------------------
template <int BLOCK, typename R>
LQG_DEV void ld_mat(const Shape<BLOCK>& sh, const DView<R>& v, long s, int t, R* dst) {
  const R* p = v.p + s * v.sb + (long)t * v.st;
  sh.each([&](int i, int j) { dst[i * sh.cols + j] = p[i * v.sr + j * v.sc]; });
}
template <int BLOCK, typename R>
LQG_DEV void ld_sym(const Shape<BLOCK>& sh, const DView<R>& v, long s, int t, R* dst) {   // symmetric part
  const R* p = v.p + s * v.sb + (long)t * v.st;
  sh.each([&](int i, int j) {
    dst[i * sh.cols + j] = (i == j) ? p[i * v.sr + i * v.sc] : R(0.5) * (p[i * v.sr + j * v.sc] + p[j * v.sr + i * v.sc]);
  });
}
template <int BLOCK, typename R>
LQG_DEV void ld_gram(const Shape<BLOCK>& sh, const DView<R>& v, long s, int t, int nv, R* dst) {   // V V^T
  const R* p = v.p + s * v.sb + (long)t * v.st;
  sh.each([&](int i, int j) {
    const int lo = i < j ? i : j, hi = i < j ? j : i;       // (min, max): both mirror entries get the same bits
    R acc = R(0);
    for (int k = 0; k < nv; ++k) acc += p[lo * v.sr + k * v.sc] * p[hi * v.sr + k * v.sc];
    dst[i * sh.cols + j] = acc;
  });
}
template <int BLOCK, typename R>
LQG_DEV void ld_vec(const DView<R>& v, long s, int t, int n, R* dst) {
  for (int i = threadIdx.x; i < n; i += BLOCK) dst[i] = v.p ? v.p[s * v.sb + (long)t * v.st + i * v.sr] : R(0);
}

// ---- small symmetric positive-definite factorizations in registers (every thread redundantly, N = run-time n) ---------
// Hi = (H + max(0, eps - lambda_min(H)) I)^-1, Ht returned too (lqr.py:27-31)
template <typename R, int N>
LQG_DEV void small_reg_inverse(const R* Hs, R eps, bool floor_, R (&Hi)[kMaxSmall * kMaxSmall], R (&Ht_)[kMaxSmall * kMaxSmall]) {
  R H[N * N], Ht[N * N], Lc[N * N], dinv[N], Li[N * N], Hv[N * N];
  LQG_UNROLL for (int i = 0; i < N * N; ++i) H[i] = Hs[i];
  LQG_UNROLL for (int i = 0; i < N * N; ++i) Ht[i] = H[i];
  if (floor_) {
    R shift = eps - min_eig_sym<R, N>(H);
    shift = (shift > R(0)) ? shift : R(0);
    LQG_UNROLL for (int i = 0; i < N; ++i) Ht[i * N + i] += shift;
  }
  chol_lower<R, N>(Ht, Lc, dinv);
  tri_inverse_lower<R, N>(Lc, dinv, Li);
  spd_inverse_from_tri<R, N>(Li, Hv);
  LQG_UNROLL for (int i = 0; i < N; ++i)
    LQG_UNROLL for (int j = 0; j < N; ++j) { Hi[i * kMaxSmall + j] = Hv[i * N + j]; Ht_[i * kMaxSmall + j] = Ht[i * N + j]; }
}
template <typename R>
LQG_DEV void reg_inverse(int n, const R* Hs, R eps, bool floor_, R (&Hi)[kMaxSmall * kMaxSmall], R (&Ht)[kMaxSmall * kMaxSmall]) {
  switch (n) {
    case 1: small_reg_inverse<R, 1>(Hs, eps, floor_, Hi, Ht); break;
    case 2: small_reg_inverse<R, 2>(Hs, eps, floor_, Hi, Ht); break;
    case 3: small_reg_inverse<R, 3>(Hs, eps, floor_, Hi, Ht); break;
    case 4: small_reg_inverse<R, 4>(Hs, eps, floor_, Hi, Ht); break;
    case 5: small_reg_inverse<R, 5>(Hs, eps, floor_, Hi, Ht); break;
    default: small_reg_inverse<R, 6>(Hs, eps, floor_, Hi, Ht); break;
  }
}
// Li = chol(S_oo)^-1 (lower), half log-determinant; S_oo = leading n x n block of a matrix with leading dimension ld
template <typename R, int N>
LQG_DEV void small_reg_chol(const R* Sg, int ld, R (&Li_)[kMaxSmall * kMaxSmall], R& hl) {
  R A[N * N], Lc[N * N], dinv[N], Li[N * N];
  LQG_UNROLL for (int i = 0; i < N; ++i)
    LQG_UNROLL for (int j = 0; j < N; ++j) A[i * N + j] = Sg[i * ld + j];
  chol_lower<R, N>(A, Lc, dinv);
  tri_inverse_lower<R, N>(Lc, dinv, Li);
  R pd = dinv[0];
  LQG_UNROLL for (int i = 1; i < N; ++i) pd *= dinv[i];
  hl = -log_<R>(pd);
  LQG_UNROLL for (int i = 0; i < N; ++i)
    LQG_UNROLL for (int j = 0; j < N; ++j) Li_[i * kMaxSmall + j] = Li[i * N + j];
}
template <typename R>
LQG_DEV void reg_chol(int n, const R* Sg, int ld, R (&Li)[kMaxSmall * kMaxSmall], R& hl) {
  switch (n) {
    case 1: small_reg_chol<R, 1>(Sg, ld, Li, hl); break;
    case 2: small_reg_chol<R, 2>(Sg, ld, Li, hl); break;
    case 3: small_reg_chol<R, 3>(Sg, ld, Li, hl); break;
    case 4: small_reg_chol<R, 4>(Sg, ld, Li, hl); break;
    case 5: small_reg_chol<R, 5>(Sg, ld, Li, hl); break;
    default: small_reg_chol<R, 6>(Sg, ld, Li, hl); break;
  }
}
// reals of the arena each kernel carves (host and device agree through these)
inline __host__ __device__ long riccati_arena_reals(int b, int u) {
  return 4L * b * b + 6L * b * u + 2L * u * u + 4L * b + 5L * u + 16;
}
inline __host__ __device__ long kalman_arena_reals(int b, int y) { return 5L * b * b + 3L * y * b + 2L * y * y + 16; }
inline __host__ __device__ long forward_arena_reals(int x, int b, int u, int y, int d) {
  const long m = x + b, o = d, rr = m - d;
  return /*Aa VVa P AP Pp KFAa*/ 6L * b * b + /*Ba BK*/ 2L * b * u + /*Fa FAa FP*/ 3L * y * b + /*WWa Gk N3 WWd*/ 4L * y * y +
         /*Ad N1*/ 2L * x * x + /*Bd*/ 1L * x * u + /*FAd N2 Fd*/ 3L * y * x + /*DB*/ 1L * y * u + /*K KN3*/ 2L * b * y +
         /*L*/ 1L * u * b + /*KFAd BdL KN2*/ 3L * b * x + /*Fj GG Sg*/ 3L * m * m + /*U2*/ rr * o + /*C*/ rr * rr + /*T1*/ m * rr + 16;
}

// ======================================================================================================================
// Riccati backward (lqr.py:16-42): carries S[b,b], s[b]; emits L_t (scratch + optional L, l, H outputs)
template <typename R, int BLOCK, bool GLOBAL>
__global__ void __launch_bounds__(BLOCK) k_coop_riccati(const Args<R> a) {
  extern __shared__ double lqg_coop_smem[];
  const long s = blockIdx.x;
  R* ar = GLOBAL ? a.arena + s * a.arena_reals : reinterpret_cast<R*>(lqg_coop_smem);
  const int b = a.b, u = a.u;
  const bool affine = a.aq.p || a.aqf.p || a.aP.p || a.ar.p;
  auto take = [&](long n) { R* p = ar; ar += n; return p; };
  R *S = take(b * b), *A = take(b * b), *Q = take(b * b), *SA = take(b * b);
  R *Bm = take(b * u), *SB = take(b * u), *P = take(b * u), *G = take(b * u), *Lm = take(b * u), *W1 = take(b * u);
  R *Rm = take(u * u), *H = take(u * u);
  R *sv = take(b), *sn = take(b), *q = take(b), *gv = take(u), *lv = take(u), *Hl = take(u), *r = take(u);
  const Shape<BLOCK> bb(b, b), bu(b, u), ub(u, b), uu(u, u);

  auto load_step = [&](int t) {
    ld_mat(bb, a.aA, s, t, A);
    ld_mat(bu, a.aB, s, t, Bm);
    ld_sym(bb, a.aQ, s, t, Q);
    ld_sym(uu, a.aR, s, t, Rm);
    if (affine) {
      if (a.aP.p) ld_mat(ub, a.aP, s, t, P);
      else ub.each([&](int i, int j) { P[i * b + j] = R(0); });
      ld_vec<BLOCK>(a.aq, s, t, b, q);
      ld_vec<BLOCK>(a.ar, s, t, u, r);
    }
  };
  ld_sym(bb, a.aQf, s, 0, S);                               // carry init (Qf, qf)  lqr.py:38
  if (affine) ld_vec<BLOCK>(a.aqf, s, 0, b, sv);
  if (a.ti) load_step(0);
  stage_end<BLOCK>();

  for (int t = a.T - 1; t >= 0; --t) {                      // reverse=True  lqr.py:40
    if (!a.ti) { load_step(t); stage_end<BLOCK>(); }
    // ---- R1: SA = S A, SB = S B
    bb.each([&](int i, int j) {
      R acc = R(0);
      for (int k = 0; k < b; ++k) acc += S[i * b + k] * A[k * b + j];
      SA[i * b + j] = acc;
    });
    bu.each([&](int i, int j) {
      R acc = R(0);
      for (int k = 0; k < b; ++k) acc += S[i * b + k] * Bm[k * u + j];
      SB[i * u + j] = acc;
    });
    stage_end<BLOCK>();
    // ---- R2: H = R + B'SB (symmetric), G = P + B'SA, g = r + B's                     lqr.py:22-24
    uu.each([&](int i, int j) {
      const int lo = i < j ? i : j, hi = i < j ? j : i;
      R acc = Rm[lo * u + hi];
      for (int k = 0; k < b; ++k) acc += Bm[k * u + lo] * SB[k * u + hi];
      H[i * u + j] = acc;
    });
    ub.each([&](int i, int j) {
      R acc = affine ? P[i * b + j] : R(0);
      for (int k = 0; k < b; ++k) acc += Bm[k * u + i] * SA[k * b + j];
      G[i * b + j] = acc;
    });
    if (affine)
      for (int i = threadIdx.x; i < u; i += BLOCK) {
        R acc = r[i];
        for (int k = 0; k < b; ++k) acc += Bm[k * u + i] * sv[k];
        gv[i] = acc;
      }
    stage_end<BLOCK>();
    // ---- R3: Ht^-1 in registers; L = -Ht^-1 G, W1 = H L + G (each thread recomputes its column of L)   lqr.py:27-33
    {
      const bool busy = (int)threadIdx.x < (u * b > u * u ? u * b : u * u) || BLOCK < u * b;
      if (busy) {
        R Hi[kMaxSmall * kMaxSmall], Ht[kMaxSmall * kMaxSmall];
        reg_inverse<R>(u, H, a.eps, true, Hi, Ht);
        ub.each([&](int i, int j) {
          R Lcol[kMaxSmall];
          LQG_UNROLL for (int k = 0; k < kMaxSmall; ++k) {
            R acc = R(0);
            if (k < u) {
              LQG_UNROLL for (int l2 = 0; l2 < kMaxSmall; ++l2)
                if (l2 < u) acc -= Hi[k * kMaxSmall + l2] * G[l2 * b + j];
            }
            Lcol[k] = acc;
          }
          R lij = Lcol[0], w = G[i * b + j];
          LQG_UNROLL for (int k = 1; k < kMaxSmall; ++k) lij = (i == k) ? Lcol[k] : lij;
          LQG_UNROLL for (int k = 0; k < kMaxSmall; ++k)
            if (k < u) w += H[i * u + k] * Lcol[k];
          Lm[i * b + j] = lij;
          W1[i * b + j] = w;
          if (a.Ls) a.Ls[(s * a.T + t) * (long)(u * b) + i * b + j] = lij;
          if (a.L.p) const_cast<R*>(a.L.p)[s * a.L.sb + (long)t * a.L.st + i * a.L.sr + j * a.L.sc] = lij;
        });
        if (a.H.p)
          uu.each([&](int i, int j) {
            R v = Ht[0];
            LQG_UNROLL for (int k = 0; k < kMaxSmall; ++k)
              LQG_UNROLL for (int c = 0; c < kMaxSmall; ++c) v = (i == k && j == c) ? Ht[k * kMaxSmall + c] : v;
            const_cast<R*>(a.H.p)[s * a.H.sb + (long)t * a.H.st + i * a.H.sr + j * a.H.sc] = v;   // regularised Ht  lqr.py:36
          });
        if (affine && threadIdx.x == 0) {                     // l = -Ht^-1 g, Hl = H l + g (u <= 6: one thread)
          R lvr[kMaxSmall];
          LQG_UNROLL for (int i = 0; i < kMaxSmall; ++i) {
            R acc = R(0);
            if (i < u) {
              LQG_UNROLL for (int k = 0; k < kMaxSmall; ++k)
                if (k < u) acc -= Hi[i * kMaxSmall + k] * gv[k];
            }
            lvr[i] = acc;
          }
          for (int i = 0; i < u; ++i) {
            R acc = gv[i];
            LQG_UNROLL for (int k = 0; k < kMaxSmall; ++k)
              if (k < u) acc += H[i * u + k] * lvr[k];
            Hl[i] = acc;
          }
          LQG_UNROLL for (int i = 0; i < kMaxSmall; ++i)
            if (i < u) lv[i] = lvr[i];
        }
      }
      if (a.l.p && !affine)
        for (int i = threadIdx.x; i < u; i += BLOCK) const_cast<R*>(a.l.p)[s * a.l.sb + (long)t * a.l.st + i * a.l.sr] = R(0);
    }
    stage_end<BLOCK>();
    // ---- R4: S = Q + A'SA + L'(HL + G) + G'L (symmetric); s = q + A's + G'l + L'(Hl + g)            lqr.py:33-34
    bb.each([&](int i, int j) {
      const int lo = i < j ? i : j, hi = i < j ? j : i;
      R acc = Q[lo * b + hi];
      for (int k = 0; k < b; ++k) acc += A[k * b + lo] * SA[k * b + hi];
      for (int k = 0; k < u; ++k) acc += Lm[k * b + lo] * W1[k * b + hi] + G[k * b + lo] * Lm[k * b + hi];
      S[i * b + j] = acc;
    });
    if (affine) {
      for (int i = threadIdx.x; i < b; i += BLOCK) {
        R acc = q[i];
        for (int k = 0; k < b; ++k) acc += A[k * b + i] * sv[k];
        for (int k = 0; k < u; ++k) acc += G[k * b + i] * lv[k] + Lm[k * b + i] * Hl[k];
        sn[i] = acc;
      }
      if (a.l.p)
        for (int i = threadIdx.x; i < u; i += BLOCK) const_cast<R*>(a.l.p)[s * a.l.sb + (long)t * a.l.st + i * a.l.sr] = lv[i];
    }
    stage_end<BLOCK>();
    if (affine) {
      for (int i = threadIdx.x; i < b; i += BLOCK) sv[i] = sn[i];
      // (sv is next read in R2 of the following step, after two more barriers)
    }
  }
}

// ======================================================================================================================
// Forward sweep: Kalman recursion (kf.py:6-21), joint system (system.py:167-207), Schur-form moment recursion
// (system.py:209-235), trial operators per step.  Kalman step t+1 is pipelined under the joint / Sigma stages of step t.
template <typename R, int BLOCK, bool GLOBAL>
__global__ void __launch_bounds__(BLOCK) k_coop_forward(const Args<R> a) {
  extern __shared__ double lqg_coop_smem[];
  const long s = blockIdx.x;
  R* ar = GLOBAL ? a.arena + s * a.arena_reals : reinterpret_cast<R*>(lqg_coop_smem);
  const int x = a.x, b = a.b, u = a.u, y = a.y, o = a.d, m = a.x + a.b, rr = m - a.d;
  auto take = [&](long n) { R* p = ar; ar += n; return p; };
  // (the arrays of the Kalman recursion come first: a gains-only call, kf.forward, needs kalman_arena_reals only)
  R *Aa = take(b * b), *VVa = take(b * b), *P = take(b * b), *AP = take(b * b), *Pp = take(b * b);
  R *Fa = take(y * b), *FP = take(y * b), *WWa = take(y * y), *Gk = take(y * y), *K = take(b * y);
  R *KFAa = take(b * b), *Ba = take(b * u), *BK = take(b * u), *FAa = take(y * b);
  R *N3 = take(y * y), *WWd = take(y * y);
  R *Ad = take(x * x), *N1 = take(x * x), *Bd = take(x * u);
  R *FAd = take(y * x), *N2 = take(y * x), *Fd = take(y * x), *DB = take(y * u);
  R *KN3 = take(b * y), *Lm = take(u * b);
  R *KFAd = take(b * x), *BdL = take(x * b), *KN2 = take(b * x);
  R *Fj = take(m * m), *GG = take(m * m), *Sg = take(m * m), *U2 = take(rr * o), *C = take(rr * rr), *T1 = take(m * rr);
  const Shape<BLOCK> bb(b, b), bu(b, u), yb(y, b), yy(y, y), xx(x, x), xu(x, u), yx(y, x), yu(y, u), by(b, y), bx(b, x),
      xb(x, b), mm_(m, m), ro(rr, o), rrs(rr, rr), mr(m, rr);
  const bool joint = a.ops || a.Sig.p;                      // false: only the Kalman gains are wanted (kf.forward)
  const R kLogNorm = R(0.5 * 1.8378770664093453) * (R)o;

  auto load_consts = [&](int t) {                            // stage L1: direct loads and Gram matrices
    ld_mat(bb, a.aA, s, t, Aa);
    ld_mat(yb, a.aF, s, t, Fa);
    ld_gram(bb, a.aV, s, t, a.nva, VVa);
    ld_gram(yy, a.aW, s, t, a.nwa, WWa);
    if (joint) {
      ld_mat(bu, a.aB, s, t, Ba);
      ld_mat(xx, a.dA, s, t, Ad);
      ld_mat(xu, a.dB, s, t, Bd);
      ld_mat(yx, a.dF, s, t, Fd);
      ld_gram(xx, a.dV, s, t, a.nvd, N1);
      ld_gram(yy, a.dW, s, t, a.nwd, WWd);
    }
  };
  auto hoist1 = [&]() {                                      // stage L2: Fa Aa, Fd Ad, Fd Bd - Fa Ba, Fd Vd Vd'
    if (!joint) return;
    yb.each([&](int i, int j) {
      R acc = R(0);
      for (int k = 0; k < b; ++k) acc += Fa[i * b + k] * Aa[k * b + j];
      FAa[i * b + j] = acc;
    });
    yx.each([&](int i, int j) {
      R acc = R(0), acc2 = R(0);
      for (int k = 0; k < x; ++k) { acc += Fd[i * x + k] * Ad[k * x + j]; acc2 += Fd[i * x + k] * N1[k * x + j]; }
      FAd[i * x + j] = acc;
      N2[i * x + j] = acc2;
    });
    yu.each([&](int i, int j) {
      R f1 = R(0), f2 = R(0);
      for (int k = 0; k < x; ++k) f1 += Fd[i * x + k] * Bd[k * u + j];
      for (int k = 0; k < b; ++k) f2 += Fa[i * b + k] * Ba[k * u + j];
      DB[i * u + j] = f1 - f2;                               // system.py:177-180
    });
  };
  auto hoist2 = [&]() {                                      // stage L3: N3 = Fd Vd Vd' Fd' + Wd Wd'
    if (!joint) return;
    yy.each([&](int i, int j) {
      const int lo = i < j ? i : j, hi = i < j ? j : i;
      R acc = WWd[lo * y + hi];
      for (int k = 0; k < x; ++k) acc += N2[lo * x + k] * Fd[hi * x + k];
      N3[i * y + j] = acc;
    });
  };
  // ---- Kalman stages (kf.py:10-14); each is followed by a stage_end by the caller
  auto kal1 = [&]() {                                        // AP = A P
    bb.each([&](int i, int j) {
      R acc = R(0);
      for (int k = 0; k < b; ++k) acc += Aa[i * b + k] * P[k * b + j];
      AP[i * b + j] = acc;
    });
  };
  auto kal2 = [&]() {                                        // Pp = AP A' + V V'
    bb.each([&](int i, int j) {
      const int lo = i < j ? i : j, hi = i < j ? j : i;
      R acc = VVa[lo * b + hi];
      for (int k = 0; k < b; ++k) acc += AP[lo * b + k] * Aa[hi * b + k];
      Pp[i * b + j] = acc;
    });
  };
  auto kal3 = [&]() {                                        // FP = F Pp
    yb.each([&](int i, int j) {
      R acc = R(0);
      for (int k = 0; k < b; ++k) acc += Fa[i * b + k] * Pp[k * b + j];
      FP[i * b + j] = acc;
    });
  };
  auto kal4 = [&]() {                                        // Gk = FP F' + W W'
    yy.each([&](int i, int j) {
      const int lo = i < j ? i : j, hi = i < j ? j : i;
      R acc = WWa[lo * y + hi];
      for (int k = 0; k < b; ++k) acc += FP[lo * b + k] * Fa[hi * b + k];
      Gk[i * y + j] = acc;
    });
  };
  auto kal5 = [&](int t) {                                   // K = (F Pp)' Gk^-1 ; P = Pp - K F Pp (row of K per thread)
    R Gi[kMaxSmall * kMaxSmall], unused[kMaxSmall * kMaxSmall];
    reg_inverse<R>(y, Gk, R(0), false, Gi, unused);
    bb.each([&](int i, int j) {
      const int lo = i < j ? i : j, hi = i < j ? j : i;
      R acc = Pp[lo * b + hi];
      LQG_UNROLL for (int k = 0; k < kMaxSmall; ++k) {
        if (k < y) {
          R kik = R(0);                                      // K[lo, k]
          LQG_UNROLL for (int l2 = 0; l2 < kMaxSmall; ++l2)
            if (l2 < y) kik += FP[l2 * b + lo] * Gi[l2 * kMaxSmall + k];
          acc -= kik * FP[k * b + hi];
        }
      }
      P[i * b + j] = acc;
    });
    by.each([&](int i, int j) {
      R kij = R(0);
      LQG_UNROLL for (int l2 = 0; l2 < kMaxSmall; ++l2) {
        if (l2 < y) {
          R g = Gi[l2 * kMaxSmall];
          LQG_UNROLL for (int c = 1; c < kMaxSmall; ++c) g = (j == c) ? Gi[l2 * kMaxSmall + c] : g;
          kij += FP[l2 * b + i] * g;
        }
      }
      K[i * y + j] = kij;
      if (a.K.p) const_cast<R*>(a.K.p)[s * a.K.sb + (long)t * a.K.st + i * a.K.sr + j * a.K.sc] = kij;
    });
  };
  // ---- joint-system stages
  auto joint1 = [&](int t) {                                 // products that only need K_t, L_t and constants
    const R* Lg = a.Ls + (s * a.T + t) * (long)(u * b);
    bu.each([&](int i, int j) {                              // BK = Ba + K DB
      R acc = Ba[i * u + j];
      for (int k = 0; k < y; ++k) acc += K[i * y + k] * DB[k * u + j];
      BK[i * u + j] = acc;
    });
    bx.each([&](int i, int j) {                              // K FAd, K N2
      R a1 = R(0), a2 = R(0);
      for (int k = 0; k < y; ++k) { a1 += K[i * y + k] * FAd[k * x + j]; a2 += K[i * y + k] * N2[k * x + j]; }
      KFAd[i * x + j] = a1;
      KN2[i * x + j] = a2;
    });
    bb.each([&](int i, int j) {                              // K FAa
      R acc = R(0);
      for (int k = 0; k < y; ++k) acc += K[i * y + k] * FAa[k * b + j];
      KFAa[i * b + j] = acc;
    });
    xb.each([&](int i, int j) {                              // Bd L
      R acc = R(0);
      for (int k = 0; k < u; ++k) acc += Bd[i * u + k] * Lg[k * b + j];
      BdL[i * b + j] = acc;
    });
    by.each([&](int i, int j) {                              // K N3
      R acc = R(0);
      for (int k = 0; k < y; ++k) acc += K[i * y + k] * N3[k * y + j];
      KN3[i * y + j] = acc;
    });
    for (int e = threadIdx.x; e < u * b; e += BLOCK) Lm[e] = Lg[e];
  };
  auto joint2 = [&](bool first) {                            // Fj, GG (system.py:167-207); first: Sigma := GG
    mm_.each([&](int i, int j) {
      R f, g;
      if (i < x) {
        f = (j < x) ? Ad[i * x + j] : BdL[i * b + (j - x)];
      } else {
        const int ib = i - x;
        if (j < x) {
          f = KFAd[ib * x + j];
        } else {
          const int jb = j - x;
          R acc = Aa[ib * b + jb] - KFAa[ib * b + jb];
          for (int k = 0; k < u; ++k) acc += BK[ib * u + k] * Lm[k * b + jb];
          f = acc;
        }
      }
      const int lo = i < j ? i : j, hi = i < j ? j : i;
      if (hi < x) g = N1[lo * x + hi];
      else if (lo < x) g = KN2[(hi - x) * x + lo];
      else {
        R acc = R(0);
        for (int k = 0; k < y; ++k) acc += KN3[(lo - x) * y + k] * K[(hi - x) * y + k];
        g = acc;
      }
      Fj[i * m + j] = f;
      GG[i * m + j] = g;
      if (first) Sg[i * m + j] = g;                          // Sigma0 := G[0] G[0]'  system.py:212
    });
  };
  // ---- conditioning on the observed block of Sg
  auto cond1 = [&](int t, bool emit_f) {                     // Li, hl in registers; U2 = S_ro Li'; operator stream
    R Li[kMaxSmall * kMaxSmall], hl;
    reg_chol<R>(o, Sg, m, Li, hl);
    ro.each([&](int p, int j) {
      R acc = R(0);
      LQG_UNROLL for (int k = 0; k < kMaxSmall; ++k) {
        if (k < o) {
          R l = Li[k];                                       // Li[j, k], run-time row j
          LQG_UNROLL for (int r2 = 1; r2 < kMaxSmall; ++r2) l = (j == r2) ? Li[r2 * kMaxSmall + k] : l;
          acc += (k <= j) ? Sg[(o + p) * m + k] * l : R(0);
        }
      }
      U2[p * o + j] = acc;
      if (a.ops) a.ops[(s * (a.T + 1) + t) * (long)a.nops + m * m + p * o + j] = acc;
    });
    if (a.ops && threadIdx.x == 0) {
      R* op = a.ops + (s * (a.T + 1) + t) * (long)a.nops + m * m + rr * o;
      int e = 0;
      LQG_UNROLL for (int i = 0; i < kMaxSmall; ++i)
        LQG_UNROLL for (int j = 0; j <= i; ++j)
          if (i < o) op[e++] = Li[i * kMaxSmall + j];
      op[e] = hl + kLogNorm;
    }
    (void)emit_f;
  };
  auto cond2 = [&]() {                                       // C = S_rr - U2 U2'
    rrs.each([&](int p, int q2) {
      const int lo = p < q2 ? p : q2, hi = p < q2 ? q2 : p;
      R acc = Sg[(o + lo) * m + o + hi];
      for (int j = 0; j < o; ++j) acc -= U2[lo * o + j] * U2[hi * o + j];
      C[p * rr + q2] = acc;
    });
  };
  auto sig1 = [&](int t) {                                   // T1 = Fj[:, o:] C ; emit Fj - [[I_o,0],[0,0]]
    mr.each([&](int i, int q2) {
      R acc = R(0);
      for (int p = 0; p < rr; ++p) acc += Fj[i * m + o + p] * C[p * rr + q2];
      T1[i * rr + q2] = acc;
    });
    if (a.ops) {
      R* op = a.ops + (s * (a.T + 1) + t) * (long)a.nops;
      mm_.each([&](int i, int j) { op[i * m + j] = (i < o && i == j) ? Fj[i * m + j] - R(1) : Fj[i * m + j]; });
    }
  };
  auto sig2 = [&](int t) {                                   // Sigma' = T1 Fj[:, o:]' + GG          system.py:223-230
    mm_.each([&](int i, int j) {
      const int lo = i < j ? i : j, hi = i < j ? j : i;
      R acc = GG[lo * m + hi];
      for (int q2 = 0; q2 < rr; ++q2) acc += T1[lo * rr + q2] * Fj[hi * m + o + q2];
      Sg[i * m + j] = acc;
      if (a.Sig.p) const_cast<R*>(a.Sig.p)[s * a.Sig.sb + (long)t * a.Sig.st + i * a.Sig.sr + j * a.Sig.sc] = acc;
    });
  };

  // ---- prologue: P0, constants, Kalman step 0
  if (a.Sigma0.p) ld_sym(bb, a.Sigma0, s, 0, P);
  else ld_gram(bb, a.aV, s, 0, a.nva, P);                    // V[0] V[0]'  system.py:79,160
  load_consts(0);
  stage_end<BLOCK>();
  hoist1();
  stage_end<BLOCK>();
  hoist2();
  kal1();
  stage_end<BLOCK>();
  kal2();
  stage_end<BLOCK>();
  kal3();
  stage_end<BLOCK>();
  kal4();
  stage_end<BLOCK>();
  kal5(0);
  stage_end<BLOCK>();

  for (int t = 0; t < a.T; ++t) {
    const bool more = t + 1 < a.T;
    if (!a.ti) {
      // time-varying specs: K_{t+1} needs the constants of step t+1 while the joint system of step t needs those of
      // step t — no pipelining: joint stages first, then reload, then the Kalman step
      if (joint) {
        joint1(t);
        if (t > 0) cond1(t, true);
        stage_end<BLOCK>();
        joint2(t == 0);
        if (t > 0) cond2();
        stage_end<BLOCK>();
        if (t == 0) { cond1(0, true); stage_end<BLOCK>(); cond2(); stage_end<BLOCK>(); }
        sig1(t);
        stage_end<BLOCK>();
        sig2(t);
        stage_end<BLOCK>();
      }
      if (more) {
        load_consts(t + 1);
        stage_end<BLOCK>();
        hoist1();
        stage_end<BLOCK>();
        hoist2();
        kal1();
        stage_end<BLOCK>();
        kal2();
        stage_end<BLOCK>();
        kal3();
        stage_end<BLOCK>();
        kal4();
        stage_end<BLOCK>();
        kal5(t + 1);
        stage_end<BLOCK>();
      }
      continue;
    }
    // ---- time-invariant: five stages per step
    if (joint) { joint1(t); if (t > 0) cond1(t, true); }     // A
    if (more) kal1();
    stage_end<BLOCK>();
    if (joint) { joint2(t == 0); if (t > 0) cond2(); }       // B
    if (more) kal2();
    stage_end<BLOCK>();
    if (joint && t == 0) { cond1(0, true); stage_end<BLOCK>(); cond2(); stage_end<BLOCK>(); }
    if (joint) sig1(t);                                      // C
    if (more) kal3();
    stage_end<BLOCK>();
    if (joint) sig2(t);                                      // D
    if (more) kal4();
    stage_end<BLOCK>();
    if (more) { kal5(t + 1); stage_end<BLOCK>(); }           // E   (K_t is dead after stage B)
  }
  // ---- last row: only the density operators of x_T
  if (joint && a.ops) {
    R Li[kMaxSmall * kMaxSmall], hl;
    reg_chol<R>(o, Sg, m, Li, hl);
    if (threadIdx.x == 0) {
      R* op = a.ops + (s * (a.T + 1) + a.T) * (long)a.nops + m * m + rr * o;
      int e = 0;
      LQG_UNROLL for (int i = 0; i < kMaxSmall; ++i)
        LQG_UNROLL for (int j = 0; j <= i; ++j)
          if (i < o) op[e++] = Li[i * kMaxSmall + j];
      op[e] = hl + kLogNorm;
    }
  }
}

// ======================================================================================================================
// Generic per-trial sweep over the operator stream (run-time m, d): mean recursion + Gaussian log-density
// (system.py:219-221, 244-248).  One trial per thread, its state in LDS columns [element][thread] (conflict-free).
// Used when k_trial<M, ND> has no instantiation for the shape (large m); the instantiated kernel reads the same stream.
template <typename R>
struct TrialArgsRT {
  DTraj<R> x, mu;
  R* ll;
  long ll_sb, ll_sn, n_trials;
  int T, m, d, nops;
};

template <typename R, int BLOCK>
__global__ void __launch_bounds__(BLOCK) k_coop_trial(const R* __restrict__ ops_all, const TrialArgsRT<R> a) {
  extern __shared__ double lqg_coop_smem[];
  R* sm = reinterpret_cast<R*>(lqg_coop_smem);
  const int m = a.m, o = a.d, rr = m - a.d, tid = threadIdx.x;
  const long sys = blockIdx.y;
  long n = (long)blockIdx.x * BLOCK + tid;
  const bool live = n < a.n_trials;
  n = live ? n : a.n_trials - 1;
  // per-thread columns: xprev[o] dO[o] xt[o] w[o] muR[rr] c[rr] mn[m]
  R *xprev = sm, *dO = xprev + o * BLOCK, *xt = dO + o * BLOCK, *w = xt + o * BLOCK, *muR = w + o * BLOCK,
    *c = muR + rr * BLOCK, *mn = c + rr * BLOCK;
#define AT(arr, i) arr[(i) * BLOCK + tid]
  const R* xr = a.x.p + sys * a.x.sb + n * a.x.sn;
  for (int i = 0; i < o; ++i) { AT(xprev, i) = xr[i * a.x.sd]; AT(dO, i) = R(0); }
  for (int i = 0; i < rr; ++i) AT(muR, i) = R(0);
  double acc = 0.0;
  const R* op = ops_all + sys * (long)(a.T + 1) * a.nops;
  const int U_OFF = m * m, L_OFF = U_OFF + rr * o, H_OFF = L_OFF + o * (o + 1) / 2;
  for (int t = 0; t <= a.T; ++t, op += a.nops) {
    for (int i = 0; i < o; ++i) AT(xt, i) = xr[(long)t * a.x.st + i * a.x.sd];
    R zz = R(0);
    int e = 0;
    for (int i = 0; i < o; ++i) {
      R v = R(0);
      for (int j = 0; j <= i; ++j) v += op[L_OFF + (e++)] * ((AT(xt, j) - AT(xprev, j)) - AT(dO, j));
      AT(w, i) = v;
      zz += v * v;
    }
    if (t > 0) acc -= (double)(R(0.5) * zz + op[H_OFF]);
    if (t < a.T) {
      for (int p = 0; p < rr; ++p) {
        R v = AT(muR, p);
        for (int j = 0; j < o; ++j) v += op[U_OFF + p * o + j] * AT(w, j);
        AT(c, p) = v;
      }
      for (int i = 0; i < m; ++i) {
        R v = R(0);
        for (int j = 0; j < o; ++j) v += op[i * m + j] * AT(xt, j);
        for (int p = 0; p < rr; ++p) v += op[i * m + o + p] * AT(c, p);
        AT(mn, i) = v;
      }
      for (int i = 0; i < o; ++i) { AT(dO, i) = AT(mn, i); AT(xprev, i) = AT(xt, i); }
      for (int p = 0; p < rr; ++p) AT(muR, p) = AT(mn, o + p);
      if (a.mu.p && live) {
        R* dst = const_cast<R*>(a.mu.p) + sys * a.mu.sb + n * a.mu.sn + (long)t * a.mu.st;
        for (int i = 0; i < m; ++i) dst[i * a.mu.sd] = (i < o) ? AT(xt, i) + AT(mn, i) : AT(mn, i);
      }
    }
  }
#undef AT
  if (a.ll && live) a.ll[sys * a.ll_sb + n * a.ll_sn] = (R)acc;
}

}  // namespace coop
}  // namespace lqg
