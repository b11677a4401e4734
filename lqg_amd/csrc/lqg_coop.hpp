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
#include <type_traits>

#include "lqg_kernels.hpp"

namespace lqg {
namespace coop {

constexpr int kMaxSmall = 4;    // u, y, d <= 4: their u x u / y x y / d x d factorizations run in registers

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
  int sparse;       // 1: run-time sparsity lists (RowLists) for the large products, at the START of the dynamic LDS
  long lists_bytes; // bytes of those lists (0 when !sparse): the LDS part of the arena starts behind them
  long lds_reals;   // GLOBAL (hybrid) variant: reals of LDS the arena may use before it continues in `arena` (L2-resident)
  R eps;
};

// The lanes that share one product: lane `ln` of STRIDE (a whole workgroup, or — WAVES mode — one wave of it).
// (row, col) of this lane's first element of an r x c result; later elements (r*c > STRIDE) by division.
template <int STRIDE>
struct Shape {
  int rows, cols, n, ln, i0, j0, qs, rs;
  LQG_DEV Shape(int lane, int r, int c) : rows(r), cols(c), n(r * c), ln(lane) {
    const int cc = c > 0 ? c : 1;
    i0 = lane / cc;
    j0 = lane - i0 * c;
    qs = STRIDE / cc;                      // (row, col) advance of one stride: the division is paid once per Shape, not
    rs = STRIDE - qs * cc;                 // once per element (a run-time integer division is ~35 instructions)
  }
  template <typename F>
  LQG_DEV void each(F f) const {
    if (ln < n) f(i0, j0);
    if (n > STRIDE) {
      int i = i0, j = j0;
      for (int e = ln + STRIDE; e < n; e += STRIDE) {
        i += qs;
        j += rs;
        if (j >= cols) { j -= cols; ++i; }
        f(i, j);
      }
    }
  }
};

// acc + sum_{k<K} a[k sa] b[k sb], terms added in order of k.  The operands of FOUR terms are fetched before their
// multiply-adds, so a block's loads are in flight together (one LDS round trip per block instead of one per term);
// the tail block clamps its indices and masks its terms instead of branching.
template <typename R>
LQG_DEV R dot4(const R* __restrict__ a, int sa, const R* __restrict__ b, int sb, int K, R acc) {
  int k = 0;
  for (; k + 4 <= K; k += 4) {
    const R a0 = a[k * sa], a1 = a[(k + 1) * sa], a2 = a[(k + 2) * sa], a3 = a[(k + 3) * sa];
    const R b0 = b[k * sb], b1 = b[(k + 1) * sb], b2 = b[(k + 2) * sb], b3 = b[(k + 3) * sb];
    acc += a0 * b0;
    acc += a1 * b1;
    acc += a2 * b2;
    acc += a3 * b3;
  }
  const int r = K - k;
  if (r > 0) {
    const int k1 = r > 1 ? k + 1 : k, k2 = r > 2 ? k + 2 : k;
    const R a0 = a[k * sa], a1 = a[k1 * sa], a2 = a[k2 * sa];
    const R b0 = b[k * sb], b1 = b[k1 * sb], b2 = b[k2 * sb];
    acc += a0 * b0;
    acc += (r > 1) ? a1 * b1 : R(0);
    acc += (r > 2) ? a2 * b2 : R(0);
  }
  return acc;
}

// ---- run-time sparsity of the LEFT operand of the large products ---------------------------------------------------------
// The reference's largest models are delay augmentations (lqg/tracking/delay.py:9-33): A = blockdiag(A0, 0) + a shifted
// identity, so A, and with it the joint dynamics Fj, hold one or two non-zeros per row — at m = 65 the dense b^3 / m^3
// products of a step are 95 % multiplications by an exact zero (37 ms per forward sweep of T = 500, measured).  Per row
// of such an operand the columns of its non-zeros are listed once (per step for matrices that change with the step);
// a product then walks the list.  Exact: the terms are added in the same (increasing) order as dot4 adds them and a
// skipped term is an exact zero times a finite number.
struct RowLists {
  unsigned char* cnt;      // [rows]
  unsigned char* idx;      // [rows][cols]
  int cols;
};
inline __host__ __device__ long row_lists_bytes(int rows, int cols) { return ((long)rows * cols + rows + 15) / 16 * 16; }
LQG_DEV RowLists take_lists(unsigned char*& p, int rows, int cols) {
  RowLists r{p, p + rows, cols};
  p += row_lists_bytes(rows, cols);
  return r;
}
// lists of the rows of the rows x cols matrix M(i, k) = M[i * rs + k * cs].  One WAVE per row: lane k tests column
// k (+ 64, + 128, ...), a ballot gives the row's non-zero mask and every lane the rank of its column among them — a row
// costs one LDS round trip instead of `cols` dependent ones (65 rows of 63 columns: ~6000 cycles on one lane each).
template <int STRIDE, typename R>
LQG_DEV void build_lists(int ln, const R* __restrict__ M, int rs, int cs, int rows, int cols, const RowLists& rl) {
  constexpr int NWV = STRIDE / 64;
  const int wv = ln >> 6, lane = ln & 63;
  for (int i = wv; i < rows; i += NWV) {
    unsigned char* ix = rl.idx + i * rl.cols;
    int n = 0;
    for (int k0 = 0; k0 < cols; k0 += 64) {
      const int k = k0 + lane;
      const bool nz = k < cols && M[i * rs + k * cs] != R(0);
      const unsigned long long mask = __ballot(nz);
      if (nz) ix[n + __popcll(mask & ((1ull << lane) - 1ull))] = (unsigned char)k;
      n += __popcll(mask);
    }
    if (lane == 0) rl.cnt[i] = (unsigned char)n;
  }
}
// acc + sum over the listed k of a[k sa] b[k sb] (a: the listed row), four terms' operands in flight together
template <typename R>
LQG_DEV R dot_list(const RowLists& rl, int row, const R* __restrict__ a, int sa, const R* __restrict__ b, int sb, R acc) {
  const unsigned char* __restrict__ ix = rl.idx + row * rl.cols;
  const int n = rl.cnt[row];
  int q = 0;
  for (; q + 4 <= n; q += 4) {
    const int k0 = ix[q], k1 = ix[q + 1], k2 = ix[q + 2], k3 = ix[q + 3];
    const R a0 = a[k0 * sa], a1 = a[k1 * sa], a2 = a[k2 * sa], a3 = a[k3 * sa];
    const R b0 = b[k0 * sb], b1 = b[k1 * sb], b2 = b[k2 * sb], b3 = b[k3 * sb];
    acc += a0 * b0;
    acc += a1 * b1;
    acc += a2 * b2;
    acc += a3 * b3;
  }
  for (; q < n; ++q) {
    const int k = ix[q];
    acc += a[k * sa] * b[k * sb];
  }
  return acc;
}

#ifdef LQG_COOP_STAMP
// developer build (-DLQG_COOP_STAMP, variant library): cycles per stage of the pipelined forward loop, accumulated by lane 0 of
// system 0 (read back with lqg_debug_coop_stamps of lqg_coop_inst.hip; scripts/coop_stamps.py)
__device__ unsigned long long g_coop_stamps[16];
#define LQG_STAMP(slot_)                                                     \
  do {                                                                       \
    if (threadIdx.x == 0 && blockIdx.x == 0) {                               \
      const unsigned long long now_ = __builtin_readcyclecounter();          \
      g_coop_stamps[slot_] += now_ - stamp_prev_;                            \
      stamp_prev_ = now_;                                                    \
    }                                                                        \
  } while (0)
#else
#define LQG_STAMP(slot_) do { } while (0)
#endif

template <int BLOCK>
LQG_DEV void stage_end() {
  __syncthreads();   // workgroup barrier + LDS/global visibility inside the workgroup
}

// ---- strided global -> arena loads (cooperative) ---------------------------------------------------------------------
template <int STRIDE, typename R>
LQG_DEV void ld_mat(const Shape<STRIDE>& sh, const DView<R>& v, long s, int t, R* dst) {
  const R* p = v.p + s * v.sb + (long)t * v.st;
  sh.each([&](int i, int j) { dst[i * sh.cols + j] = p[i * v.sr + j * v.sc]; });
}
template <int STRIDE, typename R>
LQG_DEV void ld_sym(const Shape<STRIDE>& sh, const DView<R>& v, long s, int t, R* dst) {   // symmetric part
  const R* p = v.p + s * v.sb + (long)t * v.st;
  sh.each([&](int i, int j) {
    dst[i * sh.cols + j] = (i == j) ? p[i * v.sr + i * v.sc] : R(0.5) * (p[i * v.sr + j * v.sc] + p[j * v.sr + i * v.sc]);
  });
}
template <int STRIDE, typename R>
LQG_DEV void ld_gram(const Shape<STRIDE>& sh, const DView<R>& v, long s, int t, int nv, R* dst) {   // V V^T
  const R* p = v.p + s * v.sb + (long)t * v.st;
  sh.each([&](int i, int j) {
    const int lo = i < j ? i : j, hi = i < j ? j : i;       // (min, max): both mirror entries get the same bits
    R acc = R(0);
    for (int k = 0; k < nv; ++k) acc += p[lo * v.sr + k * v.sc] * p[hi * v.sr + k * v.sc];
    dst[i * sh.cols + j] = acc;
  });
}
template <int STRIDE, typename R>
LQG_DEV void ld_vec(int ln, const DView<R>& v, long s, int t, int n, R* dst) {
  for (int i = ln; i < n; i += STRIDE) dst[i] = v.p ? v.p[s * v.sb + (long)t * v.st + i * v.sr] : R(0);
}

// ---- small symmetric positive-definite factorizations in registers --------------------------------------------------
// u, y, d <= kMaxSmall.  The extent is a COMPILE-TIME constant inside each case of dispatch_small and the factors are
// handed to the other lanes through LDS (lane 0 stores them with constant indices): a register array must never be
// indexed at run time — it would live in scratch, i.e. in global memory, thousands of cycles per step (measured: the
// first version of this file ran 7 k cycles per Riccati step that way; even chains of selects are turned into one).
// C > 0: the extent is a template constant of the kernel (per-shape instantiation) — no switch
template <int C, typename F>
LQG_DEV void dispatch_small(int n, F f) {
  if constexpr (C > 0) {
    f(std::integral_constant<int, C>{});
    return;
  }
  switch (n) {
    case 1: f(std::integral_constant<int, 1>{}); break;
    case 2: f(std::integral_constant<int, 2>{}); break;
    case 3: f(std::integral_constant<int, 3>{}); break;
    default: f(std::integral_constant<int, 4>{}); break;
  }
}
// Hi = (H + max(0, eps - lambda_min(H)) I)^-1 and the regularised Ht (lqr.py:27-31); floor_ = false: plain inverse
template <typename R, int N>
LQG_DEV void spd_inverse_reg(const R* Hs, R eps, bool floor_, R (&Hi)[N * N], R (&Ht)[N * N]) {
  R H[N * N], Lc[N * N], dinv[N], Li[N * N];
  LQG_UNROLL for (int i = 0; i < N * N; ++i) H[i] = Hs[i];
  LQG_UNROLL for (int i = 0; i < N * N; ++i) Ht[i] = H[i];
  if (floor_) {
    R shift = eps - min_eig_sym<R, N>(H);
    shift = (shift > R(0)) ? shift : R(0);
    LQG_UNROLL for (int i = 0; i < N; ++i) Ht[i * N + i] += shift;
  }
  chol_lower<R, N>(Ht, Lc, dinv);
  tri_inverse_lower<R, N>(Lc, dinv, Li);
  spd_inverse_from_tri<R, N>(Li, Hi);
}
// Li = chol(S_oo)^-1 (lower), half log-determinant; S_oo = leading N x N block of a matrix with leading dimension ld
template <typename R, int N>
LQG_DEV void chol_inverse_reg(const R* Sg, int ld, R (&Li)[N * N], R& hl) {
  R A[N * N], Lc[N * N], dinv[N];
  LQG_UNROLL for (int i = 0; i < N; ++i)
    LQG_UNROLL for (int j = 0; j < N; ++j) A[i * N + j] = Sg[i * ld + j];
  chol_lower<R, N>(A, Lc, dinv);
  tri_inverse_lower<R, N>(Lc, dinv, Li);
  R pd = dinv[0];
  LQG_UNROLL for (int i = 1; i < N; ++i) pd *= dinv[i];
  hl = -log_<R>(pd);
}

// reals of the arena each kernel carves (host and device agree through these)
inline __host__ __device__ long riccati_arena_reals(int b, int u) {
  return 4L * b * b + 6L * b * u + 4L * u * u + 4L * b + 5L * u + 16;
}
inline __host__ __device__ long kalman_arena_reals(int b, int y) { return 5L * b * b + 3L * y * b + 3L * y * y + 16; }
inline __host__ __device__ long forward_arena_reals(int x, int b, int u, int y, int d) {
  const long m = x + b, o = d, rr = m - d;
  return /*Aa VVa P AP Pp KFAa*/ 6L * b * b + /*Ba BK*/ 2L * b * u + /*Fa FAa FP*/ 3L * y * b + /*WWa Gk Gis N3 WWd*/ 5L * y * y + /*Lis*/ 1L * d * d +
         /*Ad N1*/ 2L * x * x + /*Bd*/ 1L * x * u + /*FAd N2 Fd*/ 3L * y * x + /*DB*/ 1L * y * u + /*K KN3*/ 2L * b * y +
         /*L*/ 1L * u * b + /*KFAd BdL KN2*/ 3L * b * x + /*Fj GG Sg*/ 3L * m * m + /*U2*/ rr * o + /*C*/ rr * rr + /*T1*/ m * rr + 16;
}

// ======================================================================================================================
// Riccati backward (lqr.py:16-42): carries S[b,b], s[b]; emits L_t (scratch + optional L, l, H outputs)
// CB, CU > 0: dimensions fixed at compile time (one instantiation per shape of lqg_dims.def): every index computation,
// loop bound and LDS address folds to a constant and the inner products unroll — ~5x fewer instructions per stage than
// the run-time-dims instantiation (CB = CU = 0), which stays the path for every other shape.
template <typename R, int BLOCK, bool GLOBAL, bool WAVES, int CB = 0, int CU = 0>
__global__ void __launch_bounds__(BLOCK) k_coop_riccati(const Args<R> a) {
  extern __shared__ double lqg_coop_smem[];
  const long s = blockIdx.x;
  // WAVES: the independent products of a stage go to DIFFERENT waves (SIMDs) of the workgroup — within one wave they
  // would only queue behind each other in its instruction stream — and each product is spread over that wave's 64 lanes;
  // otherwise (large matrices) every product is spread over the whole workgroup.
  constexpr int STRIDE = WAVES ? 64 : BLOCK;
  constexpr int NW = BLOCK / 64;
  const int wv = WAVES ? (int)(threadIdx.x >> 6) : 0;
  const int ln = WAVES ? (int)(threadIdx.x & 63) : (int)threadIdx.x;
  auto on = [&](int w) { return !WAVES || wv == (w % NW); };
  // the working set: in LDS behind the sparsity lists; GLOBAL = HYBRID: the arrays are handed out from LDS in the order of
  // the take() calls below until a.lds_reals are used, the rest lives in the global arena (L2-resident)
  R* ar = reinterpret_cast<R*>(reinterpret_cast<unsigned char*>(lqg_coop_smem) + a.lists_bytes);
  [[maybe_unused]] R* gar = GLOBAL ? a.arena + s * a.arena_reals : nullptr;
  [[maybe_unused]] long lds_left = GLOBAL ? a.lds_reals : 0;
  const int b = CB ? CB : a.b, u = CU ? CU : a.u;
  const bool affine = a.aq.p || a.aqf.p || a.aP.p || a.ar.p;
  auto take = [&](long n) {
    if constexpr (GLOBAL) {
      if (n > lds_left) { R* p = gar; gar += n; return p; }
      lds_left -= n;
    }
    R* p = ar;
    ar += n;
    return p;
  };
  R *S = take(b * b), *A = take(b * b), *Q = take(b * b), *SA = take(b * b);
  R *Bm = take(b * u), *SB = take(b * u), *P = take(b * u), *G = take(b * u), *Lm = take(b * u), *W1 = take(b * u);
  R *Rm = take(u * u), *H = take(u * u), *His = take(u * u), *Hts = take(u * u);
  R *sv = take(b), *sn = take(b), *q = take(b), *gv = take(u), *lv = take(u), *Hl = take(u), *r = take(u);
  const Shape<STRIDE> bb(ln, b, b), bu(ln, b, u), ub(ln, u, b), uu(ln, u, u);
  // run-time sparsity (RowLists): the columns of A, i.e. the rows of A' (LDS behind the arena, or alone in LDS when the
  // arena is global)
  const bool sp = !WAVES && CB == 0 && a.sparse;
  unsigned char* slp = reinterpret_cast<unsigned char*>(lqg_coop_smem);
  const RowLists colA = sp ? take_lists(slp, b, b) : RowLists{nullptr, nullptr, 0};

  auto load_step = [&](int t) {
    if (on(0)) ld_mat(bb, a.aA, s, t, A);
    if (on(1)) ld_mat(bu, a.aB, s, t, Bm);
    if (on(2)) ld_sym(bb, a.aQ, s, t, Q);
    if (on(3)) ld_sym(uu, a.aR, s, t, Rm);
    if (affine && on(3)) {
      if (a.aP.p) ld_mat(ub, a.aP, s, t, P);
      else ub.each([&](int i, int j) { P[i * b + j] = R(0); });
      ld_vec<STRIDE>(ln, a.aq, s, t, b, q);
      ld_vec<STRIDE>(ln, a.ar, s, t, u, r);
    }
  };
  if (on(0)) ld_sym(bb, a.aQf, s, 0, S);                    // carry init (Qf, qf)  lqr.py:38
  if (affine && on(1)) ld_vec<STRIDE>(ln, a.aqf, s, 0, b, sv);
  if (a.ti) load_step(0);
  stage_end<BLOCK>();
  if (sp && a.ti) { build_lists<STRIDE>(ln, A, 1, b, b, b, colA); stage_end<BLOCK>(); }

  for (int t = a.T - 1; t >= 0; --t) {                      // reverse=True  lqr.py:40
    if (!a.ti) {
      load_step(t);
      stage_end<BLOCK>();
      if (sp) { build_lists<STRIDE>(ln, A, 1, b, b, b, colA); stage_end<BLOCK>(); }
    }
    // ---- R1: SA = S A, SB = S B
    if (on(0)) bb.each([&](int i, int j) {
      R acc = R(0);
      acc = sp ? dot_list(colA, j, A + j, b, S + i * b, 1, acc) : dot4(S + i * b, 1, A + j, b, b, acc);
      SA[i * b + j] = acc;
    });
    if (on(1)) bu.each([&](int i, int j) {
      R acc = R(0);
      acc = dot4(S + i * b, 1, Bm + j, u, b, acc);
      SB[i * u + j] = acc;
    });
    stage_end<BLOCK>();
    // ---- R2: H = R + B'SB (symmetric), G = P + B'SA, g = r + B's                     lqr.py:22-24
    if (on(0)) uu.each([&](int i, int j) {
      const int lo = i < j ? i : j, hi = i < j ? j : i;
      R acc = Rm[lo * u + hi];
      acc = dot4(Bm + lo, u, SB + hi, u, b, acc);
      H[i * u + j] = acc;
    });
    if (on(1)) ub.each([&](int i, int j) {
      R acc = affine ? P[i * b + j] : R(0);
      acc = dot4(Bm + i, u, SA + j, b, b, acc);
      G[i * b + j] = acc;
    });
    if (affine && on(2))
      for (int i = ln; i < u; i += STRIDE) {
        R acc = r[i];
        acc = dot4(Bm + i, u, sv, 1, b, acc);
        gv[i] = acc;
      }
    stage_end<BLOCK>();
    // ---- R3a: Ht^-1 (registers, compile-time extent) -> LDS                                          lqr.py:27-31
    if (on(0)) dispatch_small<CU>(u, [&](auto nu_) {
      constexpr int NU = decltype(nu_)::value;
      R Hi[NU * NU], Ht[NU * NU];
      spd_inverse_reg<R, NU>(H, a.eps, true, Hi, Ht);
      if (ln == 0) {
        LQG_UNROLL for (int e = 0; e < NU * NU; ++e) { His[e] = Hi[e]; Hts[e] = Ht[e]; }
      }
    });
    stage_end<BLOCK>();
    // ---- R3b: L = -Ht^-1 G, W1 = H L + G (each thread recomputes its column of L)                     lqr.py:30-33
    if (on(0)) ub.each([&](int i, int j) {
      R lij = R(0), w = G[i * b + j];
      for (int k = 0; k < u; ++k) {
        const R lkj = -dot4(His + k * u, 1, G + j, b, u, R(0));
        lij = (k == i) ? lkj : lij;
        w += H[i * u + k] * lkj;
      }
      Lm[i * b + j] = lij;
      W1[i * b + j] = w;
      if (a.Ls) a.Ls[(s * a.T + t) * (long)(u * b) + i * b + j] = lij;
      if (a.L.p) const_cast<R*>(a.L.p)[s * a.L.sb + (long)t * a.L.st + i * a.L.sr + j * a.L.sc] = lij;
    });
    if (a.H.p && on(1))                                          // regularised Ht  lqr.py:36
      uu.each([&](int i, int j) { const_cast<R*>(a.H.p)[s * a.H.sb + (long)t * a.H.st + i * a.H.sr + j * a.H.sc] = Hts[i * u + j]; });
    if (affine && on(2)) {
      for (int i = ln; i < u; i += STRIDE) {             // l = -Ht^-1 g, Hl = H l + g
        R hl_ = gv[i], li = R(0);
        for (int k = 0; k < u; ++k) {
          const R lk = -dot4(His + k * u, 1, gv, 1, u, R(0));
          li = (k == i) ? lk : li;
          hl_ += H[i * u + k] * lk;
        }
        lv[i] = li;
        Hl[i] = hl_;
      }
    } else if (!affine && a.l.p && on(2)) {
      for (int i = ln; i < u; i += STRIDE) const_cast<R*>(a.l.p)[s * a.l.sb + (long)t * a.l.st + i * a.l.sr] = R(0);
    }
    stage_end<BLOCK>();
    // ---- R4: S = Q + A'SA + L'(HL + G) + G'L (symmetric); s = q + A's + G'l + L'(Hl + g)            lqr.py:33-34
    if (on(0)) bb.each([&](int i, int j) {
      const int lo = i < j ? i : j, hi = i < j ? j : i;
      R acc = Q[lo * b + hi];
      acc = sp ? dot_list(colA, lo, A + lo, b, SA + hi, b, acc) : dot4(A + lo, b, SA + hi, b, b, acc);
      acc = dot4(Lm + lo, b, W1 + hi, b, u, acc);
      acc = dot4(G + lo, b, Lm + hi, b, u, acc);
      S[i * b + j] = acc;
    });
    if (affine && on(1)) {
      for (int i = ln; i < b; i += STRIDE) {
        R acc = q[i];
        acc = dot4(A + i, b, sv, 1, b, acc);
        acc = dot4(G + i, b, lv, 1, u, acc);
        acc = dot4(Lm + i, b, Hl, 1, u, acc);
        sn[i] = acc;
      }
      if (a.l.p)
        for (int i = ln; i < u; i += STRIDE) const_cast<R*>(a.l.p)[s * a.l.sb + (long)t * a.l.st + i * a.l.sr] = lv[i];
    }
    stage_end<BLOCK>();
    if (affine && on(1)) {
      for (int i = ln; i < b; i += STRIDE) sv[i] = sn[i];
      // (sv is next read in R2 of the following step, after two more barriers)
    }
  }
}

// ======================================================================================================================
// Forward sweep: Kalman recursion (kf.py:6-21), joint system (system.py:167-207), Schur-form moment recursion
// (system.py:209-235), trial operators per step.  Kalman step t+1 is pipelined under the joint / Sigma stages of step t.
template <typename R, int BLOCK, bool GLOBAL, bool WAVES, int CX = 0, int CB = 0, int CU = 0, int CY = 0, int CD = 0>
__global__ void __launch_bounds__(BLOCK) k_coop_forward(const Args<R> a) {
  extern __shared__ double lqg_coop_smem[];
  const long s = blockIdx.x;
  // WAVES: the independent products of a stage go to DIFFERENT waves (SIMDs) of the workgroup — within one wave they
  // would only queue behind each other in its instruction stream — and each product is spread over that wave's 64 lanes;
  // otherwise (large matrices) every product is spread over the whole workgroup.
  constexpr int STRIDE = WAVES ? 64 : BLOCK;
  constexpr int NW = BLOCK / 64;
  const int wv = WAVES ? (int)(threadIdx.x >> 6) : 0;
  const int ln = WAVES ? (int)(threadIdx.x & 63) : (int)threadIdx.x;
  auto on = [&](int w) { return !WAVES || wv == (w % NW); };
  // the working set: in LDS behind the sparsity lists; GLOBAL = HYBRID: the arrays are handed out from LDS in the order of
  // the take() calls below until a.lds_reals are used, the rest lives in the global arena (L2-resident)
  R* ar = reinterpret_cast<R*>(reinterpret_cast<unsigned char*>(lqg_coop_smem) + a.lists_bytes);
  [[maybe_unused]] R* gar = GLOBAL ? a.arena + s * a.arena_reals : nullptr;
  [[maybe_unused]] long lds_left = GLOBAL ? a.lds_reals : 0;
  const int x = CX ? CX : a.x, b = CB ? CB : a.b, u = CU ? CU : a.u, y = CY ? CY : a.y, o = CD ? CD : a.d, m = x + b,
            rr = m - o;
  auto take = [&](long n) {
    if constexpr (GLOBAL) {
      if (n > lds_left) { R* p = gar; gar += n; return p; }
      lds_left -= n;
    }
    R* p = ar;
    ar += n;
    return p;
  };
  // (the arrays of the Kalman recursion come first: a gains-only call, kf.forward, needs kalman_arena_reals only)
  R *Aa = take(b * b), *VVa = take(b * b), *P = take(b * b), *AP = take(b * b), *Pp = take(b * b);
  R *Fa = take(y * b), *FP = take(y * b), *WWa = take(y * y), *Gk = take(y * y), *Gis = take(y * y), *K = take(b * y);
  R *KFAa = take(b * b), *Ba = take(b * u), *BK = take(b * u), *FAa = take(y * b);
  R *N3 = take(y * y), *WWd = take(y * y);
  R *Ad = take(x * x), *N1 = take(x * x), *Bd = take(x * u);
  R *FAd = take(y * x), *N2 = take(y * x), *Fd = take(y * x), *DB = take(y * u);
  R *KN3 = take(b * y), *Lm = take(u * b);
  R *KFAd = take(b * x), *BdL = take(x * b), *KN2 = take(b * x);
  R *Fj = take(m * m), *GG = take(m * m), *Sg = take(m * m), *U2 = take(rr * o), *C = take(rr * rr), *T1 = take(m * rr);
  R* Lis = take(o * o);
  const Shape<STRIDE> bb(ln, b, b), bu(ln, b, u), yb(ln, y, b), yy(ln, y, y), xx(ln, x, x), xu(ln, x, u), yx(ln, y, x),
      yu(ln, y, u), by(ln, b, y), bx(ln, b, x), xb(ln, x, b), mm_(ln, m, m), ro(ln, rr, o), rrs(ln, rr, rr), mr(ln, m, rr);
  const bool joint = a.ops || a.Sig.p;                      // false: only the Kalman gains are wanted (kf.forward)
  const R kLogNorm = R(0.5 * 1.8378770664093453) * (R)o;
  // run-time sparsity (RowLists): the rows of Aa (Kalman products) and of F2 = Fj[:, o:] (moment recursion)
  const bool sp = !WAVES && CB == 0 && a.sparse;
  unsigned char* slp = reinterpret_cast<unsigned char*>(lqg_coop_smem);
  const RowLists rowA = sp ? take_lists(slp, b, b) : RowLists{nullptr, nullptr, 0};
  const RowLists rowF = sp ? take_lists(slp, m, rr) : RowLists{nullptr, nullptr, 0};

  auto load_consts = [&](int t) {                            // stage L1: direct loads and Gram matrices
    if (on(0)) ld_mat(bb, a.aA, s, t, Aa);
    if (on(1)) ld_mat(yb, a.aF, s, t, Fa);
    if (on(2)) ld_gram(bb, a.aV, s, t, a.nva, VVa);
    if (on(3)) ld_gram(yy, a.aW, s, t, a.nwa, WWa);
    if (joint) {
      if (on(0)) ld_mat(bu, a.aB, s, t, Ba);
      if (on(1)) { ld_mat(xx, a.dA, s, t, Ad); ld_mat(xu, a.dB, s, t, Bd); }
      if (on(2)) ld_mat(yx, a.dF, s, t, Fd);
      if (on(3)) { ld_gram(xx, a.dV, s, t, a.nvd, N1); ld_gram(yy, a.dW, s, t, a.nwd, WWd); }
    }
  };
  auto hoist1 = [&]() {                                      // stage L2: Fa Aa, Fd Ad, Fd Bd - Fa Ba, Fd Vd Vd'
    if (!joint) return;
    if (on(0)) yb.each([&](int i, int j) {
      R acc = R(0);
      acc = dot4(Fa + i * b, 1, Aa + j, b, b, acc);
      FAa[i * b + j] = acc;
    });
    if (on(1)) yx.each([&](int i, int j) {
      R acc = R(0), acc2 = R(0);
      acc = dot4(Fd + i * x, 1, Ad + j, x, x, acc);
      acc2 = dot4(Fd + i * x, 1, N1 + j, x, x, acc2);
      FAd[i * x + j] = acc;
      N2[i * x + j] = acc2;
    });
    if (on(2)) yu.each([&](int i, int j) {
      R f1 = R(0), f2 = R(0);
      f1 = dot4(Fd + i * x, 1, Bd + j, u, x, f1);
      f2 = dot4(Fa + i * b, 1, Ba + j, u, b, f2);
      DB[i * u + j] = f1 - f2;                               // system.py:177-180
    });
  };
  auto hoist2 = [&]() {                                      // stage L3: N3 = Fd Vd Vd' Fd' + Wd Wd'
    if (!joint) return;
    if (on(0)) yy.each([&](int i, int j) {
      const int lo = i < j ? i : j, hi = i < j ? j : i;
      R acc = WWd[lo * y + hi];
      acc = dot4(N2 + lo * x, 1, Fd + hi * x, 1, x, acc);
      N3[i * y + j] = acc;
    });
  };
  // ---- Kalman stages (kf.py:10-14); each is followed by a stage_end by the caller
  auto kal1 = [&]() {                                        // AP = A P
    if (on(3)) bb.each([&](int i, int j) {
      R acc = R(0);
      acc = sp ? dot_list(rowA, i, Aa + i * b, 1, P + j, b, acc) : dot4(Aa + i * b, 1, P + j, b, b, acc);
      AP[i * b + j] = acc;
    });
  };
  auto kal2 = [&]() {                                        // Pp = AP A' + V V'
    if (on(3)) bb.each([&](int i, int j) {
      const int lo = i < j ? i : j, hi = i < j ? j : i;
      R acc = VVa[lo * b + hi];
      acc = sp ? dot_list(rowA, hi, Aa + hi * b, 1, AP + lo * b, 1, acc) : dot4(AP + lo * b, 1, Aa + hi * b, 1, b, acc);
      Pp[i * b + j] = acc;
    });
  };
  auto kal3 = [&]() {                                        // FP = F Pp
    if (on(3)) yb.each([&](int i, int j) {
      R acc = R(0);
      acc = dot4(Fa + i * b, 1, Pp + j, b, b, acc);
      FP[i * b + j] = acc;
    });
  };
  auto kal4 = [&]() {                                        // Gk = FP F' + W W'
    if (on(3)) yy.each([&](int i, int j) {
      const int lo = i < j ? i : j, hi = i < j ? j : i;
      R acc = WWa[lo * y + hi];
      acc = dot4(FP + lo * b, 1, Fa + hi * b, 1, b, acc);
      Gk[i * y + j] = acc;
    });
  };
  auto kalF = [&]() {                                        // Gk^-1 (registers, compile-time extent) -> LDS
    if (on(3)) dispatch_small<CY>(y, [&](auto ny_) {
      constexpr int NY = decltype(ny_)::value;
      R Gi[NY * NY], unused[NY * NY];
      spd_inverse_reg<R, NY>(Gk, R(0), false, Gi, unused);
      if (ln == 0) {
        LQG_UNROLL for (int e = 0; e < NY * NY; ++e) Gis[e] = Gi[e];
      }
    });
  };
  auto kal5 = [&](int t) {                                   // K = (F Pp)' Gk^-1 ; P = Pp - K F Pp (row of K per thread)
    if (on(3)) bb.each([&](int i, int j) {
      const int lo = i < j ? i : j, hi = i < j ? j : i;
      R acc = Pp[lo * b + hi];
      for (int k = 0; k < y; ++k) acc -= dot4(FP + lo, b, Gis + k, y, y, R(0)) * FP[k * b + hi];   // K[lo, k] (F Pp)[k, hi]
      P[i * b + j] = acc;
    });
    if (on(2)) by.each([&](int i, int j) {
      const R kij = dot4(FP + i, b, Gis + j, y, y, R(0));
      K[i * y + j] = kij;
      if (a.K.p) const_cast<R*>(a.K.p)[s * a.K.sb + (long)t * a.K.st + i * a.K.sr + j * a.K.sc] = kij;
    });
  };
  // ---- joint-system stages
  auto joint1 = [&](int t) {                                 // products that only need K_t, L_t and constants
    const R* Lg = a.Ls + (s * a.T + t) * (long)(u * b);
    if (on(0)) bu.each([&](int i, int j) {                   // BK = Ba + K DB
      R acc = Ba[i * u + j];
      acc = dot4(K + i * y, 1, DB + j, u, y, acc);
      BK[i * u + j] = acc;
    });
    if (on(1)) bx.each([&](int i, int j) {                   // K FAd, K N2
      R a1 = R(0), a2 = R(0);
      a1 = dot4(K + i * y, 1, FAd + j, x, y, a1);
      a2 = dot4(K + i * y, 1, N2 + j, x, y, a2);
      KFAd[i * x + j] = a1;
      KN2[i * x + j] = a2;
    });
    if (on(0)) bb.each([&](int i, int j) {                   // K FAa
      R acc = R(0);
      acc = dot4(K + i * y, 1, FAa + j, b, y, acc);
      KFAa[i * b + j] = acc;
    });
    if (on(2)) xb.each([&](int i, int j) {                   // Bd L
      R acc = R(0);
      acc = dot4(Bd + i * u, 1, Lg + j, b, u, acc);
      BdL[i * b + j] = acc;
    });
    if (on(1)) by.each([&](int i, int j) {                   // K N3
      R acc = R(0);
      acc = dot4(K + i * y, 1, N3 + j, y, y, acc);
      KN3[i * y + j] = acc;
    });
    if (on(2))
      for (int e = ln; e < u * b; e += STRIDE) Lm[e] = Lg[e];
  };
  auto joint2 = [&](bool first) {                            // Fj, GG (system.py:167-207); first: Sigma := GG
    if (on(0)) mm_.each([&](int i, int j) {
      R f;
      if (i < x) {
        f = (j < x) ? Ad[i * x + j] : BdL[i * b + (j - x)];
      } else {
        const int ib = i - x;
        if (j < x) {
          f = KFAd[ib * x + j];
        } else {
          const int jb = j - x;
          R acc = Aa[ib * b + jb] - KFAa[ib * b + jb];
          acc = dot4(BK + ib * u, 1, Lm + jb, b, u, acc);
          f = acc;
        }
      }
      Fj[i * m + j] = f;
    });
    if (on(1)) mm_.each([&](int i, int j) {
      R g;
      const int lo = i < j ? i : j, hi = i < j ? j : i;
      if (hi < x) g = N1[lo * x + hi];
      else if (lo < x) g = KN2[(hi - x) * x + lo];
      else g = dot4(KN3 + (lo - x) * y, 1, K + (hi - x) * y, 1, y, R(0));
      GG[i * m + j] = g;
      if (first) Sg[i * m + j] = g;                          // Sigma0 := G[0] G[0]'  system.py:212
    });
  };
  // ---- conditioning on the observed block of Sg
  auto condF = [&](int t) {                                  // Li = chol(S_oo)^-1, half log-det -> LDS + operator stream
    if (on(2)) dispatch_small<CD>(o, [&](auto no_) {
      constexpr int NO = decltype(no_)::value;
      R Li[NO * NO], hl;
      chol_inverse_reg<R, NO>(Sg, m, Li, hl);
      if (ln == 0) {
        LQG_UNROLL for (int e = 0; e < NO * NO; ++e) Lis[e] = Li[e];
        if (a.ops) {
          R* op = a.ops + (s * (a.T + 1) + t) * (long)a.nops + m * m + rr * o;
          int e = 0;
          LQG_UNROLL for (int i = 0; i < NO; ++i)
            LQG_UNROLL for (int j = 0; j <= i; ++j) op[e++] = Li[i * NO + j];
          op[e] = hl + kLogNorm;
        }
      }
    });
  };
  auto cond1 = [&](int t) {                                  // U2 = S_ro Li'
    if (on(2)) ro.each([&](int p, int j) {
      const R v = dot4(Sg + (o + p) * m, 1, Lis + j * o, 1, j + 1, R(0));
      U2[p * o + j] = v;
      if (a.ops) a.ops[(s * (a.T + 1) + t) * (long)a.nops + m * m + p * o + j] = v;
    });
  };
  auto cond2 = [&]() {                                       // C = S_rr - U2 U2'  (+ the row lists of this step's F2)
    if (sp) build_lists<STRIDE>(ln, Fj + o, m, 1, m, rr, rowF);
    if (on(0)) rrs.each([&](int p, int q2) {
      const int lo = p < q2 ? p : q2, hi = p < q2 ? q2 : p;
      R acc = Sg[(o + lo) * m + o + hi];
      acc = -dot4(U2 + lo * o, 1, U2 + hi * o, 1, o, -acc);
      C[p * rr + q2] = acc;
    });
  };
  auto sig1 = [&](int t) {                                   // T1 = Fj[:, o:] C ; emit Fj - I (diagonal assembled as a deviation)
    if (on(0)) mr.each([&](int i, int q2) {
      R acc = R(0);
      acc = sp ? dot_list(rowF, i, Fj + i * m + o, 1, C + q2, rr, acc) : dot4(Fj + i * m + o, 1, C + q2, rr, rr, acc);
      T1[i * rr + q2] = acc;
    });
    if (a.ops && on(1)) {
      R* op = a.ops + (s * (a.T + 1) + t) * (long)a.nops;
      mm_.each([&](int i, int j) {
        R f = Fj[i * m + j];
        if (i == j) {                                        // (A_ii - 1) first, then the small terms: see k_forward
          if (i < x) {
            f = Ad[i * x + i] - R(1);
          } else {
            const int ib = i - x;
            f = dot4(BK + ib * u, 1, Lm + ib, b, u, (Aa[ib * b + ib] - R(1)) - KFAa[ib * b + ib]);
          }
        }
        op[i * m + j] = f;
      });
    }
  };
  auto sig2 = [&](int t) {                                   // Sigma' = T1 Fj[:, o:]' + GG          system.py:223-230
    if (on(0)) mm_.each([&](int i, int j) {
      const int lo = i < j ? i : j, hi = i < j ? j : i;
      R acc = GG[lo * m + hi];
      acc = sp ? dot_list(rowF, hi, Fj + hi * m + o, 1, T1 + lo * rr, 1, acc) : dot4(T1 + lo * rr, 1, Fj + hi * m + o, 1, rr, acc);
      Sg[i * m + j] = acc;
      if (a.Sig.p) const_cast<R*>(a.Sig.p)[s * a.Sig.sb + (long)t * a.Sig.st + i * a.Sig.sr + j * a.Sig.sc] = acc;
    });
  };

  // ---- prologue: P0, constants, Kalman step 0
  auto kalman_step = [&](int t) {                            // the six Kalman stages, unpipelined
    kal1();
    stage_end<BLOCK>();
    kal2();
    stage_end<BLOCK>();
    kal3();
    stage_end<BLOCK>();
    kal4();
    stage_end<BLOCK>();
    kalF();
    stage_end<BLOCK>();
    kal5(t);
    stage_end<BLOCK>();
  };
  if (on(3)) {
    if (a.Sigma0.p) ld_sym(bb, a.Sigma0, s, 0, P);
    else ld_gram(bb, a.aV, s, 0, a.nva, P);                  // V[0] V[0]'  system.py:79,160
  }
  load_consts(0);
  stage_end<BLOCK>();
  if (sp) build_lists<STRIDE>(ln, Aa, b, 1, b, b, rowA);
  hoist1();
  stage_end<BLOCK>();
  hoist2();
  kalman_step(0);
#ifdef LQG_COOP_STAMP
  unsigned long long stamp_prev_ = __builtin_readcyclecounter();
#endif

  for (int t = 0; t < a.T; ++t) {
    const bool more = t + 1 < a.T;
    if (!a.ti || t == 0) {
      // step 0 (Sigma is only initialised inside it) and time-varying specs (K_{t+1} needs the constants of step t+1
      // while the joint system of step t needs those of step t) run unpipelined: joint stages, then the Kalman step
      if (joint) {
        joint1(t);
        if (t > 0) condF(t);
        stage_end<BLOCK>();
        joint2(t == 0);
        if (t > 0) cond1(t);
        stage_end<BLOCK>();
        if (t == 0) { condF(0); stage_end<BLOCK>(); cond1(0); stage_end<BLOCK>(); }
        cond2();
        stage_end<BLOCK>();
        sig1(t);
        stage_end<BLOCK>();
        sig2(t);
        stage_end<BLOCK>();
      }
      if (more) {
        if (!a.ti) {
          load_consts(t + 1);
          stage_end<BLOCK>();
          if (sp) build_lists<STRIDE>(ln, Aa, b, 1, b, b, rowA);
          hoist1();
          stage_end<BLOCK>();
          hoist2();
        }
        kalman_step(t + 1);
      }
      continue;
    }
    // ---- time-invariant, t > 0: six stages per step, Kalman step t+1 under the joint / Sigma stages of step t
    if (joint) { joint1(t); condF(t); }                      // A
    LQG_STAMP(0);
    if (more) kal1();
    stage_end<BLOCK>();
    LQG_STAMP(1);
    if (joint) { joint2(false); cond1(t); }                  // B   (K_t is dead after this stage)
    LQG_STAMP(2);
    if (more) kal2();
    stage_end<BLOCK>();
    LQG_STAMP(3);
    if (joint) cond2();                                      // C
    LQG_STAMP(4);
    if (more) kal3();
    stage_end<BLOCK>();
    LQG_STAMP(5);
    if (joint) sig1(t);                                      // D
    LQG_STAMP(6);
    if (more) kal4();
    stage_end<BLOCK>();
    LQG_STAMP(7);
    if (joint) sig2(t);                                      // E
    LQG_STAMP(8);
    if (more) kalF();
    stage_end<BLOCK>();
    LQG_STAMP(9);
    if (more) { kal5(t + 1); stage_end<BLOCK>(); }           // F
    LQG_STAMP(10);
  }
  // ---- last row: only the density operators of x_T
  if (joint && a.ops) condF(a.T);
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
      for (int p = 0; p < rr; ++p) AT(muR, p) = AT(c, p) + AT(mn, o + p);      // (the stream holds Fj - I)
      if (a.mu.p && live) {
        R* dst = const_cast<R*>(a.mu.p) + sys * a.mu.sb + n * a.mu.sn + (long)t * a.mu.st;
        for (int i = 0; i < m; ++i) dst[i * a.mu.sd] = (i < o) ? AT(xt, i) + AT(mn, i) : AT(muR, i - o);
      }
    }
  }
#undef AT
  if (a.ll && live) a.ll[sys * a.ll_sb + n * a.ll_sn] = (R)acc;
}

// ======================================================================================================================
// Row-parallel per-trial sweep for LARGE joint dimensions (run-time m, d), same operator stream, same arithmetic per row.
// k_coop_trial walks a trial with ONE thread: m^2 dependent multiply-adds per step behind LDS round trips — at m = 65 (the
// reference's DelayedSubjectiveActor) 210 us per step, 105 ms per evaluation whatever the number of trials (measured,
// profiles/r03_b_bench.json).  Here a group of RT = BLOCK / tpb threads owns a trial and splits the ROWS of the mean update
// (thread r: rows r, r + RT, ...); the step's operator block is staged in LDS once per workgroup (double-buffered: the loads
// of step t + 1 are in flight while step t computes) and shared by the tpb trials of the block; two barriers per step.
// TIME-CHUNKED use of the same kernel (lqg_trial_chunk.hpp has the scheme and its lane-kernel form): with one or a few
// hundred trials the sweep is ONE dependent chain of T steps (3.4 us each at m = 65) on a nearly empty chip.  mode 1: every
// (trial, chunk) runs its chunk from the zero state, and m pseudo-trials per (system, chunk) push the unit vectors through
// the chunk with the data set to zero (the chunk's transition matrix, column by column); k_coop_trial_fix then walks the chunk
// boundaries; mode 2: every (trial, chunk) again from its true start state, now evaluating the densities.  grid.z = chunk.
template <typename R>
struct TrialChunkRT {
  int mode;          // 0: one pass over the whole horizon; 1: zero-state pass; 2: density pass
  int n_chunks, chunk_len;
  R* state;          // [sys][n_chunks-1][m][n_trials]
  R* phi;            // [sys][n_chunks-1][m][m]
  double* part;      // [sys][n_chunks][n_trials]
};

// Row lists of the mean-update block (Fj - I) of a system's operator stream: column k of row i is listed when the entry is
// non-zero at ANY step (the union over the horizon: the walk below stays exact for gains whose entries vanish at some steps).
// The delay augmentations' joint dynamics are shift-structured — DelayedSubjectiveActor (m = 65): 9 % of the 4225 entries —
// and the dense update was the bulk of a candidate batch (4096 candidates x 120 trials: 582 of 784 ms).  Layout per
// system: count[m] | columns[m][m] | flag (1: lists pay, i.e. fewer than half of the entries are listed).
// Two kernels: k_coop_trial_flags (grid: systems x slices of the horizon; flag = 1 where an entry is non-zero at a step of the
// slice, plain stores of the same value into the ZEROED columns area) and k_coop_trial_lists (one workgroup per system: every
// row's flags compacted in place into its column list).
template <typename R, int BLOCK>
__global__ void __launch_bounds__(BLOCK) k_coop_trial_flags(const R* __restrict__ ops_all, unsigned char* lists_all, long lists_stride,
                                                            int T, int m, int nops, int steps_per_slice) {
  const long sys = blockIdx.x;
  const int t_lo = (int)blockIdx.y * steps_per_slice, t_hi = t_lo + steps_per_slice < T ? t_lo + steps_per_slice : T;
  const R* __restrict__ op = ops_all + sys * (long)(T + 1) * nops;
  unsigned char* flags = lists_all + sys * lists_stride + m;
  for (int e = threadIdx.x; e < m * m; e += BLOCK) {
    bool nz = false;
    for (int t = t_lo; t < t_hi; ++t) nz |= op[(long)t * nops + e] != R(0);          // (a NaN entry counts as non-zero)
    if (nz) flags[e] = 1;
  }
}
template <int BLOCK>
__global__ void __launch_bounds__(BLOCK) k_coop_trial_lists(unsigned char* lists_all, long lists_stride, int m) {
  __shared__ int total;
  unsigned char* out = lists_all + (long)blockIdx.x * lists_stride;
  if (threadIdx.x == 0) total = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < m; i += BLOCK) {
    unsigned char* ix = out + m + (long)i * m;                                        // flags in, columns out (n <= k: in place)
    int n = 0;
    for (int k = 0; k < m; ++k)
      if (ix[k]) ix[n++] = (unsigned char)k;
    out[i] = (unsigned char)n;
    atomicAdd(&total, n);
  }
  __syncthreads();
  if (threadIdx.x == 0) out[m + (long)m * m] = (2 * total < m * m) ? 1 : 0;
}

template <typename R, int BLOCK>
__global__ void __launch_bounds__(BLOCK) k_coop_trial_rows(const R* __restrict__ ops_all, const TrialArgsRT<R> a, const int tpb,
                                                           const TrialChunkRT<R> ch, const unsigned char* __restrict__ lists_all,
                                                           const long lists_stride) {
  extern __shared__ double lqg_coop_smem[];
  constexpr int MAXO = 6, MAXPF = 24 * 256 / BLOCK;          // d <= 6 (coop_supported); operator reals per thread per step (nops <= 6144)
  const int m = a.m, o = a.d, rr = m - a.d, tid = threadIdx.x, nops = a.nops;
  const int RT = BLOCK / tpb, g = tid / RT, r = tid - g * RT;
  const long sys = blockIdx.y;
  const int c = blockIdx.z;
  const int t0 = ch.mode ? c * ch.chunk_len : 0;
  const bool lastc = ch.mode == 0 || c == ch.n_chunks - 1;
  const int t_hi = (ch.mode != 1 && lastc) ? a.T + 1 : t0 + ch.chunk_len;     // (exclusive; the last chunk also scores x_T)
  const long ntot = ch.mode == 1 ? a.n_trials + m : a.n_trials;
  long n = (long)blockIdx.x * tpb + g;
  const bool live = n < ntot;
  const bool hom = ch.mode == 1 && n >= a.n_trials;          // unit vector n - n_trials, zero data
  const int unit = (int)(n - a.n_trials);
  n = (live && !hom) ? n : a.n_trials - 1;
  R* sm = reinterpret_cast<R*>(lqg_coop_smem);
  R* opb[2] = {sm, sm + nops};
  R* cv = sm + 2 * nops + (long)g * 2 * m;                   // [x_t ; c] of this group's trial
  R* st = cv + m;                                            // state: dO (o) | muR (rr)
  // row lists of the mean-update block (k_coop_trial_lists), behind the trials' vectors; absent or not worth it: dense rows
  unsigned char* lcnt = reinterpret_cast<unsigned char*>(sm + 2 * nops + (long)tpb * 2 * m);
  unsigned char* lidx = lcnt + m;
  bool listed = false;
  if (lists_all) {
    const unsigned char* __restrict__ ls = lists_all + sys * lists_stride;
    listed = ls[m + (long)m * m] != 0;                       // (workgroup-uniform)
    if (listed)
      for (int e = tid; e < m + m * m; e += BLOCK) lcnt[e] = ls[e];
  }
  const int U_OFF = m * m, L_OFF = U_OFF + rr * o, H_OFF = L_OFF + o * (o + 1) / 2;
  const R* op = ops_all + (sys * (long)(a.T + 1) + t0) * nops;
  const R* xr = a.x.p + sys * a.x.sb + n * a.x.sn;
  {
    const R* s0 = (ch.mode == 2 && c > 0) ? ch.state + (sys * (ch.n_chunks - 1) + (c - 1)) * m * a.n_trials + n : nullptr;
    for (int i = r; i < m; i += RT) st[i] = s0 ? s0[(long)i * a.n_trials] : (hom && i == unit) ? R(1) : R(0);
  }
  // operator blocks in flight.  256 threads: pf holds block t + 1, requested at the top of step t.  1024 threads (DEEP; the largest
  // batches, 5 instead of 18 reals per thread): pf holds block t + 1 requested one step EARLIER and pf2 block t + 2 requested at the
  // top of step t — the step's arithmetic on the listed rows is shorter than one L2 / HBM round trip.  (Measured both ways: two blocks
  // in flight cost the 256-thread sweep its registers — 64 candidates x 120 trials 5.96 -> 7.9 ms — and gain on the wide one:
  // 4096 x 120 fp32 82.8 -> 75.4 ms, 512 x 120 fp64 39.6 -> 23.2 ms.)
  constexpr bool DEEP = BLOCK > 256;
  R pf[MAXPF], pf2[DEEP ? MAXPF : 1];
  const int npf = (nops + BLOCK - 1) / BLOCK;                // <= MAXPF (host-checked)
  LQG_UNROLL for (int k = 0; k < MAXPF; ++k)
    if (k < npf && tid + k * BLOCK < nops) opb[0][tid + k * BLOCK] = op[tid + k * BLOCK];
  if (DEEP && t0 + 1 < t_hi) {
    const R* nx = op + (long)nops;
    LQG_UNROLL for (int k = 0; k < MAXPF; ++k)
      if (k < npf && tid + k * BLOCK < nops) pf[k] = nx[tid + k * BLOCK];
  }
  R xprev[MAXO], xnx[MAXO];
  LQG_UNROLL for (int j = 0; j < MAXO; ++j) {
    xprev[j] = (j < o && !hom) ? xr[(long)(t0 > 0 ? t0 - 1 : 0) * a.x.st + j * a.x.sd] : R(0);
    xnx[j] = (j < o && !hom) ? xr[(long)t0 * a.x.st + j * a.x.sd] : R(0);
  }
  double acc = 0.0;
  __syncthreads();
  for (int t = t0; t < t_hi; ++t) {
    const R* __restrict__ ob = opb[(t - t0) & 1];
    const bool more_ops = t + 1 < t_hi;                      // another step of this pass follows
    const bool more = t < a.T && (more_ops || ch.mode == 1); // the state is advanced (the density pass needs no end state)
    // requests of step t + 1: its operator block (into registers, parked in LDS at the end of the step) and its data row
    if constexpr (DEEP) {
      if (t + 2 < t_hi) {
        const R* nx = op + (long)(t + 2 - t0) * nops;
        LQG_UNROLL for (int k = 0; k < MAXPF; ++k)
          if (k < npf && tid + k * BLOCK < nops) pf2[k] = nx[tid + k * BLOCK];
      }
    } else if (more_ops) {
      const R* nx = op + (long)(t + 1 - t0) * nops;
      LQG_UNROLL for (int k = 0; k < MAXPF; ++k)
        if (k < npf && tid + k * BLOCK < nops) pf[k] = nx[tid + k * BLOCK];
    }
    R xt[MAXO], w[MAXO], e[MAXO];
    LQG_UNROLL for (int j = 0; j < MAXO; ++j) xt[j] = xnx[j];
    {
      const long row = (t + 1 <= a.T) ? (long)(t + 1) : (long)a.T;
      LQG_UNROLL for (int j = 0; j < MAXO; ++j)
        if (j < o && !hom) xnx[j] = xr[row * a.x.st + j * a.x.sd];
    }
    // whitened innovation (every thread of the group, redundantly: d^2 / 2 multiply-adds) and its density
    LQG_UNROLL for (int j = 0; j < MAXO; ++j) e[j] = j < o ? (xt[j] - xprev[j]) - st[j] : R(0);
    R zz = R(0);
    {
      int q = 0;
      LQG_UNROLL for (int i = 0; i < MAXO; ++i) {
        R v = R(0);
        LQG_UNROLL for (int j = 0; j < MAXO; ++j)
          if (j <= i && i < o) v += ob[L_OFF + (q++)] * e[j];
        w[i] = v;
        zz += v * v;
      }
    }
    if (ch.mode != 1 && t > 0 && r == 0) acc -= (double)(R(0.5) * zz + ob[H_OFF]);
    if (more) {
      // c = muR + U2 w (row-parallel), [x_t ; c] into LDS
      for (int p = r; p < rr; p += RT) {
        R v = st[o + p];
        LQG_UNROLL for (int j = 0; j < MAXO; ++j)
          if (j < o) v += ob[U_OFF + p * o + j] * w[j];
        cv[o + p] = v;
      }
      if (r == 0) {
        LQG_UNROLL for (int j = 0; j < MAXO; ++j)
          if (j < o) cv[j] = xt[j];
      }
    }
    __syncthreads();                                         // cv complete; every thread has read st[0:o] of this step
    if (more) {
      // (Fj - I) [x_t ; c], row-parallel; new state = [0 ; c] + that
      for (int i = r; i < m; i += RT) {
        const R* __restrict__ fr = ob + (long)i * m;
        R v0 = R(0), v1 = R(0), v2 = R(0), v3 = R(0);
        if (listed) {
          const unsigned char* __restrict__ ix = lidx + i * m;
          const int nn = lcnt[i];
          int q = 0;
          for (; q + 3 < nn; q += 4) {
            const int k0 = ix[q], k1 = ix[q + 1], k2 = ix[q + 2], k3 = ix[q + 3];
            v0 += fr[k0] * cv[k0];
            v1 += fr[k1] * cv[k1];
            v2 += fr[k2] * cv[k2];
            v3 += fr[k3] * cv[k3];
          }
          for (; q < nn; ++q) { const int k = ix[q]; v0 += fr[k] * cv[k]; }
        } else {
          int j = 0;
          for (; j + 3 < m; j += 4) {
            v0 += fr[j] * cv[j];
            v1 += fr[j + 1] * cv[j + 1];
            v2 += fr[j + 2] * cv[j + 2];
            v3 += fr[j + 3] * cv[j + 3];
          }
          for (; j < m; ++j) v0 += fr[j] * cv[j];
        }
        const R mn = (v0 + v1) + (v2 + v3);
        const R ns = i < o ? mn : cv[i] + mn;
        st[i] = ns;
        if (ch.mode == 0 && a.mu.p && live) {
          R* dst = const_cast<R*>(a.mu.p) + sys * a.mu.sb + n * a.mu.sn + (long)t * a.mu.st;
          dst[i * a.mu.sd] = i < o ? cv[i] + mn : ns;            // (cv[0:o] holds x_t)
        }
      }
      LQG_UNROLL for (int j = 0; j < MAXO; ++j) xprev[j] = xt[j];
    }
    if (more_ops) {
      R* nb = opb[(t + 1 - t0) & 1];
      LQG_UNROLL for (int k = 0; k < MAXPF; ++k)
        if (k < npf && tid + k * BLOCK < nops) nb[tid + k * BLOCK] = pf[k];
      if constexpr (DEEP) { LQG_UNROLL for (int k = 0; k < MAXPF; ++k) pf[k] = pf2[k]; }
    }
    __syncthreads();                                         // new state + next operator block visible
  }
  if (!live) return;
  if (ch.mode == 1) {
    const long slot = sys * (ch.n_chunks - 1) + c;
    if (hom) {
      R* ph = ch.phi + slot * ((long)m * m) + unit;
      for (int i = r; i < m; i += RT) ph[(long)i * m] = st[i];
    } else {
      R* sp = ch.state + slot * m * a.n_trials + n;
      for (int i = r; i < m; i += RT) sp[(long)i * a.n_trials] = st[i];
    }
  } else if (r == 0) {
    if (ch.mode == 2) ch.part[(sys * ch.n_chunks + c) * a.n_trials + n] = acc;
    else if (a.ll) a.ll[sys * a.ll_sb + n * a.ll_sn] = (R)acc;
  }
}

// start state of every chunk: s_{c+1} = Phi_c s_c + z_c along the chunk boundaries of one trial (grid: (trials, systems); the
// rows of the m x m product on the lanes).  Slot c of `state` holds the zero-state end of chunk c on entry and the start state
// of chunk c + 1 on return (slot 0 is both: chunk 0 starts from zero) — the convention of k_trial_fix.
template <typename R>
__global__ void __launch_bounds__(128) k_coop_trial_fix(const R* __restrict__ phi_all, R* state, long n_trials, int n_slots, int m) {
  extern __shared__ double lqg_coop_smem[];
  R* sb[2] = {reinterpret_cast<R*>(lqg_coop_smem), reinterpret_cast<R*>(lqg_coop_smem) + m};
  const long sys = blockIdx.y, n = blockIdx.x;
  R* sp = state + sys * n_slots * m * n_trials + n;
  const R* __restrict__ ph = phi_all + sys * n_slots * ((long)m * m);
  for (int i = threadIdx.x; i < m; i += blockDim.x) sb[0][i] = sp[(long)i * n_trials];
  __syncthreads();
  for (int c = 1; c < n_slots; ++c) {
    sp += (long)m * n_trials;
    ph += (long)m * m;
    const R* __restrict__ s = sb[(c - 1) & 1];
    R* sn = sb[c & 1];
    for (int i = threadIdx.x; i < m; i += blockDim.x) {
      const R* __restrict__ fr = ph + (long)i * m;
      R v0 = sp[(long)i * n_trials], v1 = R(0), v2 = R(0), v3 = R(0);
      int j = 0;
      for (; j + 3 < m; j += 4) {
        v0 += fr[j] * s[j];
        v1 += fr[j + 1] * s[j + 1];
        v2 += fr[j + 2] * s[j + 2];
        v3 += fr[j + 3] * s[j + 3];
      }
      for (; j < m; ++j) v0 += fr[j] * s[j];
      const R v = (v0 + v1) + (v2 + v3);
      sn[i] = v;
      sp[(long)i * n_trials] = v;
    }
    __syncthreads();
  }
}

// ======================================================================================================================
// Run-time-dims twin of k_simulate (System.simulate's per-trial scan, lqg/system.py:106-128): one (system, trial) per
// thread, the two state vectors and the per-step temporaries in LDS columns [element][thread]; gains and spec matrices
// are read straight from global memory (every thread of a system reads the same address: broadcast).  For shapes
// without an instantiated k_simulate<x,b,u,y> (e.g. the delay-12 model, x=26 b=39).
template <typename R, int BLOCK, bool RNG = false>
__global__ void __launch_bounds__(BLOCK) k_coop_simulate(const SimArgs<R> a, const int nx, const int nb, const int nu,
                                                         const int ny) {
  extern __shared__ double lqg_coop_smem[];
  R* sm = reinterpret_cast<R*>(lqg_coop_smem);
  const int tid = threadIdx.x;
  long gid = blockIdx.x * (long)BLOCK + tid;
  const bool live = gid < a.n_sys * a.n_trials;
  gid = live ? gid : a.n_sys * a.n_trials - 1;
  const long s = gid / a.n_trials, n = gid % a.n_trials;
  R *x = sm, *xn = x + nx * BLOCK, *xh = xn + nx * BLOCK, *xp = xh + nb * BLOCK, *uu = xp + nb * BLOCK,
    *yy = uu + nu * BLOCK;
#define AT(arr, i) arr[(i) * BLOCK + tid]
  for (int i = 0; i < nx; ++i) AT(x, i) = a.x0.p ? a.x0.p[s * a.x0.sb + i * a.x0.sr] : R(0);
  for (int i = 0; i < nb; ++i) AT(xh, i) = a.xh0.p ? a.xh0.p[s * a.xh0.sb + i * a.xh0.sr] : R(0);
  auto out = [&](const DTraj<R>& v, int t) { return const_cast<R*>(v.p) + s * v.sb + n * v.sn + (long)t * v.st; };
  if (live) {
    R* d0 = out(a.xs, 0);
    for (int i = 0; i < nx; ++i) d0[i * a.xs.sd] = AT(x, i);
    if (a.xh.p) { R* d1 = out(a.xh, 0); for (int i = 0; i < nb; ++i) d1[i * a.xh.sd] = AT(xh, i); }
  }
  auto M = [&](const DView<R>& v, int t, int i, int j) { return v.p[s * v.sb + (long)t * v.st + i * v.sr + j * v.sc]; };
  for (int t = 0; t < a.T; ++t) {
    for (int i = 0; i < nu; ++i) {                           // u = L xhat + l     system.py:110
      R v = a.l.p ? a.l.p[s * a.l.sb + (long)t * a.l.st + i * a.l.sr] : R(0);
      for (int k = 0; k < nb; ++k) v += M(a.L, t, i, k) * AT(xh, k);
      AT(uu, i) = v;
    }
    const R* ep = RNG ? nullptr : a.eps.p + s * a.eps.sb + n * a.eps.sn + (long)t * a.eps.st;
    const R* et = RNG ? nullptr : a.eta.p + s * a.eta.sb + n * a.eta.sn + (long)t * a.eta.st;
    for (int i = 0; i < nx; ++i) {                           // x = A x + B u + V eps   system.py:113-117
      R v = R(0);
      for (int k = 0; k < nx; ++k) v += M(a.dA, t, i, k) * AT(x, k);
      for (int k = 0; k < nu; ++k) v += M(a.dB, t, i, k) * AT(uu, k);
      AT(xn, i) = v;
    }
    if constexpr (RNG) {                                     // the same draws as k_simulate<RNG> (lqg_rng.hpp)
      for (int k0 = 0; k0 < a.nvd; k0 += 4) {
        float z[4];
        rng::normal4(a.seed, s, n, (uint32_t)t, (uint32_t)(k0 >> 2), z);
        LQG_UNROLL for (int j = 0; j < 4; ++j)
          if (k0 + j < a.nvd)
            for (int i = 0; i < nx; ++i) AT(xn, i) += M(a.dV, t, i, k0 + j) * (R)z[j];
      }
    } else {
      for (int k = 0; k < a.nvd; ++k) {
        const R e = ep[k * a.eps.sd];
        for (int i = 0; i < nx; ++i) AT(xn, i) += M(a.dV, t, i, k) * e;
      }
    }
    for (int i = 0; i < nx; ++i) AT(x, i) = AT(xn, i);
    for (int i = 0; i < ny; ++i) {                           // y = F x + W eta          system.py:120
      R v = R(0);
      for (int k = 0; k < nx; ++k) v += M(a.dF, t, i, k) * AT(x, k);
      AT(yy, i) = v;
    }
    if constexpr (RNG) {
      for (int k0 = 0; k0 < a.nwd; k0 += 4) {
        float z[4];
        rng::normal4(a.seed, s, n, (uint32_t)t, rng::kEtaBlock + (uint32_t)(k0 >> 2), z);
        LQG_UNROLL for (int j = 0; j < 4; ++j)
          if (k0 + j < a.nwd)
            for (int i = 0; i < ny; ++i) AT(yy, i) += M(a.dW, t, i, k0 + j) * (R)z[j];
      }
    } else {
      for (int k = 0; k < a.nwd; ++k) {
        const R e = et[k * a.eta.sd];
        for (int i = 0; i < ny; ++i) AT(yy, i) += M(a.dW, t, i, k) * e;
      }
    }
    for (int i = 0; i < nb; ++i) {                           // x_pred = A xhat + B u     system.py:123
      R v = R(0);
      for (int k = 0; k < nb; ++k) v += M(a.aA, t, i, k) * AT(xh, k);
      for (int k = 0; k < nu; ++k) v += M(a.aB, t, i, k) * AT(uu, k);
      AT(xp, i) = v;
    }
    if (live && a.ys.p) { R* d2 = out(a.ys, t); for (int i = 0; i < ny; ++i) d2[i * a.ys.sd] = AT(yy, i); }
    for (int i = 0; i < ny; ++i) {                           // innovation y - F x_pred (kept in yy)
      R v = AT(yy, i);
      for (int k = 0; k < nb; ++k) v -= M(a.aF, t, i, k) * AT(xp, k);
      AT(yy, i) = v;
    }
    for (int i = 0; i < nb; ++i) {                           // xhat = x_pred + K (y - F x_pred)  system.py:124
      R v = AT(xp, i);
      for (int k = 0; k < ny; ++k) v += M(a.K, t, i, k) * AT(yy, k);
      AT(xh, i) = v;
    }
    if (live) {
      R* d0 = out(a.xs, t + 1);
      for (int i = 0; i < nx; ++i) d0[i * a.xs.sd] = AT(x, i);
      if (a.xh.p) { R* d1 = out(a.xh, t + 1); for (int i = 0; i < nb; ++i) d1[i * a.xh.sd] = AT(xh, i); }
      if (a.us.p) { R* d3 = out(a.us, t); for (int i = 0; i < nu; ++i) d3[i * a.us.sd] = AT(uu, i); }
    }
  }
#undef AT
}

}  // namespace coop
}  // namespace lqg
