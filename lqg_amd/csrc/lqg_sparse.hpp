// lqg_sparse.hpp — register-resident small matrices whose STRUCTURAL ZEROS are part of the type (C++20 class-type
// template parameters).  `Mat<R, M, N, MK>` stores M*N reals of which only the entries with MK(i, j) == true are
// ever written or read; every product / sum computes its result mask at compile time (boolean matrix algebra in
// constexpr functions) and skips the terms that are structurally zero.  With all-true masks the code is exactly
// the dense code of lqg_small.hpp; with the masks of a real model (A = I + a few couplings, F = selection rows,
// V, W diagonal, ...) most of the multiply-adds of a step disappear at compile time and so do the registers that
// held the zeros.  Skipping an exactly-zero term is numerically exact (0 * x + acc == acc for finite x).
#pragma once
#include "lqg_small.hpp"

namespace lqg {

template <int M, int N>
struct Mask {
  bool b[M * N];
  constexpr bool operator()(int i, int j) const { return b[i * N + j]; }
  constexpr int count() const {
    int c = 0;
    for (int i = 0; i < M * N; ++i) c += b[i] ? 1 : 0;
    return c;
  }
};

template <int M, int N>
constexpr Mask<M, N> mask_full() {
  Mask<M, N> r{};
  for (int i = 0; i < M * N; ++i) r.b[i] = true;
  return r;
}
template <int N>
constexpr Mask<N, N> mask_eye() {
  Mask<N, N> r{};
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) r.b[i * N + j] = i == j;
  return r;
}
template <int M, int N>
constexpr Mask<M, N> mask_none() {
  Mask<M, N> r{};
  for (int i = 0; i < M * N; ++i) r.b[i] = false;
  return r;
}
template <int M, int K, int N>
constexpr Mask<M, N> mask_mul(const Mask<M, K>& a, const Mask<K, N>& b) {
  Mask<M, N> r{};
  for (int i = 0; i < M; ++i)
    for (int j = 0; j < N; ++j) {
      bool v = false;
      for (int k = 0; k < K; ++k) v = v || (a(i, k) && b(k, j));
      r.b[i * N + j] = v;
    }
  return r;
}
template <int M, int N>
constexpr Mask<N, M> mask_t(const Mask<M, N>& a) {
  Mask<N, M> r{};
  for (int i = 0; i < M; ++i)
    for (int j = 0; j < N; ++j) r.b[j * M + i] = a(i, j);
  return r;
}
template <int M, int N>
constexpr Mask<M, N> mask_or(const Mask<M, N>& a, const Mask<M, N>& b) {
  Mask<M, N> r{};
  for (int i = 0; i < M * N; ++i) r.b[i] = a.b[i] || b.b[i];
  return r;
}
template <int M, int N>
constexpr Mask<M, N> mask_and(const Mask<M, N>& a, const Mask<M, N>& b) {
  Mask<M, N> r{};
  for (int i = 0; i < M * N; ++i) r.b[i] = a.b[i] && b.b[i];
  return r;
}
// [[a, b], [c, d]] block matrix
template <int M1, int M2, int N1, int N2>
constexpr Mask<M1 + M2, N1 + N2> mask_block(const Mask<M1, N1>& a, const Mask<M1, N2>& b, const Mask<M2, N1>& c,
                                            const Mask<M2, N2>& d) {
  Mask<M1 + M2, N1 + N2> r{};
  constexpr int N = N1 + N2;
  for (int i = 0; i < M1; ++i) {
    for (int j = 0; j < N1; ++j) r.b[i * N + j] = a(i, j);
    for (int j = 0; j < N2; ++j) r.b[i * N + N1 + j] = b(i, j);
  }
  for (int i = 0; i < M2; ++i) {
    for (int j = 0; j < N1; ++j) r.b[(M1 + i) * N + j] = c(i, j);
    for (int j = 0; j < N2; ++j) r.b[(M1 + i) * N + N1 + j] = d(i, j);
  }
  return r;
}
// columns [C0, C0 + NC) of a mask
template <int C0, int NC, int M, int N>
constexpr Mask<M, NC> mask_cols(const Mask<M, N>& a) {
  Mask<M, NC> r{};
  for (int i = 0; i < M; ++i)
    for (int j = 0; j < NC; ++j) r.b[i * NC + j] = a(i, C0 + j);
  return r;
}

template <int N>
constexpr bool mask_is_diag(const Mask<N, N>& a) {
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j)
      if (i != j && a(i, j)) return false;
  return true;
}
template <int M, int N>
constexpr bool mask_eq(const Mask<M, N>& a, const Mask<M, N>& b) {
  for (int i = 0; i < M * N; ++i)
    if (a.b[i] != b.b[i]) return false;
  return true;
}

template <typename R, int M, int N, Mask<M, N> MK = mask_full<M, N>()>
struct Mat {
  R v[M * N];
  static constexpr Mask<M, N> mask = MK;
  static constexpr int rows = M, cols = N;
  LQG_DEV R at(int i, int j) const { return MK(i, j) ? v[i * N + j] : R(0); }   // indices are unrolled constants
};

// ---- loads / stores -----------------------------------------------------------------------------------------
template <typename R, int M, int N, Mask<M, N> MK>
LQG_DEV Mat<R, M, N, MK> load_masked(const R* __restrict__ p, long sr, long sc) {
  Mat<R, M, N, MK> r;
  LQG_UNROLL for (int i = 0; i < M; ++i)
    LQG_UNROLL for (int j = 0; j < N; ++j)
      if (MK(i, j)) r.v[i * N + j] = p[i * sr + j * sc];
  return r;
}
// The same loads with the address split into a WAVE-UNIFORM entry pointer (p + i sr + j sc: scalar registers, scalar arithmetic)
// and the lane's own element offset `lo` (one 64-bit add per load).  The time-varying sweeps (lqg_kernels_sp.hpp) re-form every
// address every step: as one per-lane pointer expression that was 289 integer VALU instructions + 201 v_readlane (the strides no
// longer fit the scalar registers) against 269 floating-point ones per step of k_forward_tv_sp<double> (round 6).
template <typename R, int M, int N, Mask<M, N> MK>
LQG_DEV Mat<R, M, N, MK> load_masked_u(const R* __restrict__ p, long sr, long sc, long lo) {
  Mat<R, M, N, MK> r;
  LQG_UNROLL for (int i = 0; i < M; ++i)
    LQG_UNROLL for (int j = 0; j < N; ++j)
      if (MK(i, j)) { const R* __restrict__ q = p + (i * sr + j * sc); r.v[i * N + j] = q[lo]; }
  return r;
}
// CANONICAL storage [T][row][col][system] (workload.pack_systems: sb = 1, sc = ld, sr = cols ld, st = rows cols ld with ONE ld for every
// field): an entry is the field's step pointer (scalar registers) + a 32-bit scalar offset (entry index x ld bytes) + the lane's 32-bit
// byte offset — the form the hardware addresses in one instruction (global_load v, v_lane, s[base:base+1]): no per-load vector
// arithmetic, and of the strides only `ldb` stays live.  `pt`: the field at this step (wave-uniform); ldb = ld sizeof(R); lob = lane
// sizeof(R).  The caller guarantees rows cols ldb < 2^31 and n_sys sizeof(R) < 2^32 (host check: lqg_sp_entry.hpp canon_layout).
template <typename R>
LQG_DEV R canon_at(const R* __restrict__ pt, unsigned entry, unsigned ldb, unsigned lob) {
  const char* q = reinterpret_cast<const char*>(pt) + entry * ldb;      // wave-uniform
  return *reinterpret_cast<const R*>(q + lob);
}
template <typename R, int M, int N, Mask<M, N> MK>
LQG_DEV Mat<R, M, N, MK> load_masked_c(const R* __restrict__ pt, unsigned ldb, unsigned lob) {
  Mat<R, M, N, MK> r;
  LQG_UNROLL for (int i = 0; i < M; ++i)
    LQG_UNROLL for (int j = 0; j < N; ++j)
      if (MK(i, j)) r.v[i * N + j] = canon_at<R>(pt, (unsigned)(i * N + j), ldb, lob);
  return r;
}
template <typename R, int N, Mask<N, N> MK>
LQG_DEV Mat<R, N, N, MK> load_sym_masked_c(const R* __restrict__ pt, unsigned ldb, unsigned lob) {
  Mat<R, N, N, MK> r;
  LQG_UNROLL for (int i = 0; i < N; ++i)
    LQG_UNROLL for (int j = i; j < N; ++j)
      if (MK(i, j)) {
        R v = (i == j) ? canon_at<R>(pt, (unsigned)(i * N + i), ldb, lob)
                       : R(0.5) * (canon_at<R>(pt, (unsigned)(i * N + j), ldb, lob) + canon_at<R>(pt, (unsigned)(j * N + i), ldb, lob));
        r.v[i * N + j] = v;
        r.v[j * N + i] = v;
      }
  return r;
}
// V V' of a stored V[N, nv] (nv columns in storage: run time), the entries of MV only (cf. load_gram_masked_raw)
template <typename R, int N, Mask<N, N> MK, Mask<N, N> MV>
LQG_DEV Mat<R, N, N, MK> load_gram_masked_raw_c(const R* __restrict__ pt, int nv, unsigned ldb, unsigned lob) {
  Mat<R, N, N, MK> r;
  LQG_UNROLL for (int i = 0; i < N * N; ++i) r.v[i] = R(0);
  LQG_UNROLL for (int k = 0; k < N; ++k) {
    if (k < nv) {
      R col[N];
      LQG_UNROLL for (int i = 0; i < N; ++i) col[i] = MV(i, k) ? canon_at<R>(pt, (unsigned)(i * nv + k), ldb, lob) : R(0);
      LQG_UNROLL for (int i = 0; i < N; ++i)
        LQG_UNROLL for (int j = i; j < N; ++j)
          if (MK(i, j) && MV(i, k) && MV(j, k)) r.v[i * N + j] += col[i] * col[j];
    }
  }
  for (int k = N; k < nv; ++k) {
    R col[N];
    LQG_UNROLL for (int i = 0; i < N; ++i) col[i] = canon_at<R>(pt, (unsigned)(i * nv + k), ldb, lob);
    LQG_UNROLL for (int i = 0; i < N; ++i)
      LQG_UNROLL for (int j = i; j < N; ++j)
        if (MK(i, j)) r.v[i * N + j] += col[i] * col[j];
  }
  LQG_UNROLL for (int i = 0; i < N; ++i)
    LQG_UNROLL for (int j = 0; j < i; ++j) r.v[i * N + j] = r.v[j * N + i];
  return r;
}
template <typename R, int N, Mask<N, N> MK>
LQG_DEV Mat<R, N, N, MK> load_sym_masked_u(const R* __restrict__ p, long sr, long sc, long lo) {
  Mat<R, N, N, MK> r;
  LQG_UNROLL for (int i = 0; i < N; ++i)
    LQG_UNROLL for (int j = i; j < N; ++j)
      if (MK(i, j)) {
        const R* __restrict__ q0 = p + (i * sr + j * sc);
        const R* __restrict__ q1 = p + (j * sr + i * sc);
        R v = (i == j) ? q0[lo] : R(0.5) * (q0[lo] + q1[lo]);
        r.v[i * N + j] = v;
        r.v[j * N + i] = v;
      }
  return r;
}
// symmetric part of a stored square matrix; MK must be symmetric
template <typename R, int N, Mask<N, N> MK>
LQG_DEV Mat<R, N, N, MK> load_sym_masked(const R* __restrict__ p, long sr, long sc) {
  Mat<R, N, N, MK> r;
  LQG_UNROLL for (int i = 0; i < N; ++i)
    LQG_UNROLL for (int j = i; j < N; ++j)
      if (MK(i, j)) {
        R v = (i == j) ? p[i * sr + i * sc] : R(0.5) * (p[i * sr + j * sc] + p[j * sr + i * sc]);
        r.v[i * N + j] = v;
        r.v[j * N + i] = v;
      }
  return r;
}
// V V^T for a stored V[N, nv] (run-time nv); only the entries of the symmetric mask MK are formed
template <typename R, int N, Mask<N, N> MK>
LQG_DEV Mat<R, N, N, MK> load_gram_masked(const R* __restrict__ p, long sr, long sc, int nv) {
  Mat<R, N, N, MK> r;
  LQG_UNROLL for (int i = 0; i < N * N; ++i) r.v[i] = R(0);
  for (int k = 0; k < nv; ++k) {
    R col[N];
    LQG_UNROLL for (int i = 0; i < N; ++i) col[i] = p[i * sr + k * sc];
    LQG_UNROLL for (int i = 0; i < N; ++i)
      LQG_UNROLL for (int j = i; j < N; ++j)
        if (MK(i, j)) r.v[i * N + j] += col[i] * col[j];
  }
  LQG_UNROLL for (int i = 0; i < N; ++i)
    LQG_UNROLL for (int j = 0; j < i; ++j) r.v[i * N + j] = r.v[j * N + i];
  return r;
}
// the same with the structural zeros of the stored factor itself: MV(i, k) for its first N columns (further columns, if
// nv > N, are loaded in full).  The time-varying sweeps load the factor every step, and a zoo model's noise factors are
// (block-)diagonal: 84 of the 192 memory instructions of a step of k_forward_tv_sp were zeros of V and W.
template <typename R, int N, Mask<N, N> MK, Mask<N, N> MV>
LQG_DEV Mat<R, N, N, MK> load_gram_masked_raw(const R* __restrict__ p, long sr, long sc, int nv, long lo = 0) {
  Mat<R, N, N, MK> r;
  LQG_UNROLL for (int i = 0; i < N * N; ++i) r.v[i] = R(0);
  LQG_UNROLL for (int k = 0; k < N; ++k) {
    if (k < nv) {
      R col[N];
      LQG_UNROLL for (int i = 0; i < N; ++i) {
        const R* __restrict__ q = p + (i * sr + k * sc);          // (wave-uniform entry pointer + the lane's offset: see load_masked_u)
        col[i] = MV(i, k) ? q[lo] : R(0);
      }
      LQG_UNROLL for (int i = 0; i < N; ++i)
        LQG_UNROLL for (int j = i; j < N; ++j)
          if (MK(i, j) && MV(i, k) && MV(j, k)) r.v[i * N + j] += col[i] * col[j];
    }
  }
  for (int k = N; k < nv; ++k) {
    R col[N];
    LQG_UNROLL for (int i = 0; i < N; ++i) col[i] = p[i * sr + k * sc + lo];
    LQG_UNROLL for (int i = 0; i < N; ++i)
      LQG_UNROLL for (int j = i; j < N; ++j)
        if (MK(i, j)) r.v[i * N + j] += col[i] * col[j];
  }
  LQG_UNROLL for (int i = 0; i < N; ++i)
    LQG_UNROLL for (int j = 0; j < i; ++j) r.v[i * N + j] = r.v[j * N + i];
  return r;
}
// dense image (structural zeros written out) — for loop-carried state and for the dense Cholesky kernels
template <typename R, int M, int N, Mask<M, N> MK>
LQG_DEV void to_dense(const Mat<R, M, N, MK>& a, R (&out)[M * N]) {
  LQG_UNROLL for (int i = 0; i < M; ++i)
    LQG_UNROLL for (int j = 0; j < N; ++j) out[i * N + j] = MK(i, j) ? a.v[i * N + j] : R(0);
}
template <typename R, int M, int N>
LQG_DEV Mat<R, M, N> from_dense(const R (&in)[M * N]) {
  Mat<R, M, N> r;
  LQG_UNROLL for (int i = 0; i < M * N; ++i) r.v[i] = in[i];
  return r;
}
// narrow to a mask that is known (numerically, by the host) to be a superset of the non-zeros
template <auto MK2, typename R, int M, int N, Mask<M, N> MK>
LQG_DEV Mat<R, M, N, mask_and(MK, MK2)> restrict_to(const Mat<R, M, N, MK>& a) {
  constexpr auto MR = mask_and(MK, MK2);
  Mat<R, M, N, MR> r;
  LQG_UNROLL for (int i = 0; i < M * N; ++i)
    if (MR.b[i]) r.v[i] = a.v[i];
  return r;
}

// ---- products (result mask computed at compile time; structurally-zero terms are never issued) ---------------
// C = A B
template <typename R, int M, int K, int N, Mask<M, K> MA, Mask<K, N> MB>
LQG_DEV auto mul(const Mat<R, M, K, MA>& a, const Mat<R, K, N, MB>& b) {
  constexpr auto MC = mask_mul(MA, MB);
  Mat<R, M, N, MC> c;
  LQG_UNROLL for (int i = 0; i < M; ++i)
    LQG_UNROLL for (int j = 0; j < N; ++j)
      if (MC(i, j)) {
        R acc = R(0);
        bool first = true;
        LQG_UNROLL for (int k = 0; k < K; ++k)
          if (MA(i, k) && MB(k, j)) {
            acc = first ? a.v[i * K + k] * b.v[k * N + j] : acc + a.v[i * K + k] * b.v[k * N + j];
            first = false;
          }
        c.v[i * N + j] = acc;
      }
  return c;
}
// C = A^T B   (A is K x M)
template <typename R, int M, int K, int N, Mask<K, M> MA, Mask<K, N> MB>
LQG_DEV auto mul_tn(const Mat<R, K, M, MA>& a, const Mat<R, K, N, MB>& b) {
  constexpr auto MC = mask_mul(mask_t(MA), MB);
  Mat<R, M, N, MC> c;
  LQG_UNROLL for (int i = 0; i < M; ++i)
    LQG_UNROLL for (int j = 0; j < N; ++j)
      if (MC(i, j)) {
        R acc = R(0);
        bool first = true;
        LQG_UNROLL for (int k = 0; k < K; ++k)
          if (MA(k, i) && MB(k, j)) {
            acc = first ? a.v[k * M + i] * b.v[k * N + j] : acc + a.v[k * M + i] * b.v[k * N + j];
            first = false;
          }
        c.v[i * N + j] = acc;
      }
  return c;
}
// C = A B^T   (B is N x K)
template <typename R, int M, int K, int N, Mask<M, K> MA, Mask<N, K> MB>
LQG_DEV auto mul_nt(const Mat<R, M, K, MA>& a, const Mat<R, N, K, MB>& b) {
  constexpr auto MC = mask_mul(MA, mask_t(MB));
  Mat<R, M, N, MC> c;
  LQG_UNROLL for (int i = 0; i < M; ++i)
    LQG_UNROLL for (int j = 0; j < N; ++j)
      if (MC(i, j)) {
        R acc = R(0);
        bool first = true;
        LQG_UNROLL for (int k = 0; k < K; ++k)
          if (MA(i, k) && MB(j, k)) {
            acc = first ? a.v[i * K + k] * b.v[j * K + k] : acc + a.v[i * K + k] * b.v[j * K + k];
            first = false;
          }
        c.v[i * N + j] = acc;
      }
  return c;
}
// symmetric C = A B^T + D where the caller knows the result is symmetric (e.g. B = A S with S symmetric):
// only the upper triangle is computed, then mirrored.  D may be sparse (symmetric mask).
template <typename R, int N, int K, Mask<N, K> MA, Mask<N, K> MB, Mask<N, N> MD>
LQG_DEV auto mul_nt_sym_add(const Mat<R, N, K, MA>& a, const Mat<R, N, K, MB>& b, const Mat<R, N, N, MD>& d) {
  constexpr auto MP = mask_mul(MA, mask_t(MB));
  constexpr auto MC = mask_or(mask_or(MP, mask_t(MP)), MD);
  Mat<R, N, N, MC> c;
  LQG_UNROLL for (int i = 0; i < N; ++i)
    LQG_UNROLL for (int j = i; j < N; ++j)
      if (MC(i, j)) {
        R acc = MD(i, j) ? d.v[i * N + j] : R(0);
        LQG_UNROLL for (int k = 0; k < K; ++k)
          if (MA(i, k) && MB(j, k)) acc += a.v[i * K + k] * b.v[j * K + k];
        c.v[i * N + j] = acc;
        c.v[j * N + i] = acc;
      }
  return c;
}
// symmetric C = A^T B + D (A, B are K x N), upper triangle computed and mirrored
template <typename R, int N, int K, Mask<K, N> MA, Mask<K, N> MB, Mask<N, N> MD>
LQG_DEV auto mul_tn_sym_add(const Mat<R, K, N, MA>& a, const Mat<R, K, N, MB>& b, const Mat<R, N, N, MD>& d) {
  constexpr auto MP = mask_mul(mask_t(MA), MB);
  constexpr auto MC = mask_or(mask_or(MP, mask_t(MP)), MD);
  Mat<R, N, N, MC> c;
  LQG_UNROLL for (int i = 0; i < N; ++i)
    LQG_UNROLL for (int j = i; j < N; ++j)
      if (MC(i, j)) {
        R acc = MD(i, j) ? d.v[i * N + j] : R(0);
        LQG_UNROLL for (int k = 0; k < K; ++k)
          if (MA(k, i) && MB(k, j)) acc += a.v[k * N + i] * b.v[k * N + j];
        c.v[i * N + j] = acc;
        c.v[j * N + i] = acc;
      }
  return c;
}
// symmetric C = D - A B (A is N x K, B is K x N), upper triangle computed and mirrored
template <typename R, int N, int K, Mask<N, K> MA, Mask<K, N> MB, Mask<N, N> MD>
LQG_DEV auto sym_sub_mul(const Mat<R, N, N, MD>& d, const Mat<R, N, K, MA>& a, const Mat<R, K, N, MB>& b) {
  constexpr auto MP = mask_mul(MA, MB);
  constexpr auto MC = mask_or(mask_or(MP, mask_t(MP)), MD);
  Mat<R, N, N, MC> c;
  LQG_UNROLL for (int i = 0; i < N; ++i)
    LQG_UNROLL for (int j = i; j < N; ++j)
      if (MC(i, j)) {
        R acc = MD(i, j) ? d.v[i * N + j] : R(0);
        LQG_UNROLL for (int k = 0; k < K; ++k)
          if (MA(i, k) && MB(k, j)) acc -= a.v[i * K + k] * b.v[k * N + j];
        c.v[i * N + j] = acc;
        c.v[j * N + i] = acc;
      }
  return c;
}
template <typename R, int M, int N, Mask<M, N> MK>
LQG_DEV auto transpose(const Mat<R, M, N, MK>& a) {
  constexpr auto MR = mask_t(MK);
  Mat<R, N, M, MR> r;
  LQG_UNROLL for (int i = 0; i < M; ++i)
    LQG_UNROLL for (int j = 0; j < N; ++j) if (MK(i, j)) r.v[j * M + i] = a.v[i * N + j];
  return r;
}
// elementwise: A + s * B with s = +1 / -1 folded at compile time
template <int SIGN, typename R, int M, int N, Mask<M, N> MA, Mask<M, N> MB>
LQG_DEV auto axpy(const Mat<R, M, N, MA>& a, const Mat<R, M, N, MB>& b) {
  constexpr auto MC = mask_or(MA, MB);
  Mat<R, M, N, MC> c;
  LQG_UNROLL for (int i = 0; i < M * N; ++i)
    if (MC.b[i]) {
      R x = MA.b[i] ? a.v[i] : R(0);
      R y = MB.b[i] ? b.v[i] : R(0);
      c.v[i] = SIGN > 0 ? x + y : x - y;
    }
  return c;
}
template <typename R, int M, int N, Mask<M, N> MA, Mask<M, N> MB>
LQG_DEV auto add(const Mat<R, M, N, MA>& a, const Mat<R, M, N, MB>& b) { return axpy<+1>(a, b); }
template <typename R, int M, int N, Mask<M, N> MA, Mask<M, N> MB>
LQG_DEV auto sub(const Mat<R, M, N, MA>& a, const Mat<R, M, N, MB>& b) { return axpy<-1>(a, b); }

// [[a, b], [c, d]]
template <typename R, int M1, int M2, int N1, int N2, Mask<M1, N1> MA, Mask<M1, N2> MB, Mask<M2, N1> MC_, Mask<M2, N2> MD>
LQG_DEV auto block2x2(const Mat<R, M1, N1, MA>& a, const Mat<R, M1, N2, MB>& b, const Mat<R, M2, N1, MC_>& c,
                      const Mat<R, M2, N2, MD>& d) {
  constexpr auto MR = mask_block(MA, MB, MC_, MD);
  constexpr int N = N1 + N2;
  Mat<R, M1 + M2, N, MR> r;
  LQG_UNROLL for (int i = 0; i < M1; ++i) {
    LQG_UNROLL for (int j = 0; j < N1; ++j) if (MA(i, j)) r.v[i * N + j] = a.v[i * N1 + j];
    LQG_UNROLL for (int j = 0; j < N2; ++j) if (MB(i, j)) r.v[i * N + N1 + j] = b.v[i * N2 + j];
  }
  LQG_UNROLL for (int i = 0; i < M2; ++i) {
    LQG_UNROLL for (int j = 0; j < N1; ++j) if (MC_(i, j)) r.v[(M1 + i) * N + j] = c.v[i * N1 + j];
    LQG_UNROLL for (int j = 0; j < N2; ++j) if (MD(i, j)) r.v[(M1 + i) * N + N1 + j] = d.v[i * N2 + j];
  }
  return r;
}
// columns [C0, C0 + NC)
template <int C0, int NC, typename R, int M, int N, Mask<M, N> MK>
LQG_DEV auto cols(const Mat<R, M, N, MK>& a) {
  constexpr auto MR = mask_cols<C0, NC>(MK);
  Mat<R, M, NC, MR> r;
  LQG_UNROLL for (int i = 0; i < M; ++i)
    LQG_UNROLL for (int j = 0; j < NC; ++j) if (MR(i, j)) r.v[i * NC + j] = a.v[i * N + C0 + j];
  return r;
}
// reciprocal: hardware estimate + Newton steps (the IEEE division expands to ~10 instructions)
template <typename R> LQG_DEV R rcp_(R v);
template <> LQG_DEV float rcp_<float>(float v) {
  float r = __builtin_amdgcn_rcpf(v);
  return r * (2.0f - v * r);
}
template <> LQG_DEV double rcp_<double>(double v) {
  double r = __builtin_amdgcn_rcp(v);
  r = r * (2.0 - v * r);
  return r * (2.0 - v * r);
}
// inverse of a symmetric positive-definite matrix: a structurally DIAGONAL one is inverted entry by entry (and stays
// diagonal in the type), anything else goes through the dense Cholesky kernels
template <typename R, int N, Mask<N, N> MK>
LQG_DEV auto spd_inverse_masked(const Mat<R, N, N, MK>& a) {
  if constexpr (mask_is_diag(MK)) {
    Mat<R, N, N, MK> r;
    LQG_UNROLL for (int i = 0; i < N; ++i)
      if (MK(i, i)) r.v[i * N + i] = rcp_<R>(a.v[i * N + i]);
    return r;
  } else {
    R G[N * N], Lc[N * N], dinv[N], Lt[N * N], Gi[N * N];
    to_dense(a, G);
    chol_lower<R, N>(G, Lc, dinv);
    tri_inverse_lower<R, N>(Lc, dinv, Lt);
    spd_inverse_from_tri<R, N>(Lt, Gi);
    return from_dense<R, N, N>(Gi);
  }
}
// element I of assign_state, as a compile-time recursion so that both mask tests are `if constexpr` (inside an unrolled
// loop the compiler left the mask bytes as run-time loads + v_cndmask)
template <int I, typename R, int M, int N, Mask<M, N> MD, Mask<M, N> MS>
LQG_DEV void assign_state_at(Mat<R, M, N, MD>& dst, const Mat<R, M, N, MS>& src) {
  if constexpr (I < M * N) {
    if constexpr (MD.b[I]) {
      if constexpr (MS.b[I]) dst.v[I] = src.v[I];
      else dst.v[I] = R(0);
    }
    assign_state_at<I + 1>(dst, src);
  }
}
// assign into a loop-carried matrix of FIXED mask MD (a fixed point of the recursion: the result's mask is a subset)
template <typename R, int M, int N, Mask<M, N> MD, Mask<M, N> MS>
LQG_DEV void assign_state(Mat<R, M, N, MD>& dst, const Mat<R, M, N, MS>& src) {
  static_assert(mask_eq(mask_or(MD, MS), MD), "loop-carried mask is not a fixed point of the recursion");
  assign_state_at<0>(dst, src);
}

// y = (A - [[I_O, 0], [0, 0]]) x  (the deviation form of the mean update, lqg_kernels.hpp), every mask test resolved at
// compile time by recursion over (row, column): inside unrolled loops the compiler left some of them as run-time loads of
// the mask bytes followed by v_cndmask chains.
template <int O, int I, int J, typename R, int M, Mask<M, M> MK>
LQG_DEV void dev_matvec_term(const Mat<R, M, M, MK>& a, const R (&x)[M], R& v) {
  if constexpr (J < M) {
    if constexpr (I < O && I == J) {
      if constexpr (MK.b[I * M + I]) v += (a.v[I * M + I] - R(1)) * x[J];
      else v -= x[J];
    } else if constexpr (MK.b[I * M + J]) {
      v += a.v[I * M + J] * x[J];
    }
    dev_matvec_term<O, I, J + 1>(a, x, v);
  }
}
template <int O, int I, typename R, int M, Mask<M, M> MK>
LQG_DEV void dev_matvec_row(const Mat<R, M, M, MK>& a, const R (&x)[M], R (&y)[M]) {
  if constexpr (I < M) {
    R v = R(0);
    dev_matvec_term<O, I, 0>(a, x, v);
    y[I] = v;
    dev_matvec_row<O, I + 1>(a, x, y);
  }
}

// A - I with the diagonal formed as a DEVIATION (a_ii - 1 is exact for a_ii in [1/2, 2]; a structurally zero diagonal
// entry becomes -1): the starting point of the operator stream's F block, Fj - I (lqg_kernels.hpp, k_trial)
template <typename R, int N, Mask<N, N> MK>
LQG_DEV auto minus_identity(const Mat<R, N, N, MK>& a) {
  constexpr auto MR = mask_or(MK, mask_eye<N>());
  Mat<R, N, N, MR> r;
  LQG_UNROLL for (int i = 0; i < N; ++i)
    LQG_UNROLL for (int j = 0; j < N; ++j)
      if (MR(i, j)) r.v[i * N + j] = (i == j) ? (MK(i, i) ? a.v[i * N + i] - R(1) : R(-1)) : a.v[i * N + j];
  return r;
}
// dense row-major image of a masked matrix at p in the element type OT (structural zeros written out), masks resolved
// at compile time
template <int I, typename OT, typename R, int M, int N, Mask<M, N> MK>
LQG_DEV void store_dense(const Mat<R, M, N, MK>& a, OT* __restrict__ p) {
  if constexpr (I < M * N) {
    if constexpr (MK.b[I]) p[I] = (OT)a.v[I];
    else p[I] = OT(0);
    store_dense<I + 1>(a, p);
  }
}

// y = A x with dense vectors
template <typename R, int M, int N, Mask<M, N> MK>
LQG_DEV void matvec_acc(const Mat<R, M, N, MK>& a, const R (&x)[N], R (&y)[M]) {
  LQG_UNROLL for (int i = 0; i < M; ++i)
    LQG_UNROLL for (int j = 0; j < N; ++j) if (MK(i, j)) y[i] += a.v[i * N + j] * x[j];
}

}  // namespace lqg
