// lqg_small.hpp — register-resident small dense linear algebra for one LQG system per lane (gfx950).
//
// Every routine works on plain C arrays whose extents are template constants; all loops are fully
// unrolled so that every index is a compile-time constant and the arrays live in VGPRs (a runtime index
// would send them to scratch: cdna_hip_programming.md §5.4 rule 20).  No cross-lane traffic: with one
// system (or one trial) per lane the 64 lanes of a wave run 64 independent recursions in lock step and
// every flop is a plain v_fma.
#pragma once
#include <hip/hip_runtime.h>

#define LQG_UNROLL _Pragma("unroll")
#define LQG_DEV __device__ __forceinline__

namespace lqg {

template <typename R>
struct DView {  // device copy of lqg_view (strides in elements)
  const R* p;
  long sb, st, sr, sc;
};
template <typename R>
struct DTraj {  // device copy of lqg_traj
  const R* p;
  long sb, sn, st, sd;
};

template <typename R> LQG_DEV R rsqrt_(R v);
// fp32: hardware v_rsq_f32 (~1 ulp) + one Newton-Raphson step (-> ~0.5 ulp) = 5 instructions instead of the ~35 of an
// IEEE sqrt followed by an IEEE divide; NaN / negative / zero inputs propagate NaN / inf as the exact form does.
template <> LQG_DEV float rsqrt_<float>(float v) {
  float y = __builtin_amdgcn_rsqf(v);
  return y * (1.5f - 0.5f * v * y * y);
}
// fp64: v_rsq_f64 (~2^-26 relative) + three Newton-Raphson steps (quadratic: full double precision after two, the
// third absorbs the fma-free evaluation) = 13 instructions instead of the ~80 of an IEEE sqrt + IEEE divide.
template <> LQG_DEV double rsqrt_<double>(double v) {
  double y = __builtin_amdgcn_rsq(v);
  LQG_UNROLL for (int it = 0; it < 3; ++it) y = y * (1.5 - 0.5 * v * y * y);
  return y;
}
template <typename R> LQG_DEV R sqrt_(R v);
template <> LQG_DEV float sqrt_<float>(float v) { return sqrtf(v); }
template <> LQG_DEV double sqrt_<double>(double v) { return sqrt(v); }
template <typename R> LQG_DEV R log_(R v);
template <> LQG_DEV float log_<float>(float v) { return logf(v); }
template <> LQG_DEV double log_<double>(double v) { return log(v); }
template <typename R> LQG_DEV R abs_(R v) { return v < R(0) ? -v : v; }

// "positive, finite, non-zero" as INTEGER arithmetic on the bit pattern: key(v) < kPosFiniteLimit<R> iff 0 < v < inf (fp64:
// iff the high word is in [1, 0x7FEFFFFF], i.e. v >= 2^-1022 * 2^-20 as well).  The pattern libraries are compiled with
// -fno-honor-nans -fno-honor-infinities (lqg_amd/specialize.py), under which the compiler may assume that no floating-point
// operation yields NaN / inf and fold comparisons, selects and 0 * x accordingly; it may not touch integer operations on the
// bits.  The specialised sweeps track the largest key of the products of their Cholesky pivots' reciprocal square roots
// and overwrite a log-likelihood whose factorisation ever left the positive finite range with NaN, through an integer store.
LQG_DEV unsigned pos_finite_key(float v) { return __float_as_uint(v) - 1u; }
LQG_DEV unsigned pos_finite_key(double v) { return (unsigned)((unsigned long long)__double_as_longlong(v) >> 32) - 1u; }
template <typename R> inline constexpr unsigned kPosFiniteLimit = sizeof(R) == 4 ? 0x7F7FFFFFu : 0x7FEFFFFFu;
LQG_DEV void store_or_nan(float* dst, float v, bool poison) {
  *reinterpret_cast<unsigned*>(dst) = poison ? 0x7FC00000u : __float_as_uint(v);
}
LQG_DEV void store_or_nan(double* dst, double v, bool poison) {
  *reinterpret_cast<unsigned long long*>(dst) = poison ? 0x7FF8000000000000ull : (unsigned long long)__double_as_longlong(v);
}

// ---- strided loads (per-lane base pointer already includes the system offset) -------------------------------
template <typename R, int ROWS, int COLS>
LQG_DEV void load_mat(const R* __restrict__ p, long sr, long sc, R (&out)[ROWS * COLS]) {
  LQG_UNROLL for (int i = 0; i < ROWS; ++i)
    LQG_UNROLL for (int j = 0; j < COLS; ++j) out[i * COLS + j] = p[i * sr + j * sc];
}
template <typename R, int N>
LQG_DEV void load_vec(const R* __restrict__ p, long sr, R (&out)[N]) {
  LQG_UNROLL for (int i = 0; i < N; ++i) out[i] = p[i * sr];
}
// symmetric part of a stored square matrix, full (mirrored) register image
template <typename R, int N>
LQG_DEV void load_sym(const R* __restrict__ p, long sr, long sc, R (&out)[N * N]) {
  LQG_UNROLL for (int i = 0; i < N; ++i)
    LQG_UNROLL for (int j = i; j < N; ++j) {
      R v = (i == j) ? p[i * sr + i * sc] : R(0.5) * (p[i * sr + j * sc] + p[j * sr + i * sc]);
      out[i * N + j] = v;
      out[j * N + i] = v;
    }
}
// out = V V^T for a stored V[N, nv] (nv is a run-time extent: the columns stream through registers)
template <typename R, int N>
LQG_DEV void load_gram(const R* __restrict__ p, long sr, long sc, int nv, R (&out)[N * N]) {
  LQG_UNROLL for (int i = 0; i < N * N; ++i) out[i] = R(0);
  for (int k = 0; k < nv; ++k) {
    R col[N];
    LQG_UNROLL for (int i = 0; i < N; ++i) col[i] = p[i * sr + k * sc];
    LQG_UNROLL for (int i = 0; i < N; ++i)
      LQG_UNROLL for (int j = i; j < N; ++j) out[i * N + j] += col[i] * col[j];
  }
  LQG_UNROLL for (int i = 0; i < N; ++i)
    LQG_UNROLL for (int j = 0; j < i; ++j) out[i * N + j] = out[j * N + i];
}
template <typename R, int ROWS, int COLS>
LQG_DEV void store_mat(R* __restrict__ p, long sr, long sc, const R (&in)[ROWS * COLS]) {
  LQG_UNROLL for (int i = 0; i < ROWS; ++i)
    LQG_UNROLL for (int j = 0; j < COLS; ++j) p[i * sr + j * sc] = in[i * COLS + j];
}

// ---- products --------------------------------------------------------------------------------------------------
// C[M,N] = A[M,K] B[K,N]
template <typename R, int M, int K, int N>
LQG_DEV void mm(const R (&A)[M * K], const R (&B)[K * N], R (&C)[M * N]) {
  LQG_UNROLL for (int i = 0; i < M; ++i)
    LQG_UNROLL for (int j = 0; j < N; ++j) {
      R acc = A[i * K] * B[j];
      LQG_UNROLL for (int k = 1; k < K; ++k) acc += A[i * K + k] * B[k * N + j];
      C[i * N + j] = acc;
    }
}
// C[M,N] = A[K,M]^T B[K,N]
template <typename R, int M, int K, int N>
LQG_DEV void mtm(const R (&A)[K * M], const R (&B)[K * N], R (&C)[M * N]) {
  LQG_UNROLL for (int i = 0; i < M; ++i)
    LQG_UNROLL for (int j = 0; j < N; ++j) {
      R acc = A[i] * B[j];
      LQG_UNROLL for (int k = 1; k < K; ++k) acc += A[k * M + i] * B[k * N + j];
      C[i * N + j] = acc;
    }
}
// C[M,N] = A[M,K] B[N,K]^T
template <typename R, int M, int K, int N>
LQG_DEV void mmt(const R (&A)[M * K], const R (&B)[N * K], R (&C)[M * N]) {
  LQG_UNROLL for (int i = 0; i < M; ++i)
    LQG_UNROLL for (int j = 0; j < N; ++j) {
      R acc = A[i * K] * B[j * K];
      LQG_UNROLL for (int k = 1; k < K; ++k) acc += A[i * K + k] * B[j * K + k];
      C[i * N + j] = acc;
    }
}
// symmetric result: C[N,N] = A[N,K] B[N,K]^T + D, only the upper triangle is computed, then mirrored
template <typename R, int N, int K>
LQG_DEV void mmt_sym_add(const R (&A)[N * K], const R (&B)[N * K], const R (&D)[N * N], R (&C)[N * N]) {
  LQG_UNROLL for (int i = 0; i < N; ++i)
    LQG_UNROLL for (int j = i; j < N; ++j) {
      R acc = D[i * N + j];
      LQG_UNROLL for (int k = 0; k < K; ++k) acc += A[i * K + k] * B[j * K + k];
      C[i * N + j] = acc;
      C[j * N + i] = acc;
    }
}
// symmetric result: C[N,N] = A[K,N]^T B[K,N] (upper computed, mirrored)
template <typename R, int N, int K>
LQG_DEV void mtm_sym(const R (&A)[K * N], const R (&B)[K * N], R (&C)[N * N]) {
  LQG_UNROLL for (int i = 0; i < N; ++i)
    LQG_UNROLL for (int j = i; j < N; ++j) {
      R acc = A[i] * B[j];
      LQG_UNROLL for (int k = 1; k < K; ++k) acc += A[k * N + i] * B[k * N + j];
      C[i * N + j] = acc;
      C[j * N + i] = acc;
    }
}

// ---- symmetric positive-definite kernels ----------------------------------------------------------------------
// Lower Cholesky factor of the symmetric A (upper triangle never read) and the reciprocals of its diagonal.
// A non-positive pivot yields NaN/inf exactly like an unguarded sqrt/divide (the reference propagates them).
template <typename R, int N>
LQG_DEV void chol_lower(const R (&A)[N * N], R (&Lc)[N * N], R (&dinv)[N]) {
  LQG_UNROLL for (int j = 0; j < N; ++j) {
    R dj = A[j * N + j];
    LQG_UNROLL for (int k = 0; k < j; ++k) dj -= Lc[j * N + k] * Lc[j * N + k];
    R rs = rsqrt_<R>(dj);
    dinv[j] = rs;
    Lc[j * N + j] = dj * rs;
    LQG_UNROLL for (int i = j + 1; i < N; ++i) {
      R v = A[i * N + j];
      LQG_UNROLL for (int k = 0; k < j; ++k) v -= Lc[i * N + k] * Lc[j * N + k];
      Lc[i * N + j] = v * rs;
    }
    LQG_UNROLL for (int i = 0; i < j; ++i) Lc[i * N + j] = R(0);
  }
}
// inverse of a lower-triangular factor (explicit, so that applying it is N^2/2 independent FMAs)
template <typename R, int N>
LQG_DEV void tri_inverse_lower(const R (&Lc)[N * N], const R (&dinv)[N], R (&Li)[N * N]) {
  LQG_UNROLL for (int j = 0; j < N; ++j) {
    LQG_UNROLL for (int i = 0; i < j; ++i) Li[i * N + j] = R(0);
    Li[j * N + j] = dinv[j];
    LQG_UNROLL for (int i = j + 1; i < N; ++i) {
      R acc = R(0);
      LQG_UNROLL for (int k = j; k < i; ++k) acc -= Lc[i * N + k] * Li[k * N + j];
      Li[i * N + j] = acc * dinv[i];
    }
  }
}
// Ainv = Li^T Li (symmetric, mirrored)
template <typename R, int N>
LQG_DEV void spd_inverse_from_tri(const R (&Li)[N * N], R (&Ainv)[N * N]) {
  LQG_UNROLL for (int i = 0; i < N; ++i)
    LQG_UNROLL for (int j = i; j < N; ++j) {
      R acc = R(0);
      LQG_UNROLL for (int k = j; k < N; ++k) acc += Li[k * N + i] * Li[k * N + j];
      Ainv[i * N + j] = acc;
      Ainv[j * N + i] = acc;
    }
}

// smallest eigenvalue of a symmetric matrix (lqr.py:27 uses only evals[0]).
// N = 1, 2: closed form; N >= 3: cyclic Jacobi, fixed sweep count, branch-free rotations.
template <typename R, int N>
LQG_DEV R min_eig_sym(const R (&H)[N * N]) {
  if constexpr (N == 1) {
    return H[0];
  } else if constexpr (N == 2) {
    R hm = R(0.5) * (H[0] + H[3]), hd = R(0.5) * (H[0] - H[3]);
    return hm - sqrt_<R>(hd * hd + H[1] * H[1]);
  } else {
    R a[N * N];
    LQG_UNROLL for (int i = 0; i < N * N; ++i) a[i] = H[i];
    for (int sweep = 0; sweep < 10; ++sweep) {
      LQG_UNROLL for (int p = 0; p < N; ++p)
        LQG_UNROLL for (int q = p + 1; q < N; ++q) {
          R apq = a[p * N + q];
          R nz = (apq != R(0)) ? R(1) : R(0);
          R den = (apq != R(0)) ? R(2) * apq : R(1);
          R theta = (a[q * N + q] - a[p * N + p]) / den;
          R t = ((theta >= R(0)) ? R(1) : R(-1)) / (abs_<R>(theta) + sqrt_<R>(theta * theta + R(1)));
          t *= nz;
          R c = rsqrt_<R>(t * t + R(1)), s = t * c;
          LQG_UNROLL for (int k = 0; k < N; ++k) {
            R akp = a[k * N + p], akq = a[k * N + q];
            a[k * N + p] = c * akp - s * akq;
            a[k * N + q] = s * akp + c * akq;
          }
          LQG_UNROLL for (int k = 0; k < N; ++k) {
            R apk = a[p * N + k], aqk = a[q * N + k];
            a[p * N + k] = c * apk - s * aqk;
            a[q * N + k] = s * apk + c * aqk;
          }
        }
    }
    R mn = a[0];
    LQG_UNROLL for (int i = 1; i < N; ++i) mn = (a[i * N + i] < mn) ? a[i * N + i] : mn;
    return mn;
  }
}

}  // namespace lqg
