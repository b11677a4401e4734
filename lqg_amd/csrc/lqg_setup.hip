// lqg_setup.hip — model-zoo setup arithmetic that PyTorch would hand to rocSOLVER: the point-mass discretisation of
// lqg/tracking/point_mass.py:50-144 (matrix exponential of the zero-order hold, Van Loan process-noise block, eigenvalue
// clipping, upper Cholesky factor), one candidate per lane, fp64, everything in registers.
//
// Not part of the hot path — but it sits in front of it on every evaluation of an optimiser / sampler loop: built from
// torch.linalg.matrix_exp / eigh / cholesky, constructing ONE candidate batch of PointMassBoundedActor costs ~2 ms of
// host-synchronising library calls on 3x3 .. 6x6 matrices (more than the whole log-likelihood), and cannot be captured
// into a hipGraph.  This kernel is the no-grad route (a differentiable construction keeps the torch functions).
#include <hip/hip_runtime.h>

#include <type_traits>

#include "../../include/lqg_hip.h"

namespace {

#define SETUP_UNROLL _Pragma("unroll")

template <int N>
__device__ void matmul(const double (&a)[N * N], const double (&b)[N * N], double (&c)[N * N]) {
  SETUP_UNROLL for (int i = 0; i < N; ++i)
    SETUP_UNROLL for (int j = 0; j < N; ++j) {
      double v = 0.0;
      SETUP_UNROLL for (int k = 0; k < N; ++k) v = fma(a[i * N + k], b[k * N + j], v);
      c[i * N + j] = v;
    }
}

// exp(M): scaling and squaring around a degree-16 Taylor polynomial evaluated by Horner's rule.  The scaling brings the
// 1-norm below 1/4 (truncation error < 0.25^17 / 17! ~ 2e-25), so the result is exact to rounding of the 16 + s products.
template <int N>
__device__ void expm(double (&M)[N * N]) {
  double nrm = 0.0;
  SETUP_UNROLL for (int j = 0; j < N; ++j) {
    double cs = 0.0;
    SETUP_UNROLL for (int i = 0; i < N; ++i) cs += fabs(M[i * N + j]);
    nrm = fmax(nrm, cs);
  }
  int s = 0;
  if (nrm > 0.25) s = (int)ceil(log2(nrm / 0.25));
  if (s > 60) s = 60;
  const double sc = ldexp(1.0, -s);
  double X[N * N], E[N * N], T[N * N];
  SETUP_UNROLL for (int e = 0; e < N * N; ++e) X[e] = M[e] * sc;
  SETUP_UNROLL for (int i = 0; i < N; ++i)
    SETUP_UNROLL for (int j = 0; j < N; ++j) E[i * N + j] = (i == j) ? 1.0 : 0.0;
  for (int k = 16; k >= 1; --k) {                       // E = I + X E / k
    matmul<N>(X, E, T);
    const double ik = 1.0 / (double)k;
    SETUP_UNROLL for (int i = 0; i < N; ++i)
      SETUP_UNROLL for (int j = 0; j < N; ++j) E[i * N + j] = ((i == j) ? 1.0 : 0.0) + T[i * N + j] * ik;
  }
  for (int q = 0; q < s; ++q) {
    matmul<N>(E, E, T);
    SETUP_UNROLL for (int e = 0; e < N * N; ++e) E[e] = T[e];
  }
  SETUP_UNROLL for (int e = 0; e < N * N; ++e) M[e] = E[e];
}

// cyclic Jacobi on a symmetric 3x3 matrix: A -> diagonal, U accumulates the rotations (A_in = U diag U')
__device__ void jacobi3(double (&A)[9], double (&U)[9]) {
  SETUP_UNROLL for (int e = 0; e < 9; ++e) U[e] = (e % 4 == 0) ? 1.0 : 0.0;
  auto rotate = [&](auto pc, auto qc) {
    constexpr int p = decltype(pc)::value, q = decltype(qc)::value;
    const double apq = A[p * 3 + q];
    if (apq == 0.0) return;
    const double theta = (A[q * 3 + q] - A[p * 3 + p]) / (2.0 * apq);
    const double t = copysign(1.0, theta) / (fabs(theta) + sqrt(theta * theta + 1.0));
    const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
    SETUP_UNROLL for (int k = 0; k < 3; ++k) {           // columns p, q of A and U
      const double akp = A[k * 3 + p], akq = A[k * 3 + q];
      A[k * 3 + p] = c * akp - s * akq;
      A[k * 3 + q] = s * akp + c * akq;
      const double ukp = U[k * 3 + p], ukq = U[k * 3 + q];
      U[k * 3 + p] = c * ukp - s * ukq;
      U[k * 3 + q] = s * ukp + c * ukq;
    }
    SETUP_UNROLL for (int k = 0; k < 3; ++k) {           // rows p, q of A
      const double apk = A[p * 3 + k], aqk = A[q * 3 + k];
      A[p * 3 + k] = c * apk - s * aqk;
      A[q * 3 + k] = s * apk + c * aqk;
    }
    A[p * 3 + q] = 0.0;
    A[q * 3 + p] = 0.0;
  };
  for (int sweep = 0; sweep < 12; ++sweep) {
    rotate(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
    rotate(std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{});
    rotate(std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{});
  }
}

__global__ void __launch_bounds__(64) k_point_mass_setup(long n, const double* __restrict__ damping, const double* __restrict__ mass,
                                                         const double* __restrict__ tau, const double* __restrict__ av, double dt,
                                                         double psd_eps, double* __restrict__ Ao, double* __restrict__ Bo,
                                                         double* __restrict__ Vo) {
  const long i = blockIdx.x * 64L + threadIdx.x;
  if (i >= n) return;
  const double dmp = damping[i], m = mass[i], ta = tau[i], g = 1e-2 * av[i];
  // continuous point mass + muscle filter                                    point_mass.py:113-121
  const double Ac[9] = {0.0, 1.0, 0.0, 0.0, -dmp / m, 1.0 / m, 0.0, 0.0, -1.0 / ta};
  const double Bc[3] = {0.0, 0.0, 1.0 / ta};
  {   // zero-order hold: expm([[A, B], [0, 0]] dt)                              point_mass.py:50-79
    double M[16];
    SETUP_UNROLL for (int e = 0; e < 16; ++e) M[e] = 0.0;
    SETUP_UNROLL for (int r = 0; r < 3; ++r) {
      SETUP_UNROLL for (int c = 0; c < 3; ++c) M[r * 4 + c] = Ac[r * 3 + c] * dt;
      M[r * 4 + 3] = Bc[r] * dt;
    }
    expm<4>(M);
    SETUP_UNROLL for (int r = 0; r < 3; ++r) {
      SETUP_UNROLL for (int c = 0; c < 3; ++c) Ao[i * 9 + r * 3 + c] = M[r * 4 + c];
      Bo[i * 3 + r] = M[r * 4 + 3];
    }
  }
  double Q[9];
  {   // Van Loan block of expm([[A, G G'], [0, -A']] dt)                        point_mass.py:82-110
    double M[36];
    SETUP_UNROLL for (int e = 0; e < 36; ++e) M[e] = 0.0;
    SETUP_UNROLL for (int r = 0; r < 3; ++r)
      SETUP_UNROLL for (int c = 0; c < 3; ++c) {
        M[r * 6 + c] = Ac[r * 3 + c] * dt;
        M[r * 6 + 3 + c] = (g * Bc[r]) * (g * Bc[c]) * dt;
        M[(3 + r) * 6 + 3 + c] = -Ac[c * 3 + r] * dt;
      }
    expm<6>(M);
    SETUP_UNROLL for (int r = 0; r < 3; ++r)
      SETUP_UNROLL for (int c = 0; c < 3; ++c) Q[r * 3 + c] = M[r * 6 + 3 + c];
  }
  // symmetrise, clip the eigenvalues from below, rebuild                      point_mass.py:130-144
  double S[9], U[9];
  SETUP_UNROLL for (int r = 0; r < 3; ++r)
    SETUP_UNROLL for (int c = 0; c < 3; ++c) S[r * 3 + c] = 0.5 * (Q[r * 3 + c] + Q[c * 3 + r]);
  jacobi3(S, U);
  double w[3];
  SETUP_UNROLL for (int k = 0; k < 3; ++k) w[k] = fmax(S[k * 3 + k], psd_eps);
  double P[9];
  SETUP_UNROLL for (int r = 0; r < 3; ++r)
    SETUP_UNROLL for (int c = 0; c < 3; ++c) {
      double v = 0.0;
      SETUP_UNROLL for (int k = 0; k < 3; ++k) v = fma(U[r * 3 + k] * w[k], U[c * 3 + k], v);
      P[r * 3 + c] = v;
    }
  // upper Cholesky factor R, R' R = P (jax.scipy.linalg.cholesky's default)  point_mass.py:123
  double R[9];
  SETUP_UNROLL for (int e = 0; e < 9; ++e) R[e] = 0.0;
  SETUP_UNROLL for (int j = 0; j < 3; ++j) {
    double dsum = P[j * 3 + j];
    SETUP_UNROLL for (int k = 0; k < 3; ++k)
      if (k < j) dsum -= R[k * 3 + j] * R[k * 3 + j];
    const double rjj = sqrt(dsum);
    R[j * 3 + j] = rjj;
    SETUP_UNROLL for (int c = 0; c < 3; ++c)
      if (c > j) {
        double v = 0.5 * (P[j * 3 + c] + P[c * 3 + j]);
        SETUP_UNROLL for (int k = 0; k < 3; ++k)
          if (k < j) v -= R[k * 3 + j] * R[k * 3 + c];
        R[j * 3 + c] = v / rjj;
      }
  }
  SETUP_UNROLL for (int e = 0; e < 9; ++e) Vo[i * 9 + e] = R[e];
}

}  // namespace

extern "C" int lqg_point_mass_setup(int64_t n, const double* damping, const double* mass, const double* tau,
                                    const double* action_variability, double dt, double psd_eps, double* A, double* B, double* V,
                                    void* stream) {
  if (n < 0) return LQG_ERR_ARG;
  if (n == 0) return 0;
  if (!damping || !mass || !tau || !action_variability || !A || !B || !V) return LQG_ERR_NULL;
  hipLaunchKernelGGL(k_point_mass_setup, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, (hipStream_t)stream, (long)n, damping, mass,
                     tau, action_variability, dt, psd_eps, A, B, V);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}


// ---------------------------------------------------------------------------------------------------------------------------
// lqg_precondition_flags (include/lqg_hip.h): the value-dependent preconditions a FROZEN evaluation path rests on, checked
// on the device so that a captured hipGraph (lqg_amd/infer/graphed.py) can poison its result instead of being silently
// wrong.  Gershgorin bounds only (elementwise; conservative): every eigenvalue of a symmetric M lies in
// [min_i (M_ii - sum_{j != i} |M_ij|), max_i (M_ii + sum_{j != i} |M_ij|)].
namespace {

template <typename R>
struct GuardArgs {
  const R *Rm, *Q, *Qf, *V;
  long R_sb, R_st, R_sr, R_sc, Q_sb, Q_st, Q_sr, Q_sc, Qf_sb, Qf_sr, Qf_sc, V_sb, V_st, V_sr, V_sc;
  long n_sys;
  int T, b, u, d, nv;
  double eps, max_cond;
  int check_cond;
  int32_t* ok;
};

template <typename R>
__device__ void gersh(const R* p, long sr, long sc, int n, double& lo, double& hi) {
  lo = 1e300;
  hi = -1e300;
  for (int i = 0; i < n; ++i) {
    double diag = (double)p[i * sr + i * sc], off = 0.0;
    for (int j = 0; j < n; ++j)
      if (j != i) off += fabs(0.5 * ((double)p[i * sr + j * sc] + (double)p[j * sr + i * sc]));
    lo = fmin(lo, diag - off);
    hi = fmax(hi, diag + off);
  }
}

template <typename R>
__global__ void __launch_bounds__(256) k_precondition_flags(const GuardArgs<R> a) {
  __shared__ int bad;
  if (threadIdx.x == 0) bad = 0;
  __syncthreads();
  int mine = 0;
  for (long s = threadIdx.x; s < a.n_sys; s += blockDim.x) {
    double lo, hi;
    gersh(a.Qf + s * a.Qf_sb, a.Qf_sr, a.Qf_sc, a.b, lo, hi);
    if (!(lo >= -1e-12)) mine = 1;
    const int tr = a.R_st ? a.T : 1, tq = a.Q_st ? a.T : 1, tv = a.V_st ? a.T : 1;
    for (int t = 0; t < tr; ++t) {
      gersh(a.Rm + s * a.R_sb + t * a.R_st, a.R_sr, a.R_sc, a.u, lo, hi);
      if (!(lo >= a.eps)) mine = 1;
    }
    for (int t = 0; t < tq; ++t) {
      gersh(a.Q + s * a.Q_sb + t * a.Q_st, a.Q_sr, a.Q_sc, a.b, lo, hi);
      if (!(lo >= -1e-12)) mine = 1;
    }
    if (a.check_cond) {
      for (int t = 0; t < tv; ++t) {                          // Gershgorin bounds of (V V')[:d, :d]
        const R* V = a.V + s * a.V_sb + t * a.V_st;
        lo = 1e300;
        hi = -1e300;
        for (int i = 0; i < a.d; ++i) {
          double diag = 0.0, off = 0.0;
          for (int j = 0; j < a.d; ++j) {
            double g = 0.0;
            for (int k = 0; k < a.nv; ++k) g += (double)V[i * a.V_sr + k * a.V_sc] * (double)V[j * a.V_sr + k * a.V_sc];
            if (j == i) diag = g; else off += fabs(g);
          }
          lo = fmin(lo, diag - off);
          hi = fmax(hi, diag + off);
        }
        if (!(lo > 0.0 && hi <= a.max_cond * lo)) mine = 1;
      }
    }
  }
  if (mine) atomicOr(&bad, 1);
  __syncthreads();
  if (threadIdx.x == 0) a.ok[0] = bad ? 0 : 1;
}

}  // namespace

extern "C" int lqg_precondition_flags(const lqg_problem* p, double max_cond, int32_t check_cond, int32_t* ok, void* stream) {
  if (!p || !ok) return LQG_ERR_NULL;
  if (p->dtype != LQG_F32 && p->dtype != LQG_F64) return LQG_ERR_ARG;
  const lqg_spec& a = p->actor;
  const lqg_view& V = p->dynamics.V;
  if (!a.R.ptr || !a.Q.ptr || !a.Qf.ptr || (check_cond && !V.ptr)) return LQG_ERR_NULL;
  if (p->n_sys <= 0) return 0;
#define LQG_GUARD(R_)                                                                                                      \
  {                                                                                                                        \
    GuardArgs<R_> g{static_cast<const R_*>(a.R.ptr), static_cast<const R_*>(a.Q.ptr), static_cast<const R_*>(a.Qf.ptr),   \
                    static_cast<const R_*>(V.ptr), a.R.sb, p->T > 1 ? a.R.st : 0, a.R.sr, a.R.sc, a.Q.sb,                 \
                    p->T > 1 ? a.Q.st : 0, a.Q.sr, a.Q.sc, a.Qf.sb, a.Qf.sr, a.Qf.sc, V.sb, p->T > 1 ? V.st : 0, V.sr,    \
                    V.sc, (long)p->n_sys, p->T, p->dims.b, p->dims.u, p->dims.d, p->dims.nvd, p->eps, max_cond,           \
                    check_cond, ok};                                                                                       \
    hipLaunchKernelGGL(k_precondition_flags<R_>, dim3(1), dim3(256), 0, (hipStream_t)stream, g);                           \
  }
  if (p->dtype == LQG_F64) LQG_GUARD(double) else LQG_GUARD(float)
#undef LQG_GUARD
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

// ---------------------------------------------------------------- central differences around the candidate sweep
// The one-vector value + gradient of the inference loops (lqg/infer/mle.py:17-23, lqg/optim.py:142-147 use jax.grad) is K
// points x (2 P + 1) perturbed parameter vectors through ONE candidate sweep (lqg_amd/infer/graphed.py).  Inside the replayed
// hipGraph every kernel costs ~4.7 us whatever it does, and the torch glue on either side of the sweep was 12 of its 26
// kernels: these two replace it.
namespace {
// flat[c, f] = base[f] + sum_p exp(z[k, p] + s(c, p) h) D[p, f],  c = k (2 P + 1) + j:  j = 0 centre, 1 + p: +h on p, 1 + P + p: -h
template <typename R>
__global__ void __launch_bounds__(256) k_fd_candidates(const double* __restrict__ z, const double* __restrict__ base,
                                                       const double* __restrict__ Dm, R* __restrict__ flat, long K, int P, long F,
                                                       double h) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  const long C = K * (2 * P + 1);
  if (e >= C * F) return;
  const long c = e / F, f = e - c * F;
  const long k = c / (2 * P + 1);
  const int j = (int)(c - k * (2 * P + 1));
  double acc = base[f];
  for (int p = 0; p < P; ++p) {
    const double s = (j == 1 + p) ? h : (j == 1 + P + p) ? -h : 0.0;
    acc = fma(exp(z[k * P + p] + s), Dm[(long)p * F + f], acc);
  }
  flat[e] = (R)acc;
}
// out[k, 0] = obj[k, 0], out[k, 1 + p] = (obj[k, 1 + p] - obj[k, 1 + P + p]) / (2 h); NaN everywhere when *ok == 0
__global__ void __launch_bounds__(256) k_fd_combine(const double* __restrict__ obj, const int32_t* __restrict__ ok, double* __restrict__ out,
                                                    long K, int P, double h) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= K * (1 + P)) return;
  const long k = e / (1 + P);
  const int j = (int)(e - k * (1 + P));
  const double* f = obj + k * (2 * P + 1);
  double v = (j == 0) ? f[0] : (f[j] - f[P + j]) / (2.0 * h);
  if (ok && *ok == 0) v = __builtin_nan("");
  out[e] = v;
}
}  // namespace

extern "C" int lqg_fd_candidates(const double* z, const double* base, const double* D, void* flat, int32_t dtype, int64_t K,
                                 int32_t P, int64_t F, double h, void* stream) {
  if (!z || !base || !D || !flat) return LQG_ERR_NULL;
  if (K <= 0 || P <= 0 || F <= 0 || (dtype != LQG_F32 && dtype != LQG_F64)) return LQG_ERR_ARG;
  const long n = K * (2 * P + 1) * F;
  const dim3 grid((unsigned)((n + 255) / 256)), block(256);
  if (dtype == LQG_F64)
    hipLaunchKernelGGL(k_fd_candidates<double>, grid, block, 0, (hipStream_t)stream, z, base, D, static_cast<double*>(flat), (long)K,
                       P, (long)F, h);
  else
    hipLaunchKernelGGL(k_fd_candidates<float>, grid, block, 0, (hipStream_t)stream, z, base, D, static_cast<float*>(flat), (long)K, P,
                       (long)F, h);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

extern "C" int lqg_fd_combine(const double* obj, const int32_t* ok, double* out, int64_t K, int32_t P, double h, void* stream) {
  if (!obj || !out) return LQG_ERR_NULL;
  if (K <= 0 || P <= 0 || !(h > 0.0)) return LQG_ERR_ARG;
  const long n = K * (1 + P);
  hipLaunchKernelGGL(k_fd_combine, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, obj, ok, out, (long)K, P, h);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}
