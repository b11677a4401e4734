/*
 * lqg_oracle.c — plain-C CPU restatement of the reference's LQG solve path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load liblqg_oracle.so; nothing under lqg_amd/ links, loads or calls it.  It is the checker (and the
 * "port" CPU baseline timed beside the GPU path), never the product path.
 *
 * Parity status: PINNED — checked in tests/test_oracle.py against the golden vectors in tests/golden/,
 * which were produced by the reference's own source executed under oracle/jax_standin.py in the build
 * container (generator: oracle/gen_golden.py), and against reference-independent analytic pins.  The
 * reference's own tests (tests/lqg_test.py, tests/infer_test.py) hold no numeric golden vectors.
 *
 * It follows the reference operation for operation (no symmetry exploitation, no Schur form, LU solves
 * with partial pivoting where the reference calls jnp.linalg.solve/inv, Jacobi eigenvalues for eigh,
 * Cholesky for the MultivariateNormal) so that it is an independent check of the restructured HIP
 * kernels.  One structural liberty: per-system work (Riccati, Kalman, joint system, Sigma recursion) is
 * done once per system instead of once per trial — under jax.vmap the reference does the same
 * (lqg/system.py:241 batches only the data-dependent values).
 *
 * Uses the argument structs of include/lqg_hip.h with HOST pointers.
 *
 * Functions (reference file:line in lqg_oracle_body.inc):
 *   lqg_oracle_riccati_backward   lqg/control/lqr.py:16-42
 *   lqg_oracle_kalman_forward     lqg/belief/kf.py:6-21
 *   lqg_oracle_conditional_moments lqg/system.py:142-235
 *   lqg_oracle_log_likelihood     lqg/system.py:237-248
 *   lqg_oracle_simulate           lqg/system.py:62-140
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/lqg_hip.h"

#define MAXB 40
#define MAXB2 (MAXB * MAXB)
#define MAXM (2 * MAXB)
#define MAXM2 (MAXM * MAXM)
#define LQG_ORACLE_MAXU MAXB

#define REAL double
#define SUF _f64
#define FABS fabs
#define SQRT sqrt
#define LOG log
#include "lqg_oracle_body.inc"
#undef REAL
#undef SUF
#undef FABS
#undef SQRT
#undef LOG

#define REAL float
#define SUF _f32
#define FABS fabsf
#define SQRT sqrtf
#define LOG logf
#include "lqg_oracle_body.inc"
#undef REAL
#undef SUF
#undef FABS
#undef SQRT
#undef LOG

static int check(const lqg_problem* p) {
  if (!p) return LQG_ERR_NULL;
  const lqg_dims* d = &p->dims;
  if (d->x > MAXB || d->b > MAXB || d->u > MAXB || d->y > MAXB || d->nva > MAXB || d->nwa > MAXB ||
      d->nvd > MAXB || d->nwd > MAXB || d->d > d->x || d->d < 1)
    return LQG_ERR_DIMS;
  if (p->dtype != LQG_F32 && p->dtype != LQG_F64) return LQG_ERR_ARG;
  return 0;
}

int lqg_oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

void lqg_oracle_set_threads(int n) {
#ifdef _OPENMP
  omp_set_num_threads(n);
#else
  (void)n;
#endif
}

int lqg_oracle_riccati_backward(const lqg_problem* p, lqg_view L, lqg_view l, lqg_view H) {
  int rc = check(p);
  if (rc) return rc;
  const int T = p->T, u = p->dims.u, b = p->dims.b;
#pragma omp parallel for schedule(static)
  for (int64_t s = 0; s < p->n_sys; ++s) {
    if (p->dtype == LQG_F64) {
      double* w = (double*)malloc(sizeof(double) * (size_t)T * (u * b + u + u * u));
      double *Lo = w, *lo = Lo + (size_t)T * u * b, *Ho = lo + (size_t)T * u;
      riccati_one_f64(p, s, Lo, lo, Ho);
      for (int t = 0; t < T; ++t) {
        store_mat_f64(&L, s, t, u, b, Lo + (size_t)t * u * b);
        store_vec_f64(&l, s, t, u, lo + (size_t)t * u);
        store_mat_f64(&H, s, t, u, u, Ho + (size_t)t * u * u);
      }
      free(w);
    } else {
      float* w = (float*)malloc(sizeof(float) * (size_t)T * (u * b + u + u * u));
      float *Lo = w, *lo = Lo + (size_t)T * u * b, *Ho = lo + (size_t)T * u;
      riccati_one_f32(p, s, Lo, lo, Ho);
      for (int t = 0; t < T; ++t) {
        store_mat_f32(&L, s, t, u, b, Lo + (size_t)t * u * b);
        store_vec_f32(&l, s, t, u, lo + (size_t)t * u);
        store_mat_f32(&H, s, t, u, u, Ho + (size_t)t * u * u);
      }
      free(w);
    }
  }
  return 0;
}

int lqg_oracle_kalman_forward(const lqg_problem* p, lqg_view K) {
  int rc = check(p);
  if (rc) return rc;
  const int T = p->T, y = p->dims.y, b = p->dims.b;
#pragma omp parallel for schedule(static)
  for (int64_t s = 0; s < p->n_sys; ++s) {
    if (p->dtype == LQG_F64) {
      double* w = (double*)malloc(sizeof(double) * (size_t)T * b * y);
      kalman_one_f64(p, s, w);
      for (int t = 0; t < T; ++t) store_mat_f64(&K, s, t, b, y, w + (size_t)t * b * y);
      free(w);
    } else {
      float* w = (float*)malloc(sizeof(float) * (size_t)T * b * y);
      kalman_one_f32(p, s, w);
      for (int t = 0; t < T; ++t) store_mat_f32(&K, s, t, b, y, w + (size_t)t * b * y);
      free(w);
    }
  }
  return 0;
}

static int moments_impl(const lqg_problem* p, const lqg_traj* x, const lqg_traj* mu, const lqg_view* Sigma,
                        void* ll, int64_t ll_sb, int64_t ll_sn) {
  int rc = check(p);
  if (rc) return rc;
  if (!x || !x->ptr) return LQG_ERR_NULL;
#pragma omp parallel
  {
    /* one scratch buffer per thread (not per system: 256 threads hammering malloc would measure the allocator) */
    void* w = malloc(p->dtype == LQG_F64 ? sizeof(double) * work_reals_f64(p) : sizeof(float) * work_reals_f32(p));
#pragma omp for schedule(dynamic, 8)
    for (int64_t s = 0; s < p->n_sys; ++s) {
      if (p->dtype == LQG_F64)
        moments_one_f64(p, s, x, mu, Sigma, ll ? (double*)ll + s * ll_sb : NULL, ll_sn, (double*)w);
      else
        moments_one_f32(p, s, x, mu, Sigma, ll ? (float*)ll + s * ll_sb : NULL, ll_sn, (float*)w);
    }
    free(w);
  }
  return 0;
}

int lqg_oracle_conditional_moments(const lqg_problem* p, lqg_traj x, lqg_traj mu, lqg_view Sigma) {
  return moments_impl(p, &x, &mu, &Sigma, NULL, 0, 0);
}

int lqg_oracle_log_likelihood(const lqg_problem* p, lqg_traj x, void* ll, int64_t ll_sb, int64_t ll_sn) {
  if (!ll) return LQG_ERR_NULL;
  return moments_impl(p, &x, NULL, NULL, ll, ll_sb, ll_sn);
}

int lqg_oracle_simulate(const lqg_problem* p, lqg_view L, lqg_view l, lqg_view K, lqg_traj eps, lqg_traj eta,
                        lqg_view x0, lqg_view xhat0, lqg_traj xs, lqg_traj xhat, lqg_traj ys, lqg_traj us) {
  int rc = check(p);
  if (rc) return rc;
  const int64_t total = p->n_sys * p->n_trials;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < total; ++i) {
    const int64_t s = i / p->n_trials, n = i % p->n_trials;
    if (p->dtype == LQG_F64) simulate_one_f64(p, s, n, &L, &l, &K, &eps, &eta, &x0, &xhat0, &xs, &xhat, &ys, &us);
    else simulate_one_f32(p, s, n, &L, &l, &K, &eps, &eta, &x0, &xhat0, &xs, &xhat, &ys, &us);
  }
  return 0;
}
