/*
 * lqg_hip.h — C ABI of liblqg_hip.so: MI355X (gfx950) batched LQG solve path.
 *
 * The reference (RothkopfLab/lqg) is pure Python on JAX and has NO FFI/plugin layer; its boundary for
 * this path is the Python call surface listed below.  Each entry point names the reference function it
 * replaces (paths relative to the reference checkout).  INTEGRATION.md shows the ctypes stub a
 * maintainer of the reference would add.
 *
 * Conventions
 *   - Plain C: pointers, sizes, POD structs.  No torch / HIP types in signatures (`stream` is a
 *     hipStream_t passed as void*; NULL = the default stream).
 *   - The caller owns ALL device memory (inputs, outputs, workspace).  The library never allocates,
 *     frees or retains a pointer past the call, reads NO environment variable and holds no option state:
 *     everything that selects between the library's equivalent kernels travels in the problem (`lqg_tuning`,
 *     ABI 3).  Re-entrant.  (Internal, result-neutral caches: the per-(device, kernel) record that a kernel's
 *     dynamic-LDS limit has been raised, behind a mutex.)
 *   - Calls are stream-ordered and asynchronous; the library never synchronises.
 *   - Return 0 on success; <0 invalid argument / unsupported shape (nothing launched; see
 *     lqg_last_error()); >0 a hipError_t from the launch.
 *   - NaN/inf in results are data, not errors (the reference propagates them silently).
 *   - All strides are in ELEMENTS of the problem dtype.  A batch stride of 0 shares the array between
 *     systems (parameter candidates); a time stride of 0 marks it time-invariant (what
 *     lqg/utils.py:10-35 `time_stack_spec` builds by replicating T copies).
 *   - "system" = one parameter candidate (one actor spec + one dynamics spec); "trial" = one observed
 *     trajectory.  B systems x N trials per call.
 *   - Q, Qf, R, Sigma0 are cost / covariance matrices: the library uses their symmetric part
 *     (identical to the reference for symmetric input, which is the only meaningful input).
 */
#ifndef LQG_HIP_H
#define LQG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LQG_ABI_VERSION 3

/* LQG_F32 / LQG_F64: arithmetic and storage type of every array of the problem.
 * LQG_F32_SYS64 (mixed; lqg_log_likelihood / lqg_log_likelihood_sp / lqg_workspace_bytes only; always through the operator
 * stream, whatever n_trials — the in-lane one / two-trial sweeps are single-precision paths): an fp32
 * problem — trajectories x, results ll and the internal operator stream are float — whose SPEC arrays (and Sigma0) are
 * handed over as double and whose per-system sweeps (Riccati, Kalman, moment recursion: data-independent, amortised
 * over the trials) run in fp64; the per-trial sweep stays fp32.  The operators reach the fp32 sweep rounded ONCE
 * instead of carrying the rounding of three fp32 recursions over T steps: this is what holds 1e-6 relative on the
 * log-likelihood at horizons >= 1000.  Spec strides are then in elements of double.
 * The structure-specialised libraries (lqg_log_likelihood_sp) additionally keep what that one rounding dropped from the
 * operator's Fj - I block (a second float stream in the workspace, sized by lqg_workspace_bytes) and apply hi + lo in the
 * per-trial sweep of every system whose block reaches 2.0 in magnitude, from the first such step on (point-mass models:
 * whitening gains of 10 .. 70): rounded operators leave eps32 |Fj - I| |state| of systematic error per step there. */
typedef enum lqg_dtype { LQG_F32 = 0, LQG_F64 = 1, LQG_F32_SYS64 = 2 } lqg_dtype;

/* error codes (negative return values) */
#define LQG_ERR_NULL        (-1)  /* required pointer missing            */
#define LQG_ERR_DIMS        (-2)  /* dims not instantiated in this build */
#define LQG_ERR_ARG         (-3)  /* bad size / dtype / flag             */
#define LQG_ERR_WORKSPACE   (-4)  /* workspace too small                 */

/* strided view of a [system][time][row][col] array (vectors: sc unused, scalars: sr, sc unused) */
typedef struct lqg_view {
  void*   ptr;
  int64_t sb;   /* stride between systems; 0 = shared by all systems */
  int64_t st;   /* stride between time steps; 0 = time-invariant     */
  int64_t sr;   /* row stride                                        */
  int64_t sc;   /* column stride                                     */
} lqg_view;

/* strided view of a [system][trial][time][component] array (x, mu, simulated trajectories) */
typedef struct lqg_traj {
  void*   ptr;
  int64_t sb;   /* stride between systems; 0 = all systems see the same trials */
  int64_t sn;   /* stride between trials                                        */
  int64_t st;   /* stride between time steps                                    */
  int64_t sd;   /* stride between components                                    */
} lqg_traj;

/* lqg/spec.py:5-19 LQGSpec, field for field.  Shapes per (system, time):
 * Q[b,b] q[b] Qf[b,b] qf[b] P[u,b] R[u,u] r[u] A[b,b] B[b,u] V[b,nv] F[y,b] W[y,nw]
 * (Qf, qf have no time axis: st ignored).  q, qf, P, r may have ptr == NULL meaning "all zero"
 * (what time_stack_spec builds, lqg/utils.py:29-33). */
typedef struct lqg_spec {
  lqg_view Q, q, Qf, qf, P, R, r, A, B, V, F, W;
} lqg_spec;

typedef struct lqg_dims {
  int32_t x;    /* true state dim       (System.xdim, lqg/system.py:27-33)  */
  int32_t b;    /* belief / actor dim   (System.bdim, :45-51)               */
  int32_t u;    /* action dim           (System.udim, :54-60)               */
  int32_t y;    /* observation dim      (System.ydim, :36-42)               */
  int32_t d;    /* observed dims of the data x[..., :d], d <= x (system.py:152) */
  int32_t nva, nwa;  /* columns of actor V, W    */
  int32_t nvd, nwd;  /* columns of dynamics V, W */
} lqg_dims;

/* Tuning switches that travel WITH the problem (ABI 3; rounds 1-3 read LQG_* environment variables inside the library, some
 * cached in function-local statics, so that the behaviour of a loaded library depended on when a variable had been set).
 * All zero = the library's default rules.  Nothing here changes WHAT is computed, only which of the library's equivalent
 * kernels / launch geometries computes it (results agree to rounding; tests/test_gpu_coop.py, test_gpu_trial_chunks.py,
 * test_gpu_scan.py pin every setting against the default).  The Python package fills it from `lqg_amd.options`
 * (which documents the LQG_* developer switches it reads — in ONE place, per call). */
typedef struct lqg_tuning {
  int32_t coop;                  /* strategy: 0 default rule (lane kernels where they exist), 1 cooperative kernels wherever
                                    they are supported, -1 never when lane kernels exist                              */
  int32_t trial_chunks;          /* lane per-trial sweep cut along time: 0 default rule, k > 0 exactly k chunks (clamped to
                                    chunks of >= 4 steps), -1 one pass                                                 */
  int32_t trial_chunk_waves;     /* default rule: waves to put in flight (0 = 16384)                                   */
  int32_t trial_chunk_max_waves; /* default rule: chunk only while the trials alone are at most this many waves (0 = 2048) */
  int32_t trial_chunk_tpl;       /* trials per lane of the zero-state pass: 0 rule, 1, 2                               */
  int32_t coop_trial_chunks;     /* the same cut for the row-parallel sweep of large joint dimensions: 0 rule, k, -1    */
  int32_t coop_trial_rows;       /* row-parallel per-trial sweep (k_coop_trial_rows): 0 on, -1 off (k_coop_trial)       */
  int32_t coop_sparse;           /* run-time sparsity lists of the cooperative sweeps: 0 on, -1 off                     */
  int32_t scan_lane;             /* one-launch scans of 1x1 .. 3x3 windows (k_scan_lane): 0 on, -1 per-level launches   */
  int32_t scan_rt_waves;         /* waves per window of k_scan_level_rt: 0 = 16, 8                                      */
  int32_t coop_adjoint;          /* 1: the cooperative reverse-mode sweep also for shapes that have adjoint lane kernels  */
  int32_t scan_order;            /* levels of the scans over windows of 25 .. 64: 0 rule (work-efficient order once a level
                                    holds more combines than the chip runs at once), 1 always, 2 plain Brent-Kung, -1 Hillis-Steele */
  int32_t coop_trial_tpb;        /* row-parallel per-trial sweep: most trials that share a workgroup (and its copy of the
                                    step's operator block): 0 rule, else a power of two <= 128                          */
  int32_t coop_trial_wide;       /* that sweep on 1024-thread workgroups: 0 rule (at 128 trials per workgroup), 1 always, -1 never */
  int32_t trial_lds;             /* geometry of the lane per-trial sweep of the pattern libraries: 0 rule (256-lane workgroups,
                                    two trials per lane, from 768 trials per system and 256 systems: a candidate's trials walk
                                    its operator stream together through one CU's scalar cache), -1 64-lane workgroups,
                                    1 k_trial_lds (operators staged in LDS), 2 .. 5 A/B geometries (csrc/lqg_sp_entry.hpp)  */
  int32_t hilo;                  /* LQG_F32_SYS64 through a pattern library: 0 rule (the residual stream of the operator's Fj - I
                                    block is sized by lqg_workspace_bytes and hi + lo operators serve the systems whose block
                                    reaches 2.0), -1 off: no residual stream in the workspace, operators rounded once for every
                                    system (what the generic kernels of the main library do in any case)                   */
} lqg_tuning;

typedef struct lqg_problem {
  int32_t  dtype;      /* lqg_dtype: arithmetic and storage type of every array (LQG_F32_SYS64: see above) */
  int32_t  T;          /* number of steps (System.T, lqg/system.py:17-24); data has T+1 rows */
  int64_t  n_sys;      /* B */
  int64_t  n_trials;   /* N trials per system */
  lqg_dims dims;
  lqg_spec actor;      /* all fields used                                        */
  lqg_spec dynamics;   /* only A, B, F, V, W are read (lqg/system.py:331-344)    */
  lqg_view Sigma0;     /* [b,b] initial belief covariance for the Kalman sweep; ptr NULL =
                          actor.V[0] actor.V[0]^T (lqg/system.py:79,160)          */
  double   eps;        /* eigenvalue floor of lqr.backward (lqg/control/lqr.py:16, default 1e-8) */
  void*    phase_events[4]; /* optional profiling: caller-created hipEvent_t handles (or NULL) that
                          lqg_log_likelihood / lqg_conditional_moments record on `stream`
                          [0] before the Riccati sweep, [1] after it, [2] after the forward sweep,
                          [3] after the per-trial sweep.  The library only records; the caller reads them. */
  lqg_tuning tuning;   /* all zero = default rules (see lqg_tuning) */
} lqg_problem;

int lqg_abi_version(void);
/* thread-local, valid until the next failing call on this thread */
const char* lqg_last_error(void);
/* 1 if (dtype, dims) has a compiled instantiation in this build, else 0 */
int lqg_dims_supported(int32_t dtype, const lqg_dims* dims);
/* Per kernel FAMILY: lqr.backward only depends on (b, u), kf.forward on (b, y), simulate on (x, b, u, y), the per-trial
 * sweep on (x + b, d); the fused forward sweep and the gradient on the full (x, b, u, y, d).  1 if this library holds
 * the family's instantiation for `dims` (fields the family ignores are not looked at), else 0. */
#define LQG_FAMILY_FORWARD  0
#define LQG_FAMILY_RICCATI  1
#define LQG_FAMILY_KALMAN   2
#define LQG_FAMILY_TRIAL    3
#define LQG_FAMILY_SIMULATE 4
#define LQG_FAMILY_ADJOINT  5
int lqg_kernel_supported(int32_t family, const lqg_dims* dims);
/* Two kernel strategies serve every entry point below.  LANE: one system per lane, matrices in registers, kernels
 * instantiated per model shape (lqg_kernel_supported) — the throughput path for >= 10^4 systems.  COOP: one
 * workgroup per system, matrices staged in LDS, dimensions are run-time arguments (any x, b; u, y, d <= 4) — for few
 * systems (one parameter vector x many trials) and for shapes no lane kernel holds (x + b > 20: the reference's
 * DelayedSubjectiveActor, lqg/tracking/delay.py:44-51).  lqg_strategy(p) tells which one THIS library runs for p
 * (p->tuning.coop overrides the default rule); lqg_workspace_bytes accounts for it. */
#define LQG_STRATEGY_LANE 0
#define LQG_STRATEGY_COOP 1
int lqg_coop_supported(const lqg_dims* dims);
int lqg_strategy(const lqg_problem* p);
/* name of the code object target this library was compiled for ("gfx950") */
const char* lqg_target_arch(void);

/* Replaces lqg.control.lqr.backward(spec, eps) -> Gains(L, l, H)   [lqg/control/lqr.py:16-42]
 * Reads p->actor.  L[B,T,u,b], l[B,T,u], H[B,T,u,u] (H = regularised Ht) in forward time order.
 * Any of l, H may have ptr NULL (not written). */
int lqg_riccati_backward(const lqg_problem* p, lqg_view L, lqg_view l, lqg_view H, void* stream);

/* Replaces lqg.belief.kf.forward(spec, Sigma0) -> K               [lqg/belief/kf.py:6-21]
 * Reads p->actor (A, F, V, W) and p->Sigma0.  K[B,T,b,y]. */
int lqg_kalman_forward(const lqg_problem* p, lqg_view K, void* stream);

/* Bytes of caller-provided device workspace the call `op` needs for problem p (with n_trials == 1 every op runs
 * fused and needs only the gain scratch; otherwise also the per-system operator stream). */
#define LQG_OP_LOG_LIKELIHOOD      0
#define LQG_OP_CONDITIONAL_MOMENTS 1
size_t lqg_workspace_bytes(const lqg_problem* p, int32_t op);

/* Replaces System.conditional_moments vmapped over trials         [lqg/system.py:142-235, :241]
 * x[B,N,T+1,d] -> mu[B,N,T,m] (m = x+b), Sigma[B,T,m,m].  Sigma does not depend on the data
 * (only mu does), so it is emitted once per system; the reference returns N identical copies.
 * mu or Sigma may have ptr NULL (not written). */
int lqg_conditional_moments(const lqg_problem* p, lqg_traj x, lqg_traj mu, lqg_view Sigma,
                            void* workspace, size_t workspace_bytes, void* stream);

/* Replaces System.log_likelihood(x) [lqg/system.py:246-248] (= conditional_distribution(x)
 * .log_prob(x[:,1:]), :237-244, numpyro MultivariateNormal): the fused hot path
 * Riccati -> Kalman -> joint system -> moment recursion -> Gaussian log-density summed over time.
 * x[B,N,T+1,d] -> ll[b*ll_sb + n*ll_sn], one value per (system, trial), dtype of the problem. */
int lqg_log_likelihood(const lqg_problem* p, lqg_traj x, void* ll, int64_t ll_sb, int64_t ll_sn,
                       void* workspace, size_t workspace_bytes, void* stream);

/* TIME-PARALLEL twins of the two entry points above (lqg_amd/csrc/lqg_scan.hpp): the Riccati recursion of lqr.py:16-42, the
 * Kalman recursion of kf.py:6-21 and the moment recursion of system.py:209-235 are evaluated as ASSOCIATIVE SCANS over the
 * time axis — log2(T) dependent combines of (A, C, J) window elements, one wave per window, instead of T dependent steps
 * — then the same per-trial sweep.  For few systems with many steps (one parameter vector x many trials).  Same
 * arguments, same results to rounding (fp64 arithmetic inside, whatever the problem dtype).  Preconditions the CALLER
 * guarantees: the eigenvalue floor of lqr.py:27-28 is inactive (lambda_min(R) >= eps, Q, Qf >= 0 suffice); checked here:
 * no affine cost terms (q, qf, P, r NULL), u, y, d <= 4, y <= b, T >= 2, and x + b <= 24 (windows in LDS) or b <= 64,
 * x + b - d <= 64 with the per-step working sets within 160 KB of LDS (windows in registers: the delay-augmented models
 * of lqg/tracking/delay.py, b = 39, x + b = 65) (lqg_scan_supported).
 * Workspace: lqg_scan_workspace_bytes(p). */
int lqg_scan_supported(const lqg_problem* p);
size_t lqg_scan_workspace_bytes(const lqg_problem* p);
int lqg_log_likelihood_scan(const lqg_problem* p, lqg_traj x, void* ll, int64_t ll_sb, int64_t ll_sn,
                            void* workspace, size_t workspace_bytes, void* stream);
int lqg_conditional_moments_scan(const lqg_problem* p, lqg_traj x, lqg_traj mu, lqg_view Sigma,
                                 void* workspace, size_t workspace_bytes, void* stream);

/* The per-trial sweep of a structure-specialised library (csrc/lqg_sp_entry.hpp: `lqg_trial_sweep_sp`, the operator's
 * structural zeros compiled out) over an operator stream produced elsewhere.  `ops` points at the stream
 * [n_sys][T+1][ops_reals]; the scratch of the time-chunked sweep follows it (csrc/lqg_trial_chunk.hpp).
 * Stream format (ABI 2), per (system, step), m = x + b, o = d, r = m - o:  (Fj - I)[m,m] row-major | U2[r,o] | Li (lower
 * triangle of chol(Sigma_oo)^-1, packed by rows) | half log-det + o/2 log(2 pi), padded to a multiple of 4 reals.  The
 * mean state is (dO, muR) with observed mean = x_{t-1} + dO:  w = Li ((x_t - x_{t-1}) - dO),  c = muR + U2 w,
 * [dO' ; muR'] = [0 ; c] + (Fj - I) [x_t ; c].  The identity is taken off Fj BEFORE rounding (ABI 1 kept it on the
 * unobserved rows): fl(F_ii) ~ 1 loses the digits of F_ii - 1 that integrate the mean over the horizon. */
typedef int (*lqg_trial_sweep_fn)(const lqg_problem* p, lqg_traj x, void* ll, int64_t ll_sb, int64_t ll_sn,
                                  const void* ops, void* stream);
/* lqg_log_likelihood_scan with the per-trial sweep delegated to `trial_sweep` (NULL: the library's own kernels; a
 * non-zero return of the delegate also falls back to them). */
int lqg_log_likelihood_scan_with(const lqg_problem* p, lqg_traj x, void* ll, int64_t ll_sb, int64_t ll_sn,
                                 void* workspace, size_t workspace_bytes, void* stream, lqg_trial_sweep_fn trial_sweep);

/* Structure-specialised twin (NOT in liblqg_hip.so): libraries generated per sparsity pattern by
 * lqg_amd/specialize.py (lqg_amd/csrc/pat/pat_<hash>.so, kernels in csrc/lqg_kernels_sp.hpp) export
 *     int lqg_log_likelihood_sp(<exactly the argument list of lqg_log_likelihood>);
 * with the same contract, restricted to time-invariant specs without affine terms and to the dims / pattern
 * they were compiled for (anything else returns LQG_ERR_ARG / LQG_ERR_DIMS without launching).  They are an
 * optimisation layer: results are identical to lqg_log_likelihood's up to rounding, and every caller falls back to it.
 * The same libraries export the materialising pass for specs that may vary in TIME as long as they keep the pattern (a model
 * whose entries move: the masks are then the union over systems and steps, specialize.pattern_of_time_varying):
 *     int lqg_solve_materialised_sp(<the argument list of lqg_solve_materialised without ll_sn>);
 * one trial per system, no affine cost terms; every output view optional; the materialised L doubles as the forward sweep's
 * gain stream.  Mode M2 of bench_m2.py runs on it (csrc/lqg_kernels_sp.hpp: k_riccati_tv_sp, k_forward_tv_sp). */

/* Everything the reference's path materialises, in ONE pass (two or three kernels instead of the seven that separate
 * lqr.backward + kf.forward + conditional_moments + log_likelihood calls launch — each of which recomputes the gains
 * inside, as the reference's own functions do, lqg/system.py:157-161):  L, l, H (lqr.py:42), K (kf.py:21), mu, Sigma
 * (system.py:235) and ll (system.py:248).  Every output view is optional (ptr NULL = not written).
 * With genuinely time-varying specs in and all outputs on this is SURVEY.md §8(d) mode M2, the HBM-bound regime. */
int lqg_solve_materialised(const lqg_problem* p, lqg_traj x, lqg_view L, lqg_view l, lqg_view H, lqg_view K,
                           lqg_traj mu, lqg_view Sigma, void* ll, int64_t ll_sb, int64_t ll_sn, void* workspace,
                           size_t workspace_bytes, void* stream);

/* Objective of lqg_model / the candidate sweep [lqg/infer/models.py:34, notebooks/Tutorial.ipynb
 * cell 38 `.log_likelihood(x).sum()`]: out[b] = sum_n ll[b,n], accumulated and stored in fp64
 * regardless of the problem dtype (so the cross-GPU all-reduce is order-insensitive to ~1e-15). */
int lqg_sum_trials(int32_t dtype, const void* ll, int64_t n_sys, int64_t n_trials, int64_t ll_sb,
                   int64_t ll_sn, double* out, void* workspace, size_t workspace_bytes, void* stream);
/* workspace for lqg_sum_trials (0 when n_trials fits one reduction chunk): partial sums of a fixed two-stage tree,
 * so that the result is bitwise reproducible run to run (no atomics) */
size_t lqg_sum_trials_workspace_bytes(int64_t n_sys, int64_t n_trials);

/* Model-zoo setup of lqg/tracking/point_mass.py:113-127 (`point_mass_dynamics_matrices`): per candidate i the zero-order-hold
 * discretisation (A[i] 3x3, B[i] 3x1; :50-79), the Van Loan process-noise block (:82-110) made positive definite by
 * eigenvalue clipping at psd_eps (:130-144) and its UPPER Cholesky factor V[i] 3x3 (:123).  fp64, contiguous [n] inputs,
 * row-major outputs.  Replaces torch.linalg.matrix_exp / eigh / cholesky on 3x3..6x6 matrices (host-synchronising
 * rocSOLVER calls) when no gradient is asked for. */
int lqg_point_mass_setup(int64_t n, const double* damping, const double* mass, const double* tau,
                         const double* action_variability, double dt, double psd_eps, double* A, double* B, double* V,
                         void* stream);

/* Device-side check of the value-dependent preconditions of the exact shortcuts (block decoupling, lqg_amd/decouple.py;
 * the time-parallel entries above): the eigenvalue floor of lqr.py:27-28 can never be active (Gershgorin lower bound of
 * sym(R) >= p->eps and of sym(Q), sym(Qf) >= 0 for every system and step) and, with check_cond != 0, the Gershgorin
 * condition bound of (V V')[:d, :d] of the dynamics is <= max_cond.  ok[0] (device memory) = 1 when all hold, else 0.
 * Conservative (Gershgorin), elementwise, stream-ordered, no synchronisation: usable inside a captured hipGraph whose
 * frozen decisions rest on these preconditions (lqg_amd/infer/graphed.py poisons its result with NaN when ok == 0). */
int lqg_precondition_flags(const lqg_problem* p, double max_cond, int32_t check_cond, int32_t* ok, void* stream);

/* Central differences around the candidate sweep, for callers that replay one evaluation as a hipGraph (the reference takes
 * jax.grad of the summed log-likelihood: lqg/infer/models.py:34, lqg/optim.py:142-147).  All pointers are device memory.
 * lqg_fd_candidates: z[K, P] (fp64 log-parameters) -> flat[K (2 P + 1), F] in `dtype` (LQG_F32 / LQG_F64), the flattened
 *   specs of the centre and of the +-h perturbations of every point through the affine map theta -> base[F] + theta D[P, F]
 *   (theta = exp(z)); candidate c = k (2 P + 1) + j with j = 0 the centre, 1 + p: +h on parameter p, 1 + P + p: -h.
 * lqg_fd_combine: obj[K (2 P + 1)] (the summed log-likelihoods of those candidates) -> out[K, 1 + P] = value and the
 *   central-difference gradient w.r.t. z; ok (NULL or the flag of lqg_precondition_flags): *ok == 0 turns out into NaN. */
int lqg_fd_candidates(const double* z, const double* base, const double* D, void* flat, int32_t dtype, int64_t K, int32_t P,
                      int64_t F, double h, void* stream);
int lqg_fd_combine(const double* obj, const int32_t* ok, double* out, int64_t K, int32_t P, double h, void* stream);

/* Replaces the per-trial scan of System.simulate [lqg/system.py:106-128] with the standard-normal
 * draws supplied by the caller (the reference draws them from jax.random, :102-105).
 * gains L[B,T,u,b], l[B,T,u] (l.ptr NULL = 0), K[B,T,b,y] as produced by the two calls above;
 * eps[B,N,T,xdim], eta[B,N,T,ydim]; x0[B,xdim], xhat0[B,bdim] views (ptr NULL = zeros; only sb, sr used).
 * Outputs xs[B,N,T+1,x] (row 0 = x0), and optionally xhat[B,N,T+1,b], ys[B,N,T,y], us[B,N,T,u]. */
int lqg_simulate(const lqg_problem* p, lqg_view L, lqg_view l, lqg_view K, lqg_traj eps, lqg_traj eta,
                 lqg_view x0, lqg_view xhat0, lqg_traj xs, lqg_traj xhat, lqg_traj ys, lqg_traj us,
                 void* stream);

/* lqg_simulate with the draws of lqg/system.py:100-105 (eps ~ N(0, I), eta ~ N(0, I) per (trial, step), jax.random in the
 * reference) made IN THE KERNEL: counter-based Philox4x32-10 keyed by `seed`, counter = (trial, system, step, block)
 * (ABI 3; indices below 2^32), Box-Muller normals evaluated in fp32 and widened for an fp64 problem (csrc/lqg_rng.hpp: an
 * fp64 simulation draws 24-bit normals).  The trajectory of trial n of system s is a pure function of (seed, s, n): it does
 * not depend on how many systems or trials the call holds, on which kernel serves the shape, or on the mapping to lanes.
 * Systems and trials are numbered WITHIN the call: ranks that simulate disjoint shards use distinct seeds.
 * Nothing but the trajectories crosses HBM (lqg_simulate reads 2 T (x + y) reals per trial it was handed). */
int lqg_simulate_rng(const lqg_problem* p, lqg_view L, lqg_view l, lqg_view K, uint64_t seed, lqg_view x0,
                     lqg_view xhat0, lqg_traj xs, lqg_traj xhat, lqg_traj ys, lqg_traj us, void* stream);

/* Replaces numpyro MultivariateNormal(mu, Sigma).to_event(1).log_prob(value) as used by
 * System.conditional_distribution [lqg/system.py:244,248]: value[B,N,T,k], mu[B,N,T,k] (k = dims.d
 * components read from each), Sigma[B,T,m,m] of which the leading k x k block is used;
 * out[b*sb + n*sn] = sum_t log N(value_t; mu_t, Sigma_t). */
int lqg_gaussian_logprob(int32_t dtype, int32_t k, int32_t T, int64_t n_sys, int64_t n_trials,
                         lqg_traj value, lqg_traj mu, lqg_view Sigma, void* out, int64_t out_sb,
                         int64_t out_sn, void* stream);

/* ---- reverse-mode gradient of the log-likelihood -------------------------------------------------------------------
 * Replaces jax.grad / jax.value_and_grad of System.log_likelihood, which every inference driver of the reference takes
 * through all three scans [lqg/optim.py:142-147 `jit(grad(fun))`, lqg/infer/utils.py:18,37-39, lqg/infer/mle.py:17-23,
 * notebooks/Tutorial.ipynb cell 40 `grad(ll)`]:  for the objective  J = sum_{s,n} g[s,n] * ll[s,n]  it returns
 * dJ/d(spec matrix) PER (system, trial) pair — the caller sums over the trial axis.  With TIME-INVARIANT specs (every
 * field's st == 0; the whole model zoo) there is ONE accumulated bar per matrix; if any field varies over time the bars
 * are per step (lqg_grad_slabs(p) == T slabs instead of 1).  Gradients are taken w.r.t. the dynamics' A, B, F, VV = V V', WW = W W' and the
 * actor's A, B, F, VV, WW, Q, R, Qf (+ Sigma0 when given); chain VVbar to V as (VVbar + VVbar') V.  q, r, qf, P get none
 * (the likelihood ignores the affine gain).  Bars of symmetric quantities (VV, WW, Q, Qf, Sigma0) are symmetrised;
 * the eigenvalue-floor shift of lqr.py:27-28 is held constant (it is 0 whenever R + B'SB > eps).
 *
 * grad: [lqg_grad_slabs(p)][lqg_grad_elements(dims)][ld] reals of the problem dtype, ld >= n_sys * n_trials
 * (lane = s * n_trials + n); slab t holds the bars of the step-t matrices (slab 0 also Qf and Sigma0, which have no time
 * axis); within a slab, consecutive row-major matrices in this order:
 *   dyn A[x,x] B[x,u] F[y,x] VV[x,x] WW[y,y] | actor A[b,b] B[b,u] F[y,b] VV[b,b] WW[y,y] Q[b,b] R[u,u] Qf[b,b] Sigma0[b,b]
 *   | actor A[b,b], B[b,u] (second part: ADD to the first; the Riccati sweep writes it separately)
 * With Sigma0.ptr NULL (default V V') the Sigma0 bar is already folded into the actor's VV bar.
 * g: upstream weights (NULL = all 1), element [s*g_sb + n*g_sn]; ll (optional): the value, as lqg_log_likelihood.
 * workspace: lqg_grad_workspace_bytes(p, ld) bytes (the kept forward state, [T][per-step reals][ld]).
 * phases: 1 = forward sweeps only (fill the workspace, write ll — what an autograd forward() runs), 2 = reverse sweeps
 * only (consume the workspace a phase-1 call with the same problem, x and ld left, write grad — backward(), where g
 * first becomes known), 3 = both in one call. */
int lqg_grad_supported(int32_t dtype, const lqg_dims* dims);
int64_t lqg_grad_elements(const lqg_dims* dims);
int32_t lqg_grad_slabs(const lqg_problem* p);
size_t lqg_grad_workspace_bytes(const lqg_problem* p, int64_t ld);
/* Lanes of the gradient array per system: n_trials for shapes with adjoint LANE kernels (one lane per (system, trial) pair,
 * the caller sums over the trials), 1 for every other supported shape (fp64; x, b <= 64; u, y, d <= 4: the cooperative
 * sweep of csrc/lqg_coop_adjoint.hip — one workgroup per system — returns the bars ALREADY summed over the trials with the
 * weights g; then ld >= n_sys and lane = system).  lqg_grad_workspace_bytes accounts for either. */
int32_t lqg_grad_lanes_per_system(const lqg_problem* p);
/* Structure-specialised twin (NOT in liblqg_hip.so; round 5): the adjoint libraries generated per sparsity pattern by
 * lqg_amd/specialize.py (lqg_amd/csrc/pat/padj_<key>.so; kernels in csrc/lqg_adjoint_sp.hpp, lqg_adjoint_trial_sp.hpp) export
 *     int    lqg_log_likelihood_grad_sp(<exactly the argument list of lqg_log_likelihood_grad>);
 *     size_t lqg_grad_workspace_bytes_sp(const lqg_problem* p);
 * for time-invariant specs without affine cost terms and the dims / pattern they were compiled for (anything else returns
 * LQG_ERR_ARG / LQG_ERR_DIMS without launching).  The sweep is cut like the forward path: Riccati, Kalman, joint system, moment
 * recursion AND their adjoints run once per SYSTEM (one lane each); the trials enter through per-step trial sums formed by a
 * per-trial mu-bar sweep over the operator stream (1 or 2 trials: in the system's lane); nothing per step is parked in HBM but
 * checkpoints every few steps.  The bars come back ALREADY SUMMED over the trials with the weights g: grad is
 * [lqg_grad_elements(dims)][ld] with ld >= n_sys, lane = system (as lqg_grad_lanes_per_system == 1); same element order; bars of
 * fields that no model parameter moves (the pattern's live flags) are written as zeros, bars of structurally zero entries are
 * not formed.  phases as above: a phases = 2 call records phase_events [0] before the per-trial reverse sweep, [1] after it,
 * [2] after the system reverse sweep, [3] after the Riccati adjoint. */
int lqg_log_likelihood_grad(const lqg_problem* p, lqg_traj x, const void* g, int64_t g_sb, int64_t g_sn, void* ll,
                            int64_t ll_sb, int64_t ll_sn, void* grad, int64_t ld, void* workspace,
                            size_t workspace_bytes, int32_t phases, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LQG_HIP_H */
