// lqg_abi.hip — the C ABI of liblqg_hip.so (include/lqg_hip.h): argument checking, dispatch on
// (dtype, dims) to the compiled kernel instantiations, stream-ordered launches.  No allocation, no
// synchronisation, no global mutable state (the error string is thread-local).
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/lqg_hip.h"
#include "lqg_adjoint_launch.hpp"
#include "lqg_coop_launch.hpp"
#include "lqg_launch.hpp"
// the instantiation lists: lqg_dims.def for the main library; lqg_amd/build.py compiles auxiliary libraries for
// shapes that are not listed there with -DLQG_DIMS_DEF="<generated single-shape lists>"
#ifndef LQG_DIMS_DEF
#define LQG_DIMS_DEF "lqg_dims.def"
#endif
#include LQG_DIMS_DEF

// ---------------------------------------------------------------- extern instantiations (defined by lqg_inst.hip)
#define X(B_, U_)                                                                                              \
  extern template hipError_t lqg::host::launch_riccati<float, B_, U_>(const lqg_problem*, lqg_view, lqg_view, \
                                                                       lqg_view, void*, long, hipStream_t);    \
  extern template hipError_t lqg::host::launch_riccati<double, B_, U_>(const lqg_problem*, lqg_view, lqg_view, \
                                                                        lqg_view, void*, long, hipStream_t);
LQG_RICCATI_DIMS(X)
#undef X
#define X(B_, Y_)                                                                                          \
  extern template hipError_t lqg::host::launch_kalman<float, B_, Y_>(const lqg_problem*, lqg_view, hipStream_t); \
  extern template hipError_t lqg::host::launch_kalman<double, B_, Y_>(const lqg_problem*, lqg_view, hipStream_t);
LQG_KALMAN_DIMS(X)
#undef X
#define X(X_, B_, U_, Y_, D_)                                                                                   \
  extern template hipError_t lqg::host::launch_forward<float, X_, B_, U_, Y_, D_>(                              \
      const lqg_problem*, const void*, long, bool, lqg_traj, void*, long, void*, lqg_view, lqg_traj, lqg_view, \
      hipStream_t);                                                                                             \
  extern template hipError_t lqg::host::launch_forward<double, X_, B_, U_, Y_, D_>(                             \
      const lqg_problem*, const void*, long, bool, lqg_traj, void*, long, void*, lqg_view, lqg_traj, lqg_view, \
      hipStream_t);
LQG_FORWARD_DIMS(X)
#undef X
#define X(X_, B_, U_, Y_, D_)                                                                                   \
  extern template hipError_t lqg::host::launch_forward_ops32<X_, B_, U_, Y_, D_>(const lqg_problem*, const void*, long, \
                                                                                  void*, hipStream_t);
LQG_FORWARD_DIMS(X)
#undef X
#define X(M_, D_)                                                                                              \
  extern template hipError_t lqg::host::launch_trial<float, M_, D_>(const lqg_problem*, const void*, lqg_traj, \
                                                                     lqg_traj, void*, long, long, hipStream_t); \
  extern template hipError_t lqg::host::launch_trial<double, M_, D_>(const lqg_problem*, const void*, lqg_traj, \
                                                                      lqg_traj, void*, long, long, hipStream_t);
LQG_TRIAL_DIMS(X)
#undef X
#define X(X_, B_, U_, Y_)                                                                                       \
  extern template hipError_t lqg::host::launch_simulate<float, X_, B_, U_, Y_>(                                 \
      const lqg_problem*, lqg_view, lqg_view, lqg_view, lqg_traj, lqg_traj, lqg_view, lqg_view, lqg_traj,       \
      lqg_traj, lqg_traj, lqg_traj, hipStream_t, unsigned long long);                                           \
  extern template hipError_t lqg::host::launch_simulate<double, X_, B_, U_, Y_>(                                \
      const lqg_problem*, lqg_view, lqg_view, lqg_view, lqg_traj, lqg_traj, lqg_view, lqg_view, lqg_traj,       \
      lqg_traj, lqg_traj, lqg_traj, hipStream_t, unsigned long long);
LQG_SIM_DIMS(X)
#undef X

#ifndef LQG_ADJOINT_DIMS
#define LQG_ADJOINT_DIMS(X)
#endif
#define X(X_, B_, U_, Y_, D_)                                                                                    \
  extern template hipError_t lqg::host::launch_adjoint<float, X_, B_, U_, Y_, D_>(                               \
      const lqg_problem*, lqg_traj, const void*, long, long, void*, long, long, void*, long, void*, int, hipStream_t); \
  extern template hipError_t lqg::host::launch_adjoint<double, X_, B_, U_, Y_, D_>(                              \
      const lqg_problem*, lqg_traj, const void*, long, long, void*, long, long, void*, long, void*, int, hipStream_t);
LQG_ADJOINT_DIMS(X)
#undef X

using namespace lqg::host;

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

template <typename R>
hipError_t dispatch_riccati(const lqg_problem* p, lqg_view L, lqg_view l, lqg_view H, void* Ls, long ldb,
                            hipStream_t st, bool* found) {
  *found = true;
#define X(B_, U_) \
  if (p->dims.b == B_ && p->dims.u == U_) return launch_riccati<R, B_, U_>(p, L, l, H, Ls, ldb, st);
  LQG_RICCATI_DIMS(X)
#undef X
  *found = false;
  return hipSuccess;
}
template <typename R>
hipError_t dispatch_kalman(const lqg_problem* p, lqg_view K, hipStream_t st, bool* found) {
  *found = true;
#define X(B_, Y_) \
  if (p->dims.b == B_ && p->dims.y == Y_) return launch_kalman<R, B_, Y_>(p, K, st);
  LQG_KALMAN_DIMS(X)
#undef X
  *found = false;
  return hipSuccess;
}
template <typename R>
hipError_t dispatch_forward(const lqg_problem* p, const void* Ls, long ldb, bool fused, lqg_traj x, void* ll,
                            long ll_sb, void* ops, lqg_view Sig, lqg_traj mu, lqg_view Kout, hipStream_t st,
                            bool* found) {
  *found = true;
  const lqg_dims& d = p->dims;
#define X(X_, B_, U_, Y_, D_)                                                  \
  if (d.x == X_ && d.b == B_ && d.u == U_ && d.y == Y_ && d.d == D_)          \
    return launch_forward<R, X_, B_, U_, Y_, D_>(p, Ls, ldb, fused, x, ll, ll_sb, ops, Sig, mu, Kout, st);
  LQG_FORWARD_DIMS(X)
#undef X
  *found = false;
  return hipSuccess;
}
hipError_t dispatch_forward_ops32(const lqg_problem* p, const void* Ls, long ldb, void* ops, hipStream_t st, bool* found) {
  *found = true;
  const lqg_dims& d = p->dims;
#define X(X_, B_, U_, Y_, D_)                                                  \
  if (d.x == X_ && d.b == B_ && d.u == U_ && d.y == Y_ && d.d == D_)          \
    return launch_forward_ops32<X_, B_, U_, Y_, D_>(p, Ls, ldb, ops, st);
  LQG_FORWARD_DIMS(X)
#undef X
  *found = false;
  return hipSuccess;
}
template <typename R>
hipError_t dispatch_trial(const lqg_problem* p, const void* ops, lqg_traj x, lqg_traj mu, void* ll, long ll_sb,
                          long ll_sn, hipStream_t st, bool* found) {
  *found = true;
  const int m = p->dims.x + p->dims.b;
#define X(M_, D_) \
  if (m == M_ && p->dims.d == D_) return launch_trial<R, M_, D_>(p, ops, x, mu, ll, ll_sb, ll_sn, st);
  LQG_TRIAL_DIMS(X)
#undef X
  *found = false;
  return hipSuccess;
}
template <typename R>
hipError_t dispatch_simulate(const lqg_problem* p, lqg_view L, lqg_view l, lqg_view K, lqg_traj eps, lqg_traj eta,
                             lqg_view x0, lqg_view xhat0, lqg_traj xs, lqg_traj xhat, lqg_traj ys, lqg_traj us,
                             hipStream_t st, bool* found, unsigned long long seed = 0) {
  *found = true;
  const lqg_dims& d = p->dims;
#define X(X_, B_, U_, Y_)                                     \
  if (d.x == X_ && d.b == B_ && d.u == U_ && d.y == Y_)      \
    return launch_simulate<R, X_, B_, U_, Y_>(p, L, l, K, eps, eta, x0, xhat0, xs, xhat, ys, us, st, seed);
  LQG_SIM_DIMS(X)
#undef X
  *found = false;
  return hipSuccess;
}

bool has_forward(const lqg_dims& d) {
#define X(X_, B_, U_, Y_, D_) \
  if (d.x == X_ && d.b == B_ && d.u == U_ && d.y == Y_ && d.d == D_) return true;
  LQG_FORWARD_DIMS(X)
#undef X
  return false;
}

// ---- strategy: lane-per-system kernels (register-resident, instantiated per shape) or the cooperative
// workgroup-per-system kernels (lqg_coop.hpp: run-time dims, LDS-staged).  LQG_COOP=1 forces the cooperative path
// wherever it is defined, LQG_COOP=0 restricts it to shapes without a lane instantiation (the default).  Measured
// (profiles/r02_b_configs24_*.jsonl, one system, T=500 / 1000): the cooperative forward sweep beats the single-lane one
// 1.5x at m=8 (0.84 vs 1.26 ms) but its Riccati sweep is 2x slower (0.31 vs 0.14 ms) and at m=10 with two merged
// components it loses overall — so LQG_COOP_MAX_SYS (systems per call up to which a shape WITH lane kernels takes the
// cooperative path) defaults to 0; build with -DLQG_COOP_MAX_SYS=512 or set LQG_COOP=1 to turn it on.
#ifndef LQG_COOP_MAX_SYS
#define LQG_COOP_MAX_SYS 0
#endif
#ifndef LQG_COOP_MIN_M
#define LQG_COOP_MIN_M 8
#endif
bool use_coop(const lqg_problem* p) {
  if (!coop_supported(p->dims)) return false;
  if (!has_forward(p->dims)) return true;
  if (p->tuning.coop != 0) return p->tuning.coop > 0;            // (include/lqg_hip.h: lqg_tuning)
  return p->n_sys <= LQG_COOP_MAX_SYS && p->dims.x + p->dims.b >= LQG_COOP_MIN_M;
}

// mixed: the entry point serves LQG_F32_SYS64 (include/lqg_hip.h) — every other one reads all arrays in ONE type
int check_problem(const lqg_problem* p, const char* who, bool mixed = false) {
  if (!p) return fail(LQG_ERR_NULL, "%s: problem is NULL", who);
  if (p->dtype == LQG_F32_SYS64 && !mixed)
    return fail(LQG_ERR_ARG, "%s: dtype LQG_F32_SYS64 is served by lqg_log_likelihood only", who);
  if (p->dtype != LQG_F32 && p->dtype != LQG_F64 && p->dtype != LQG_F32_SYS64)
    return fail(LQG_ERR_ARG, "%s: bad dtype %d", who, p->dtype);
  if (p->T < 1) return fail(LQG_ERR_ARG, "%s: T=%d < 1", who, p->T);
  if (p->n_sys < 0 || p->n_trials < 0) return fail(LQG_ERR_ARG, "%s: negative batch", who);
  const lqg_dims& d = p->dims;
  if (d.x < 1 || d.b < 1 || d.u < 1 || d.y < 1) return fail(LQG_ERR_DIMS, "%s: non-positive dims", who);
  // lqg_tuning (include/lqg_hip.h): enumerated values only — an out-of-range value must not reach a kernel's launch geometry
  // (coop_trial_tpb > 128 made k_coop_trial_rows divide by RT = BLOCK / tpb = 0)
  const lqg_tuning& tn = p->tuning;
  auto tri = [](int v) { return v >= -1 && v <= 1; };
  const int tpb = tn.coop_trial_tpb;
  if (!tri(tn.coop) || !tri(tn.coop_trial_rows) || !tri(tn.coop_sparse) || !tri(tn.scan_lane) || !tri(tn.coop_trial_wide) ||
      tn.trial_lds < -1 || tn.trial_lds > 5 || tn.coop_adjoint < 0 || tn.coop_adjoint > 1 || tn.scan_order < -1 || tn.scan_order > 2 ||
      (tn.scan_rt_waves != 0 && tn.scan_rt_waves != 8 && tn.scan_rt_waves != 16) || tn.trial_chunks < -1 ||
      tn.coop_trial_chunks < -1 || tn.trial_chunk_waves < 0 || tn.trial_chunk_max_waves < 0 || tn.trial_chunk_tpl < 0 ||
      tn.trial_chunk_tpl > 2 || tpb < 0 || tpb > 128 || (tpb & (tpb - 1)) != 0 || tn.hilo < -1 || tn.hilo > 0)
    return fail(LQG_ERR_ARG, "%s: lqg_tuning holds a value outside its documented range (include/lqg_hip.h)", who);
  return 0;
}
int need(const lqg_view& v, const char* who, const char* name) {
  return v.ptr ? 0 : fail(LQG_ERR_NULL, "%s: %s.ptr is NULL", who, name);
}
int unsupported(const lqg_problem* p, const char* who) {
  const lqg_dims& d = p->dims;
  return fail(LQG_ERR_DIMS,
              "%s: no kernel instantiation for dims (x=%d,b=%d,u=%d,y=%d,d=%d) in this library; lqg_amd compiles an "
              "auxiliary library for new shapes on demand (lqg_amd.build.build_dims_library), or add the shape to "
              "lqg_amd/csrc/lqg_dims.def and rebuild",
              who, d.x, d.b, d.u, d.y, d.d);
}
int done(hipError_t e, const char* who) {
  if (e == hipSuccess) return 0;
  fail((int)e, "%s: %s", who, hipGetErrorString(e));
  return (int)e;
}

int check_full(const lqg_problem* p, const char* who, bool mixed = false) {
  if (int rc = check_problem(p, who, mixed)) return rc;
  const lqg_spec& a = p->actor;
  const lqg_spec& d = p->dynamics;
  if (need(a.Q, who, "actor.Q") || need(a.Qf, who, "actor.Qf") || need(a.R, who, "actor.R") ||
      need(a.A, who, "actor.A") || need(a.B, who, "actor.B") || need(a.F, who, "actor.F") ||
      need(a.V, who, "actor.V") || need(a.W, who, "actor.W") || need(d.A, who, "dynamics.A") ||
      need(d.B, who, "dynamics.B") || need(d.F, who, "dynamics.F") || need(d.V, who, "dynamics.V") ||
      need(d.W, who, "dynamics.W"))
    return LQG_ERR_NULL;
  if (p->dims.d < 1 || p->dims.d > p->dims.x)
    return fail(LQG_ERR_DIMS, "%s: observed dims d=%d must be in [1, x=%d]", who, p->dims.d, p->dims.x);
  if (!has_forward(p->dims) && !coop_supported(p->dims)) return unsupported(p, who);
  return 0;
}

struct GainOutputs {
  lqg_view L, l, H, K;
};

// Shared driver of lqg_log_likelihood / lqg_conditional_moments / lqg_solve_materialised.
// One trial per system runs FUSED (the trial is swept in-lane by k_forward, no operator stream, no k_trial);
// more trials go through the per-system operator stream and k_trial.
template <typename R>
int run_moments(const lqg_problem* p, lqg_traj x, lqg_traj mu, lqg_view Sigma, void* ll, long ll_sb, long ll_sn,
                const GainOutputs& g, void* workspace, size_t workspace_bytes, hipStream_t st, const char* who);

template <typename R>
int run_coop(const lqg_problem* p, lqg_traj x, lqg_traj mu, lqg_view Sigma, void* ll, long ll_sb, long ll_sn,
             const GainOutputs& g, void* workspace, size_t workspace_bytes, hipStream_t st, const char* who);

template <typename R>
int run_moments(const lqg_problem* p, lqg_traj x, lqg_traj mu, lqg_view Sigma, void* ll, long ll_sb, long ll_sn,
                const GainOutputs& g, void* workspace, size_t workspace_bytes, hipStream_t st, const char* who) {
  if (use_coop(p)) return run_coop<R>(p, x, mu, Sigma, ll, ll_sb, ll_sn, g, workspace, workspace_bytes, st, who);
  const bool fused = p->n_trials == 1;
  const Workspace w = carve(p, !fused);
  if (!workspace || workspace_bytes < w.total)
    return fail(LQG_ERR_WORKSPACE, "%s: workspace %zu B < required %zu B", who, workspace_bytes, w.total);
  char* base = static_cast<char*>(workspace);
  void* Ls = base + w.ls_off;
  void* ops = fused ? nullptr : base + w.ops_off;
  bool found;
  auto mark = [&](int i) {
    if (p->phase_events[i]) (void)hipEventRecord(static_cast<hipEvent_t>(p->phase_events[i]), st);
  };
  mark(0);
  hipError_t e = dispatch_riccati<R>(p, g.L, g.l, g.H, Ls, w.ldb, st, &found);
  if (!found) return unsupported(p, who);
  if (e != hipSuccess) return done(e, who);
  mark(1);
  const lqg_traj no_mu{nullptr, 0, 0, 0, 0};
  e = dispatch_forward<R>(p, Ls, w.ldb, fused, x, ll, ll_sb, ops, Sigma, fused ? mu : no_mu, g.K, st, &found);
  if (!found) return unsupported(p, who);
  if (e != hipSuccess) return done(e, who);
  mark(2);
  if (!fused && p->n_trials > 0 && (ll || mu.ptr)) {
    e = dispatch_trial<R>(p, ops, x, mu, ll, ll_sb, ll_sn, st, &found);
    if (!found) return unsupported(p, who);
  }
  mark(3);
  return done(e, who);
}

// LQG_F32_SYS64: Riccati and forward sweeps in fp64 over the double spec arrays, operator stream rounded to float on the
// way out, per-trial sweep in fp32 (lane kernels only; time-invariant or not, affine terms or not).
int run_mixed(const lqg_problem* p, lqg_traj x, void* ll, long ll_sb, long ll_sn, void* workspace, size_t workspace_bytes,
              hipStream_t st, const char* who) {
  if (p->n_trials < 1) return fail(LQG_ERR_ARG, "%s: LQG_F32_SYS64 needs n_trials >= 1", who);
  if (use_coop(p)) return fail(LQG_ERR_ARG, "%s: LQG_F32_SYS64 is served by the lane kernels only", who);
  const Workspace w = carve(p, true);
  if (!workspace || workspace_bytes < w.total)
    return fail(LQG_ERR_WORKSPACE, "%s: workspace %zu B < required %zu B", who, workspace_bytes, w.total);
  char* base = static_cast<char*>(workspace);
  void* Ls = base + w.ls_off;
  void* ops = base + w.ops_off;
  bool found;
  auto mark = [&](int i) {
    if (p->phase_events[i]) (void)hipEventRecord(static_cast<hipEvent_t>(p->phase_events[i]), st);
  };
  const lqg_view none{nullptr, 0, 0, 0, 0};
  const lqg_traj no_mu{nullptr, 0, 0, 0, 0};
  mark(0);
  hipError_t e = dispatch_riccati<double>(p, none, none, none, Ls, w.ldb, st, &found);
  if (!found) return unsupported(p, who);
  if (e != hipSuccess) return done(e, who);
  mark(1);
  e = dispatch_forward_ops32(p, Ls, w.ldb, ops, st, &found);
  if (!found) return unsupported(p, who);
  if (e != hipSuccess) return done(e, who);
  mark(2);
  e = dispatch_trial<float>(p, ops, x, no_mu, ll, ll_sb, ll_sn, st, &found);
  if (!found) return unsupported(p, who);
  mark(3);
  return done(e, who);
}

// The same three phases on the cooperative kernels: Riccati -> forward sweep (always through the operator stream) ->
// per-trial sweep (the instantiated k_trial when the shape has one, else the run-time-dims k_coop_trial).
template <typename R>
int run_coop(const lqg_problem* p, lqg_traj x, lqg_traj mu, lqg_view Sigma, void* ll, long ll_sb, long ll_sn,
             const GainOutputs& g, void* workspace, size_t workspace_bytes, hipStream_t st, const char* who) {
  const Workspace w = carve(p, true);
  const size_t arena_bytes = coop_arena_bytes(p, false, false);
  if (!workspace || workspace_bytes < w.total + arena_bytes)
    return fail(LQG_ERR_WORKSPACE, "%s: workspace %zu B < required %zu B", who, workspace_bytes, w.total + arena_bytes);
  char* base = static_cast<char*>(workspace);
  void* Ls = base + w.ls_off;
  void* ops = base + w.ops_off;
  void* arena = arena_bytes ? base + w.total : nullptr;
  auto mark = [&](int i) {
    if (p->phase_events[i]) (void)hipEventRecord(static_cast<hipEvent_t>(p->phase_events[i]), st);
  };
  mark(0);
  hipError_t e = coop_riccati<R>(p, g.L, g.l, g.H, Ls, arena, st);
  if (e != hipSuccess) return done(e, who);
  mark(1);
  e = coop_forward<R>(p, Ls, ops, Sigma, g.K, arena, st);
  if (e != hipSuccess) return done(e, who);
  mark(2);
  if (p->n_trials > 0 && (ll || mu.ptr)) {
    bool found;
    e = dispatch_trial<R>(p, ops, x, mu, ll, ll_sb, ll_sn, st, &found);
    if (!found) e = coop_trial<R>(p, ops, x, mu, ll, ll_sb, ll_sn, st);
  }
  mark(3);
  return done(e, who);
}

// Time-parallel system sweeps (lqg_scan.hpp), then the per-trial sweep over the operator stream they leave.
template <typename R>
int run_scan_path(const lqg_problem* p, lqg_traj x, lqg_traj mu, lqg_view Sigma, void* ll, long ll_sb, long ll_sn,
                  void* workspace, size_t workspace_bytes, hipStream_t st, const char* who,
                  lqg_trial_sweep_fn trial_sweep = nullptr) {
  const size_t need = scan_workspace_bytes(p);
  if (!workspace || workspace_bytes < need)
    return fail(LQG_ERR_WORKSPACE, "%s: workspace %zu B < required %zu B", who, workspace_bytes, need);
  auto mark = [&](int i) {
    if (p->phase_events[i]) (void)hipEventRecord(static_cast<hipEvent_t>(p->phase_events[i]), st);
  };
  mark(0);
  mark(1);                                  // (no separate Riccati phase: the three scans are reported as "forward")
  void* ops = nullptr;
  hipError_t e = scan_system_sweeps<R>(p, Sigma, workspace, &ops, st);
  if (e != hipSuccess) return done(e, who);
  mark(2);
  if (p->n_trials > 0 && (ll || mu.ptr)) {
    // (a structure-specialised library's sweep when the caller brought one and it accepts the problem)
    const bool delegated = trial_sweep && ll && !mu.ptr && trial_sweep(p, x, ll, ll_sb, ll_sn, ops, st) == 0;
    if (!delegated) {
      bool found;
      e = dispatch_trial<R>(p, ops, x, mu, ll, ll_sb, ll_sn, st, &found);
      if (!found) e = coop_trial<R>(p, ops, x, mu, ll, ll_sb, ll_sn, st);
    }
  }
  mark(3);
  return done(e, who);
}

}  // namespace

extern "C" {

int lqg_abi_version(void) { return LQG_ABI_VERSION; }
const char* lqg_last_error(void) { return g_err; }
const char* lqg_target_arch(void) { return "gfx950"; }

int lqg_scan_supported(const lqg_problem* p) { return p && check_problem(p, "lqg_scan_supported") == 0 && scan_supported(p) ? 1 : 0; }
size_t lqg_scan_workspace_bytes(const lqg_problem* p) { return p ? scan_workspace_bytes(p) : 0; }

int lqg_log_likelihood_scan(const lqg_problem* p, lqg_traj x, void* ll, int64_t ll_sb, int64_t ll_sn, void* workspace,
                            size_t workspace_bytes, void* stream) {
  return lqg_log_likelihood_scan_with(p, x, ll, ll_sb, ll_sn, workspace, workspace_bytes, stream, nullptr);
}

int lqg_log_likelihood_scan_with(const lqg_problem* p, lqg_traj x, void* ll, int64_t ll_sb, int64_t ll_sn, void* workspace,
                                 size_t workspace_bytes, void* stream, lqg_trial_sweep_fn trial_sweep) {
  static const char* who = "lqg_log_likelihood_scan";
  if (int rc = check_full(p, who)) return rc;
  if (!scan_supported(p)) return fail(LQG_ERR_DIMS, "%s: needs u, y, d <= 4, x + b <= 24 (or b <= 64, x + b - d <= 64 with per-step working sets within LDS), no affine cost terms, T >= 2", who);
  if (p->n_sys == 0 || p->n_trials == 0) return 0;
  if (!x.ptr) return fail(LQG_ERR_NULL, "%s: x.ptr is NULL", who);
  if (!ll) return fail(LQG_ERR_NULL, "%s: ll is NULL", who);
  const lqg_traj no_mu{nullptr, 0, 0, 0, 0};
  const lqg_view no_sig{nullptr, 0, 0, 0, 0};
  return p->dtype == LQG_F64 ? run_scan_path<double>(p, x, no_mu, no_sig, ll, ll_sb, ll_sn, workspace, workspace_bytes,
                                                     (hipStream_t)stream, who, trial_sweep)
                             : run_scan_path<float>(p, x, no_mu, no_sig, ll, ll_sb, ll_sn, workspace, workspace_bytes,
                                                    (hipStream_t)stream, who, trial_sweep);
}

int lqg_conditional_moments_scan(const lqg_problem* p, lqg_traj x, lqg_traj mu, lqg_view Sigma, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  static const char* who = "lqg_conditional_moments_scan";
  if (int rc = check_full(p, who)) return rc;
  if (!scan_supported(p)) return fail(LQG_ERR_DIMS, "%s: needs u, y, d <= 4, x + b <= 24 (or b <= 64, x + b - d <= 64 with per-step working sets within LDS), no affine cost terms, T >= 2", who);
  if (!x.ptr) return fail(LQG_ERR_NULL, "%s: x.ptr is NULL", who);
  if (p->n_sys == 0) return 0;
  return p->dtype == LQG_F64 ? run_scan_path<double>(p, x, mu, Sigma, nullptr, 0, 0, workspace, workspace_bytes,
                                                     (hipStream_t)stream, who)
                             : run_scan_path<float>(p, x, mu, Sigma, nullptr, 0, 0, workspace, workspace_bytes,
                                                    (hipStream_t)stream, who);
}

int lqg_coop_supported(const lqg_dims* dims) { return dims && coop_supported(*dims) ? 1 : 0; }
int lqg_strategy(const lqg_problem* p) { return p && use_coop(p) ? LQG_STRATEGY_COOP : LQG_STRATEGY_LANE; }

static bool has_adjoint_lane(const lqg_dims& d);
int lqg_kernel_supported(int32_t family, const lqg_dims* dims) {
  if (!dims) return 0;
  const lqg_dims& d = *dims;
  (void)d;
  switch (family) {
    case LQG_FAMILY_FORWARD: return has_forward(d) ? 1 : 0;
    case LQG_FAMILY_RICCATI:
#define X(B_, U_) if (d.b == B_ && d.u == U_) return 1;
      LQG_RICCATI_DIMS(X)
#undef X
      return 0;
    case LQG_FAMILY_KALMAN:
#define X(B_, Y_) if (d.b == B_ && d.y == Y_) return 1;
      LQG_KALMAN_DIMS(X)
#undef X
      return 0;
    case LQG_FAMILY_TRIAL:
#define X(M_, D_) if (d.x + d.b == M_ && d.d == D_) return 1;
      LQG_TRIAL_DIMS(X)
#undef X
      return 0;
    case LQG_FAMILY_SIMULATE:
#define X(X_, B_, U_, Y_) if (d.x == X_ && d.b == B_ && d.u == U_ && d.y == Y_) return 1;
      LQG_SIM_DIMS(X)
#undef X
      return 0;
    case LQG_FAMILY_ADJOINT: return has_adjoint_lane(*dims) ? 1 : 0;   // (lane kernels; lqg_grad_supported also counts the cooperative sweep)
    default: return 0;
  }
}

int lqg_dims_supported(int32_t dtype, const lqg_dims* dims) {
  if (!dims || (dtype != LQG_F32 && dtype != LQG_F64)) return 0;
  return has_forward(*dims) ? 1 : 0;
}

int lqg_riccati_backward(const lqg_problem* p, lqg_view L, lqg_view l, lqg_view H, void* stream) {
  static const char* who = "lqg_riccati_backward";
  if (int rc = check_problem(p, who)) return rc;
  const lqg_spec& a = p->actor;
  if (need(a.Q, who, "actor.Q") || need(a.Qf, who, "actor.Qf") || need(a.R, who, "actor.R") ||
      need(a.A, who, "actor.A") || need(a.B, who, "actor.B") || need(L, who, "L"))
    return LQG_ERR_NULL;
  if (p->n_sys == 0) return 0;
  bool found = false;
  hipError_t e = hipSuccess;
  const bool can_coop = p->dims.u <= 6 && coop_fits_lds(p, false, true);     // (y, d play no part in lqr.backward)
  if (!(p->tuning.coop > 0 && can_coop))
    e = p->dtype == LQG_F64 ? dispatch_riccati<double>(p, L, l, H, nullptr, 0, (hipStream_t)stream, &found)
                            : dispatch_riccati<float>(p, L, l, H, nullptr, 0, (hipStream_t)stream, &found);
  if (!found) {                 // no (b, u) instantiation: the cooperative run-time-dims kernel (working set in LDS)
    if (!can_coop) return unsupported(p, who);
    e = p->dtype == LQG_F64 ? coop_riccati<double>(p, L, l, H, nullptr, nullptr, (hipStream_t)stream)
                            : coop_riccati<float>(p, L, l, H, nullptr, nullptr, (hipStream_t)stream);
  }
  return done(e, who);
}

int lqg_kalman_forward(const lqg_problem* p, lqg_view K, void* stream) {
  static const char* who = "lqg_kalman_forward";
  if (int rc = check_problem(p, who)) return rc;
  const lqg_spec& a = p->actor;
  if (need(a.A, who, "actor.A") || need(a.F, who, "actor.F") || need(a.V, who, "actor.V") ||
      need(a.W, who, "actor.W") || need(K, who, "K"))
    return LQG_ERR_NULL;
  if (p->n_sys == 0) return 0;
  bool found = false;
  hipError_t e = hipSuccess;
  const bool can_coop = p->dims.y <= 6 && coop_fits_lds(p, true, false);     // (u, d play no part in kf.forward)
  if (!(p->tuning.coop > 0 && can_coop))
    e = p->dtype == LQG_F64 ? dispatch_kalman<double>(p, K, (hipStream_t)stream, &found)
                            : dispatch_kalman<float>(p, K, (hipStream_t)stream, &found);
  if (!found) {                 // no (b, y) instantiation: the cooperative run-time-dims kernel (working set in LDS)
    if (!can_coop) return unsupported(p, who);
    const lqg_view none{nullptr, 0, 0, 0, 0};
    e = p->dtype == LQG_F64 ? coop_forward<double>(p, nullptr, nullptr, none, K, nullptr, (hipStream_t)stream)
                            : coop_forward<float>(p, nullptr, nullptr, none, K, nullptr, (hipStream_t)stream);
  }
  return done(e, who);
}

size_t lqg_workspace_bytes(const lqg_problem* p, int32_t op) {
  if (!p) return 0;
  (void)op;   // every op runs fused (no operator stream) when there is one trial per system
  if (use_coop(p)) return carve(p, true).total + coop_arena_bytes(p, false, false);
  return carve(p, p->n_trials != 1 || p->dtype == LQG_F32_SYS64).total;     // (mixed: always through the operator stream)
}

int lqg_conditional_moments(const lqg_problem* p, lqg_traj x, lqg_traj mu, lqg_view Sigma, void* workspace,
                            size_t workspace_bytes, void* stream) {
  static const char* who = "lqg_conditional_moments";
  if (int rc = check_full(p, who)) return rc;
  if (!x.ptr) return fail(LQG_ERR_NULL, "%s: x.ptr is NULL", who);
  if (p->n_sys == 0) return 0;
  const GainOutputs none{};
  return p->dtype == LQG_F64
             ? run_moments<double>(p, x, mu, Sigma, nullptr, 0, 0, none, workspace, workspace_bytes,
                                   (hipStream_t)stream, who)
             : run_moments<float>(p, x, mu, Sigma, nullptr, 0, 0, none, workspace, workspace_bytes,
                                  (hipStream_t)stream, who);
}

int lqg_solve_materialised(const lqg_problem* p, lqg_traj x, lqg_view L, lqg_view l, lqg_view H, lqg_view K,
                           lqg_traj mu, lqg_view Sigma, void* ll, int64_t ll_sb, int64_t ll_sn, void* workspace,
                           size_t workspace_bytes, void* stream) {
  static const char* who = "lqg_solve_materialised";
  if (int rc = check_full(p, who)) return rc;
  if (!x.ptr) return fail(LQG_ERR_NULL, "%s: x.ptr is NULL", who);
  if (p->n_sys == 0) return 0;
  const GainOutputs g{L, l, H, K};
  return p->dtype == LQG_F64
             ? run_moments<double>(p, x, mu, Sigma, ll, ll_sb, ll_sn, g, workspace, workspace_bytes,
                                   (hipStream_t)stream, who)
             : run_moments<float>(p, x, mu, Sigma, ll, ll_sb, ll_sn, g, workspace, workspace_bytes,
                                  (hipStream_t)stream, who);
}

int lqg_log_likelihood(const lqg_problem* p, lqg_traj x, void* ll, int64_t ll_sb, int64_t ll_sn, void* workspace,
                       size_t workspace_bytes, void* stream) {
  static const char* who = "lqg_log_likelihood";
  if (int rc = check_full(p, who, true)) return rc;
  if (p->n_sys == 0 || p->n_trials == 0) return 0;   // empty batch: nothing to do (pointers may be NULL)
  if (!x.ptr) return fail(LQG_ERR_NULL, "%s: x.ptr is NULL", who);
  if (!ll) return fail(LQG_ERR_NULL, "%s: ll is NULL", who);
  if (p->dtype == LQG_F32_SYS64)
    return run_mixed(p, x, ll, ll_sb, ll_sn, workspace, workspace_bytes, (hipStream_t)stream, who);
  const lqg_traj no_mu{nullptr, 0, 0, 0, 0};
  const lqg_view no_sig{nullptr, 0, 0, 0, 0};
  const GainOutputs none{};
  return p->dtype == LQG_F64
             ? run_moments<double>(p, x, no_mu, no_sig, ll, ll_sb, ll_sn, none, workspace, workspace_bytes,
                                   (hipStream_t)stream, who)
             : run_moments<float>(p, x, no_mu, no_sig, ll, ll_sb, ll_sn, none, workspace, workspace_bytes,
                                  (hipStream_t)stream, who);
}

size_t lqg_sum_trials_workspace_bytes(int64_t n_sys, int64_t n_trials) {
  const size_t chunks = (size_t)((n_trials + lqg::kSumChunk - 1) / lqg::kSumChunk);
  return chunks > 1 ? (size_t)n_sys * chunks * sizeof(double) : 0;
}

int lqg_sum_trials(int32_t dtype, const void* ll, int64_t n_sys, int64_t n_trials, int64_t ll_sb, int64_t ll_sn,
                   double* out, void* workspace, size_t workspace_bytes, void* stream) {
  static const char* who = "lqg_sum_trials";
  if (!ll || !out) return fail(LQG_ERR_NULL, "%s: NULL pointer", who);
  if (dtype != LQG_F32 && dtype != LQG_F64) return fail(LQG_ERR_ARG, "%s: bad dtype %d", who, dtype);
  if (n_sys <= 0) return 0;
  const long chunks = (long)((n_trials + lqg::kSumChunk - 1) / lqg::kSumChunk);
  const bool two_stage = chunks > 1;
  if (two_stage && (!workspace || workspace_bytes < lqg_sum_trials_workspace_bytes(n_sys, n_trials)))
    return fail(LQG_ERR_WORKSPACE, "%s: workspace %zu B < required %zu B", who, workspace_bytes,
                lqg_sum_trials_workspace_bytes(n_sys, n_trials));
  double* part = two_stage ? static_cast<double*>(workspace) : out;
  const dim3 grid((unsigned)(chunks > 0 ? chunks : 1), (unsigned)n_sys), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == LQG_F64)
    hipLaunchKernelGGL((lqg::k_sum_trials<double>), grid, block, 0, st, static_cast<const double*>(ll), (long)n_trials,
                       (long)ll_sb, (long)ll_sn, part, chunks > 0 ? chunks : 1);
  else
    hipLaunchKernelGGL((lqg::k_sum_trials<float>), grid, block, 0, st, static_cast<const float*>(ll), (long)n_trials,
                       (long)ll_sb, (long)ll_sn, part, chunks > 0 ? chunks : 1);
  if (two_stage) hipLaunchKernelGGL((lqg::k_sum_partials<0>), dim3((unsigned)n_sys), block, 0, st, part, chunks, out);
  return done(hipGetLastError(), who);
}

static int simulate_impl(const lqg_problem* p, lqg_view L, lqg_view l, lqg_view K, lqg_traj eps, lqg_traj eta, lqg_view x0,
                         lqg_view xhat0, lqg_traj xs, lqg_traj xhat, lqg_traj ys, lqg_traj us, void* stream, bool rng,
                         unsigned long long seed, const char* who);

int lqg_simulate(const lqg_problem* p, lqg_view L, lqg_view l, lqg_view K, lqg_traj eps, lqg_traj eta, lqg_view x0,
                 lqg_view xhat0, lqg_traj xs, lqg_traj xhat, lqg_traj ys, lqg_traj us, void* stream) {
  return simulate_impl(p, L, l, K, eps, eta, x0, xhat0, xs, xhat, ys, us, stream, false, 0, "lqg_simulate");
}

int lqg_simulate_rng(const lqg_problem* p, lqg_view L, lqg_view l, lqg_view K, uint64_t seed, lqg_view x0, lqg_view xhat0,
                     lqg_traj xs, lqg_traj xhat, lqg_traj ys, lqg_traj us, void* stream) {
  const lqg_traj none{nullptr, 0, 0, 0, 0};
  return simulate_impl(p, L, l, K, none, none, x0, xhat0, xs, xhat, ys, us, stream, true, (unsigned long long)seed,
                       "lqg_simulate_rng");
}

static int simulate_impl(const lqg_problem* p, lqg_view L, lqg_view l, lqg_view K, lqg_traj eps, lqg_traj eta, lqg_view x0,
                         lqg_view xhat0, lqg_traj xs, lqg_traj xhat, lqg_traj ys, lqg_traj us, void* stream, bool rng,
                         unsigned long long seed, const char* who) {
  if (int rc = check_problem(p, who)) return rc;
  const lqg_spec& a = p->actor;
  const lqg_spec& d = p->dynamics;
  if (need(a.A, who, "actor.A") || need(a.B, who, "actor.B") || need(a.F, who, "actor.F") ||
      need(d.A, who, "dynamics.A") || need(d.B, who, "dynamics.B") || need(d.F, who, "dynamics.F") ||
      need(d.V, who, "dynamics.V") || need(d.W, who, "dynamics.W") || need(L, who, "L") || need(K, who, "K"))
    return LQG_ERR_NULL;
  if (!rng && (!eps.ptr || !eta.ptr)) return fail(LQG_ERR_NULL, "%s: eps/eta must be non-NULL", who);
  if (!xs.ptr) return fail(LQG_ERR_NULL, "%s: xs must be non-NULL", who);
  if (p->n_sys == 0 || p->n_trials == 0) return 0;
  bool found;
  hipError_t e = p->dtype == LQG_F64
                     ? dispatch_simulate<double>(p, L, l, K, eps, eta, x0, xhat0, xs, xhat, ys, us,
                                                 (hipStream_t)stream, &found, seed)
                     : dispatch_simulate<float>(p, L, l, K, eps, eta, x0, xhat0, xs, xhat, ys, us,
                                                (hipStream_t)stream, &found, seed);
  if (!found)                   // no (x, b, u, y) instantiation: the run-time-dims kernel (per-thread state in LDS)
    e = p->dtype == LQG_F64
            ? coop_simulate<double>(p, L, l, K, eps, eta, x0, xhat0, xs, xhat, ys, us, (hipStream_t)stream, seed)
            : coop_simulate<float>(p, L, l, K, eps, eta, x0, xhat0, xs, xhat, ys, us, (hipStream_t)stream, seed);
  return done(e, who);
}

int lqg_gaussian_logprob(int32_t dtype, int32_t k, int32_t T, int64_t n_sys, int64_t n_trials, lqg_traj value,
                         lqg_traj mu, lqg_view Sigma, void* out, int64_t out_sb, int64_t out_sn, void* stream) {
  static const char* who = "lqg_gaussian_logprob";
  if (!value.ptr || !mu.ptr || !Sigma.ptr || !out) return fail(LQG_ERR_NULL, "%s: NULL pointer", who);
  if (dtype != LQG_F32 && dtype != LQG_F64) return fail(LQG_ERR_ARG, "%s: bad dtype %d", who, dtype);
  if (k < 1 || k > 6) return fail(LQG_ERR_DIMS, "%s: event dim %d not in [1,6]", who, k);
  if (n_sys <= 0 || n_trials <= 0) return 0;
  const dim3 grid(blocks_for(n_trials), (unsigned)n_sys), block(LQG_BLOCK);
  hipStream_t st = (hipStream_t)stream;
#define LP(R, KD)                                                                                           \
  {                                                                                                         \
    lqg::LogprobArgs<R> a{dt<R>(value), dt<R>(mu), dv<R>(Sigma), static_cast<R*>(out), (long)out_sb,        \
                          (long)out_sn, (long)n_sys, (long)n_trials, T};                                    \
    hipLaunchKernelGGL((lqg::k_gaussian_logprob<R, KD>), grid, block, 0, st, a);                            \
  }
#define LPK(KD)                 \
  case KD:                      \
    if (dtype == LQG_F64) LP(double, KD) else LP(float, KD) break;
  switch (k) { LPK(1) LPK(2) LPK(3) LPK(4) LPK(5) LPK(6) }
#undef LPK
#undef LP
  return done(hipGetLastError(), who);
}

static bool has_adjoint_lane(const lqg_dims& d) {
  (void)d;
#define X(X_, B_, U_, Y_, D_) \
  if (d.x == X_ && d.b == B_ && d.u == U_ && d.y == Y_ && d.d == D_) return true;
  LQG_ADJOINT_DIMS(X)
#undef X
  return false;
}

int lqg_grad_supported(int32_t dtype, const lqg_dims* dims) {
  if (!dims || (dtype != LQG_F32 && dtype != LQG_F64)) return 0;
  if (has_adjoint_lane(*dims)) return 1;
  return coop_adjoint_supported(dtype, *dims);       // fp64, any x, b <= 64 with u, y, d <= 4 (lqg_coop_adjoint.hip)
}

// tuning.coop_adjoint == 1: the cooperative sweep also for shapes WITH adjoint lane kernels (fp64 problems;
// tests pin it against the lane kernels on every golden case)
static bool adjoint_on_lanes(const lqg_problem* p) {
  return has_adjoint_lane(p->dims) && !(p->tuning.coop_adjoint == 1 && coop_adjoint_supported(p->dtype, p->dims));
}

int32_t lqg_grad_lanes_per_system(const lqg_problem* p) {
  if (!p) return 0;
  return adjoint_on_lanes(p) ? (int32_t)p->n_trials : 1;
}

int64_t lqg_grad_elements(const lqg_dims* dims) { return dims ? (int64_t)adj_grad_elements(*dims) : 0; }

int32_t lqg_grad_slabs(const lqg_problem* p) { return !p ? 0 : (adj_time_invariant(p) ? 1 : p->T); }

size_t lqg_grad_workspace_bytes(const lqg_problem* p, int64_t ld) {
  if (!p || p->T <= 0 || ld <= 0) return 0;
  if (!adjoint_on_lanes(p)) return coop_adjoint_supported(p->dtype, p->dims) ? coop_adjoint_workspace_bytes(p) : 0;
  const size_t e = p->dtype == LQG_F64 ? 8 : 4;
  return (size_t)p->T * (size_t)adj_step_reals(p->dims) * (size_t)ld * e;
}

int lqg_log_likelihood_grad(const lqg_problem* p, lqg_traj x, const void* g, int64_t g_sb, int64_t g_sn, void* ll,
                            int64_t ll_sb, int64_t ll_sn, void* grad, int64_t ld, void* workspace,
                            size_t workspace_bytes, int32_t phases, void* stream) {
  static const char* who = "lqg_log_likelihood_grad";
  if (phases < 1 || phases > 3) return fail(LQG_ERR_ARG, "%s: phases must be 1 (forward), 2 (reverse) or 3 (both)", who);
  if (int rc = check_problem(p, who)) return rc;
  const lqg_spec& a = p->actor;
  const lqg_spec& dy = p->dynamics;
  if (need(a.Q, who, "actor.Q") || need(a.Qf, who, "actor.Qf") || need(a.R, who, "actor.R") ||
      need(a.A, who, "actor.A") || need(a.B, who, "actor.B") || need(a.F, who, "actor.F") ||
      need(a.V, who, "actor.V") || need(a.W, who, "actor.W") || need(dy.A, who, "dynamics.A") ||
      need(dy.B, who, "dynamics.B") || need(dy.F, who, "dynamics.F") || need(dy.V, who, "dynamics.V") ||
      need(dy.W, who, "dynamics.W"))
    return LQG_ERR_NULL;
  if (p->n_sys == 0 || p->n_trials == 0) return 0;
  if (!x.ptr || !workspace || ((phases & 2) && !grad)) return fail(LQG_ERR_NULL, "%s: NULL x / grad / workspace", who);
  const bool lane = adjoint_on_lanes(p);
  if (!lane && !coop_adjoint_supported(p->dtype, p->dims)) return unsupported(p, who);
  if (ld < p->n_sys * (lane ? p->n_trials : 1))
    return fail(LQG_ERR_ARG, "%s: ld %lld < n_sys * lqg_grad_lanes_per_system", who, (long long)ld);
  if (workspace_bytes < lqg_grad_workspace_bytes(p, ld))
    return fail(LQG_ERR_WORKSPACE, "%s: workspace %zu < %zu bytes", who, workspace_bytes, lqg_grad_workspace_bytes(p, ld));
  const lqg_dims& d = p->dims;
  (void)d;
  hipStream_t st = (hipStream_t)stream;
#define X(X_, B_, U_, Y_, D_)                                                                                   \
  if (lane && d.x == X_ && d.b == B_ && d.u == U_ && d.y == Y_ && d.d == D_)                                   \
    return done(p->dtype == LQG_F64                                                                             \
                    ? launch_adjoint<double, X_, B_, U_, Y_, D_>(p, x, g, g_sb, g_sn, ll, ll_sb, ll_sn, grad, ld, \
                                                                 workspace, phases, st)                                  \
                    : launch_adjoint<float, X_, B_, U_, Y_, D_>(p, x, g, g_sb, g_sn, ll, ll_sb, ll_sn, grad, ld,  \
                                                                workspace, phases, st),                                  \
                who);
  LQG_ADJOINT_DIMS(X)
#undef X
  // no lane kernels for this shape: one workgroup per system, bars summed over the trials (lqg_coop_adjoint.hip)
  return done(coop_adjoint_run(p, x, g, g_sb, g_sn, ll, ll_sb, ll_sn, grad, ld, (long)adj_grad_elements(p->dims),
                               adj_time_invariant(p), workspace, phases, st), who);
}

}  // extern "C"
