// lqg_adjoint_launch.hpp — host-side launcher of the four adjoint sweeps (lqg_adjoint.hpp); instantiated per
// (dtype, dims) by lqg_adjoint_inst.hip, dispatched by lqg_abi.hip (lqg_log_likelihood_grad).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/lqg_hip.h"
#include "lqg_adjoint.hpp"

namespace lqg {
namespace host {

template <typename R>
inline lqg::DView<R> adv(const lqg_view& v) {
  return lqg::DView<R>{static_cast<const R*>(v.ptr), (long)v.sb, (long)v.st, (long)v.sr, (long)v.sc};
}

inline long adj_step_reals(const lqg_dims& d) {
  const long m = d.x + d.b;
  return (long)d.b * (d.b + 1) / 2 * 2 + (long)d.u * d.b + m * (m + 1) / 2 + m;
}
inline long adj_grad_elements(const lqg_dims& d) {
  return 2L * d.x * d.x + (long)d.x * d.u + (long)d.y * d.x + 2L * d.y * d.y   // dA dVV dB dF dWW aWW
         + 6L * d.b * d.b + 2L * d.b * d.u + (long)d.y * d.b + (long)d.u * d.u;  // aA aVV aQ aQf aS0 aA2 aB aB2 aF aR
}

inline bool adj_ti(const lqg_view& v, int T) { return v.ptr == nullptr || v.st == 0 || T <= 1; }
// true when every field the gradient sweep reads is time-invariant (one accumulated bar per matrix)
inline bool adj_time_invariant(const lqg_problem* p) {
  const lqg_spec& a = p->actor;
  const lqg_spec& d = p->dynamics;
  const int T = p->T;
  return adj_ti(a.Q, T) && adj_ti(a.P, T) && adj_ti(a.R, T) && adj_ti(a.A, T) && adj_ti(a.B, T) && adj_ti(a.F, T) &&
         adj_ti(a.V, T) && adj_ti(a.W, T) && adj_ti(d.A, T) && adj_ti(d.B, T) && adj_ti(d.F, T) && adj_ti(d.V, T) &&
         adj_ti(d.W, T);
}

template <typename R, int NX, int NB, int NU, int NY, int ND>
hipError_t launch_adjoint(const lqg_problem* p, lqg_traj x, const void* g, long g_sb, long g_sn, void* ll, long ll_sb,
                          long ll_sn, void* grad, long ld, void* ws, int phases, hipStream_t stream) {
  using Lay = lqg::adj::Layout<NX, NB, NU, NY>;
  static_assert(Lay::STEP > 0 && Lay::TOTAL > 0, "layout");
  lqg::adj::AdjArgs<R> a;
  const lqg_spec& ac = p->actor;
  const lqg_spec& dy = p->dynamics;
  a.Q = adv<R>(ac.Q); a.Qf = adv<R>(ac.Qf); a.P = adv<R>(ac.P); a.Rm = adv<R>(ac.R); a.A = adv<R>(ac.A);
  a.B = adv<R>(ac.B); a.F = adv<R>(ac.F); a.V = adv<R>(ac.V); a.W = adv<R>(ac.W);
  a.dA = adv<R>(dy.A); a.dB = adv<R>(dy.B); a.dF = adv<R>(dy.F); a.dV = adv<R>(dy.V); a.dW = adv<R>(dy.W);
  a.Sigma0 = adv<R>(p->Sigma0);
  a.x = lqg::DTraj<R>{static_cast<const R*>(x.ptr), (long)x.sb, (long)x.sn, (long)x.st, (long)x.sd};
  a.g = static_cast<const R*>(g); a.g_sb = g_sb; a.g_sn = g_sn;
  a.ll = static_cast<R*>(ll); a.ll_sb = ll_sb; a.ll_sn = ll_sn;
  a.ws = static_cast<R*>(ws);
  a.out = static_cast<R*>(grad);
  a.ld = ld;
  a.n_trials = p->n_trials;
  a.n_lanes = p->n_sys * p->n_trials;
  a.T = p->T;
  a.nva = p->dims.nva; a.nwa = p->dims.nwa; a.nvd = p->dims.nvd; a.nwd = p->dims.nwd;
  a.eps = (R)p->eps;
  const unsigned nb = (unsigned)((a.n_lanes + LQG_BLOCK - 1) / LQG_BLOCK);
#define LQG_ADJ_LAUNCH(K_, TI_) \
  hipLaunchKernelGGL((lqg::adj::K_<R, NX, NB, NU, NY, ND, TI_>), dim3(nb), dim3(LQG_BLOCK), 0, stream, a)
  const bool ti = adj_time_invariant(p);
  if (phases & 1) {   // forward sweeps: fill the workspace, write the value
    if (ti) { LQG_ADJ_LAUNCH(k_adj_riccati, true); LQG_ADJ_LAUNCH(k_adj_forward, true); }
    else { LQG_ADJ_LAUNCH(k_adj_riccati, false); LQG_ADJ_LAUNCH(k_adj_forward, false); }
  }
  if (phases & 2) {   // reverse sweeps: consume the workspace, write the bars (one slab, or one per step)
    if (ti) { LQG_ADJ_LAUNCH(k_adj_reverse, true); LQG_ADJ_LAUNCH(k_adj_riccati_rev, true); }
    else { LQG_ADJ_LAUNCH(k_adj_reverse, false); LQG_ADJ_LAUNCH(k_adj_riccati_rev, false); }
  }
#undef LQG_ADJ_LAUNCH
  return hipGetLastError();
}

}  // namespace host
}  // namespace lqg
