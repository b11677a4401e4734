// lqg_adjoint_sp_entry.hpp — host side of a structure-specialised ADJOINT library (one per sparsity pattern, generated and
// compiled by lqg_amd/specialize.py: csrc/pat/padj_<key>.so).  Exposes the contract of lqg_log_likelihood_grad
// (include/lqg_hip.h) restricted to what it is compiled for — time-invariant specs, no affine cost terms, fixed dims and
// pattern — with the bars returned ALREADY summed over the trials of a system (lqg_grad_lanes_per_system == 1: ld >= n_sys,
// lane = system).  Anything else is refused with LQG_ERR_ARG / LQG_ERR_DIMS before launching; the caller then uses the
// round-1 lane kernels of the main library.
#pragma once
#include "lqg_adjoint_sp.hpp"
#include "lqg_adjoint_trial_sp.hpp"
#include "lqg_launch.hpp"

namespace lqg {
namespace host {

#ifndef LQG_ASP_CK
#define LQG_ASP_CK 4            // steps per checkpoint of the system sweeps (the chunk's states live in registers)
#endif
#ifndef LQG_ASP_FWD_WIDE
#define LQG_ASP_FWD_WIDE 1      // the checkpoint-keeping forward per-trial sweep on 256-lane workgroups for many trials x many candidates
#endif
#ifndef LQG_ASP_CKT
#define LQG_ASP_CKT 8           // steps per checkpoint of the per-trial sweeps
#endif

struct AspWorkspace {
  size_t sck_off, ck_off, ops_off, tck_off, sums_off, gsum_off, lbar_off, kbar_off, total;
  long ldb, npad;
  int nck, nckt, parts;
};

template <int NX, int NB, int NU, int NY, int ND, int NSUM>
inline AspWorkspace asp_carve(const lqg_problem* p) {
  constexpr int M = NX + NB;
  AspWorkspace w{};
  const size_t esz = p->dtype == LQG_F64 ? 8 : 4;
  const bool fused = p->n_trials <= 2;
  const int ntr = fused ? (int)p->n_trials : 0;
  w.ldb = round_up(p->n_sys, 64);
  w.nck = (p->T + LQG_ASP_CK - 1) / LQG_ASP_CK;
  w.nckt = (p->T + LQG_ASP_CKT - 1) / LQG_ASP_CKT;
  w.npad = round_up(p->n_trials, 64);
  const long per_block = (long)LQG_ASP_TRIAL_BLOCK * LQG_ASP_TPL;
  w.parts = fused ? 0 : (int)((p->n_trials + per_block - 1) / per_block);
  auto al = [](size_t v) { return (v + 255) / 256 * 256; };
  size_t off = 0;
  w.sck_off = off;
  off += al((size_t)w.nck * (NB * (NB + 1) / 2) * w.ldb * esz);
  w.ck_off = off;
  off += al((size_t)(w.nck + 1) * (NB * (NB + 1) / 2 + M * (M + 1) / 2 + ntr * M) * w.ldb * esz);
  w.ops_off = off;
  off += fused ? 0 : al((size_t)p->n_sys * (size_t)(p->T + 1) * ops_reals(p->dims) * esz);
  w.tck_off = off;
  off += fused ? 0 : al((size_t)p->n_sys * (size_t)(w.nckt + 1) * (M - ND) * w.npad * esz);   // (c_{t-1}: M - ND reals per record)
  w.sums_off = off;
  off += fused ? 0 : al((size_t)w.parts * p->n_sys * (size_t)p->T * NSUM * esz);
  w.gsum_off = off;
  off += fused ? 0 : al((size_t)w.parts * p->n_sys * esz);
  w.lbar_off = off;
  off += al((size_t)p->T * NU * NB * w.ldb * esz);
  w.kbar_off = off;
  off += al((size_t)p->T * NB * NY * w.ldb * esz);
  w.total = off;
  return w;
}

template <typename PAT, int NX, int NB, int NU, int NY, int ND>
inline int asp_check(const lqg_problem* p) {
  if (!p) return LQG_ERR_NULL;
  const lqg_dims& dm = p->dims;
  if (dm.x != NX || dm.b != NB || dm.u != NU || dm.y != NY || dm.d != ND) return LQG_ERR_DIMS;
  if (p->n_trials < 1 || p->T < 1) return LQG_ERR_ARG;
  if (!forward_ti(p) || !actor_ti_riccati(p) || affine(p)) return LQG_ERR_ARG;
  if (p->dtype != LQG_F32 && p->dtype != LQG_F64) return LQG_ERR_ARG;
  return 0;
}

template <typename PAT, int NX, int NB, int NU, int NY, int ND>
size_t grad_workspace_bytes_sp(const lqg_problem* p) {
  if (asp_check<PAT, NX, NB, NU, NY, ND>(p) != 0) return 0;
  using SMd = lqg::asp::Sums<NX + NB, ND, lqg::asp::Masks<PAT, NX, NB, NU, NY, true>::FJ>;
  return asp_carve<NX, NB, NU, NY, ND, SMd::N>(p).total;       // (the dense-P mask is a superset: the larger record)
}

template <typename R, typename PAT, int NX, int NB, int NU, int NY, int ND, bool DENSE_P>
int run_asp(const lqg_problem* p, lqg_traj x, const void* g, long g_sb, long g_sn, void* ll, long ll_sb, long ll_sn, void* grad,
            long ld, void* workspace, size_t workspace_bytes, int phases, hipStream_t st) {
  constexpr int M = NX + NB;
  constexpr int CK = LQG_ASP_CK;
  using MK = lqg::asp::Masks<PAT, NX, NB, NU, NY, DENSE_P>;
  using SM = lqg::asp::Sums<M, ND, MK::FJ>;
  using SMd = lqg::asp::Sums<M, ND, lqg::asp::Masks<PAT, NX, NB, NU, NY, true>::FJ>;
  const AspWorkspace w = asp_carve<NX, NB, NU, NY, ND, SMd::N>(p);
  if (!workspace || workspace_bytes < w.total) return LQG_ERR_WORKSPACE;
  if (ld < p->n_sys) return LQG_ERR_ARG;
  char* base = static_cast<char*>(workspace);
  const lqg_spec& a = p->actor;
  const lqg_spec& d = p->dynamics;
  const lqg_view none{nullptr, 0, 0, 0, 0};
  const lqg_traj no_traj{nullptr, 0, 0, 0, 0};
  const bool fused = p->n_trials <= 2;
  R* Sck = reinterpret_cast<R*>(base + w.sck_off);
  R* ops = fused ? nullptr : reinterpret_cast<R*>(base + w.ops_off);
  lqg::asp::AspArgs<R> A{};
  A.f = lqg::ForwardArgs<R>{dv<R>(a.A), dv<R>(a.B), dv<R>(a.F), dv<R>(a.V), dv<R>(a.W),
                            dv<R>(d.A), dv<R>(d.B), dv<R>(d.F), dv<R>(d.V), dv<R>(d.W),
                            dv<R>(p->Sigma0), Sck, w.ldb, dt<R>(fused ? x : no_traj), static_cast<R*>(fused ? ll : nullptr), ll_sb, ops,
                            dv<R>(none), dt<R>(no_traj), dv<R>(none), (long)p->n_sys, p->T,
                            p->dims.nva, p->dims.nwa, p->dims.nvd, p->dims.nwd};
  A.rc = lqg::RiccatiArgs<R>{dv<R>(a.Q), dv<R>(a.q), dv<R>(a.Qf), dv<R>(a.qf), dv<R>(a.P), dv<R>(a.R), dv<R>(a.r),
                             dv<R>(a.A), dv<R>(a.B), dv<R>(none), dv<R>(none), dv<R>(none), Sck, w.ldb,
                             (long)p->n_sys, p->T, (R)p->eps};
  A.g = static_cast<const R*>(g);
  A.g_sb = g_sb;
  A.g_sn = g_sn;
  A.ll_sn = ll_sn;
  A.ck = reinterpret_cast<R*>(base + w.ck_off);
  A.sums = fused ? nullptr : reinterpret_cast<R*>(base + w.sums_off);
  A.parts = w.parts;
  A.gsum = fused ? nullptr : reinterpret_cast<R*>(base + w.gsum_off);
  A.Lbar = reinterpret_cast<R*>(base + w.lbar_off);
  A.Kbar = reinterpret_cast<R*>(base + w.kbar_off);
  A.out = static_cast<R*>(grad);
  A.ld = ld;
  lqg::asp::TrialRevArgs<R> tr{};
  tr.x = dt<R>(x);
  tr.g = static_cast<const R*>(g);
  tr.g_sb = g_sb;
  tr.g_sn = g_sn;
  tr.ll = static_cast<R*>(ll);
  tr.ll_sb = ll_sb;
  tr.ll_sn = ll_sn;
  tr.tck = fused ? nullptr : reinterpret_cast<R*>(base + w.tck_off);
  tr.npad = w.npad;
  tr.nckt = w.nckt;
  tr.sums = A.sums;
  tr.gsum = fused ? nullptr : reinterpret_cast<R*>(base + w.gsum_off);
  tr.n_sys = (long)p->n_sys;
  tr.n_trials = (long)p->n_trials;
  tr.T = p->T;
  const dim3 grid(blocks_for(p->n_sys)), block(LQG_BLOCK);
  const dim3 tgrid((unsigned)(fused ? 1 : w.parts), (unsigned)p->n_sys), tblock(LQG_ASP_TRIAL_BLOCK);
  auto mark = [&](int i) {
    if (p->phase_events[i]) (void)hipEventRecord(static_cast<hipEvent_t>(p->phase_events[i]), st);
  };
#define LQG_ASP_SYS(K_, NTR_) hipLaunchKernelGGL((lqg::asp::K_<R, NX, NB, NU, NY, ND, PAT, NTR_, DENSE_P, CK>), grid, block, 0, st, A)
  if (phases & 1) {
    mark(0);
    hipLaunchKernelGGL((lqg::k_riccati_sp<R, NB, NU, PAT, CK>), grid, block, 0, st, A.rc);
    mark(1);
    if (p->n_trials == 1) LQG_ASP_SYS(k_asp_sys_fwd, 1);
    else if (p->n_trials == 2) LQG_ASP_SYS(k_asp_sys_fwd, 2);
    else LQG_ASP_SYS(k_asp_sys_fwd, 0);
    mark(2);
    if (!fused) {   // the per-trial sweep of the forward path (k_trial_sp), keeping c_{t-1} for every CKT-th row
      lqg::TrialArgs<R> tk{dt<R>(x), dt<R>(no_traj), static_cast<R*>(ll), ll_sb, ll_sn, (long)p->n_trials, p->T, tr.tck, w.npad, w.nckt};
      constexpr auto FMT = lqg::trial_operator_mask<PAT, NX, NB, NU, NY, ND, DENSE_P>();
      const long lanes4 = (long)p->n_sys * ((p->n_trials + 4 * LQG_BLOCK - 1) / (4 * LQG_BLOCK)) * LQG_BLOCK;
      const bool wide = lanes4 >= 2L * 1024 * 64;
      if (LQG_ASP_FWD_WIDE && p->n_trials >= 768 && p->n_sys >= 256) {      // many trials x many candidates: 256-lane workgroups (lqg_sp_entry.hpp)
        const dim3 wgrid((unsigned)((p->n_trials + 511) / 512), (unsigned)p->n_sys);
        hipLaunchKernelGGL((lqg::k_trial_sp<R, M, ND, 2, FMT, LQG_ASP_CKT, 256>), wgrid, dim3(256), 0, st, ops, tk);
      } else {
        const long per_block = (long)LQG_BLOCK * (wide ? LQG_TRIALS_PER_LANE : 1);
        const dim3 fgrid((unsigned)((p->n_trials + per_block - 1) / per_block), (unsigned)p->n_sys);
        if (wide) hipLaunchKernelGGL((lqg::k_trial_sp<R, M, ND, LQG_TRIALS_PER_LANE, FMT, LQG_ASP_CKT>), fgrid, block, 0, st, ops, tk);
        else hipLaunchKernelGGL((lqg::k_trial_sp<R, M, ND, 1, FMT, LQG_ASP_CKT>), fgrid, block, 0, st, ops, tk);
      }
    }
    mark(3);
  }
  if (phases & 2) {
    if (!grad) return LQG_ERR_NULL;
    // (a reverse-only call records the same four caller events: [0] before the per-trial reverse sweep, [1] after it,
    // [2] after the system reverse sweep, [3] after the Riccati adjoint)
    const bool rev_only = !(phases & 1);
    if (rev_only) mark(0);
    if (!fused)
      hipLaunchKernelGGL((lqg::asp::k_asp_trial_rev<R, M, ND, LQG_ASP_TPL, LQG_ASP_CKT, MK::FJ>), tgrid, tblock, 0, st, ops, tr);
    if (rev_only) mark(1);
#if LQG_ASP_SPLIT_KAL
    if (p->n_trials == 1) { LQG_ASP_SYS(k_asp_sys_rev, 1); LQG_ASP_SYS(k_asp_kal_rev, 1); }
    else if (p->n_trials == 2) { LQG_ASP_SYS(k_asp_sys_rev, 2); LQG_ASP_SYS(k_asp_kal_rev, 2); }
    else { LQG_ASP_SYS(k_asp_sys_rev, 0); LQG_ASP_SYS(k_asp_kal_rev, 0); }
#else
    if (p->n_trials == 1) LQG_ASP_SYS(k_asp_sys_rev_fused, 1);
    else if (p->n_trials == 2) LQG_ASP_SYS(k_asp_sys_rev_fused, 2);
    else LQG_ASP_SYS(k_asp_sys_rev_fused, 0);
#endif
    if (rev_only) mark(2);
    hipLaunchKernelGGL((lqg::asp::k_asp_ric_rev<R, NB, NU, NX, NY, PAT, CK>), grid, block, 0, st, A);
    if (rev_only) mark(3);
  }
#undef LQG_ASP_SYS
  (void)sizeof(SM);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

template <typename PAT, int NX, int NB, int NU, int NY, int ND>
int log_likelihood_grad_sp(const lqg_problem* p, lqg_traj x, const void* g, int64_t g_sb, int64_t g_sn, void* ll, int64_t ll_sb,
                           int64_t ll_sn, void* grad, int64_t ld, void* workspace, size_t workspace_bytes, int32_t phases,
                           void* stream) {
  const int rc = asp_check<PAT, NX, NB, NU, NY, ND>(p);
  if (rc != 0) return rc;
  if (!x.ptr) return LQG_ERR_NULL;
  if (!(phases & 3)) return LQG_ERR_ARG;
  if (p->n_sys == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const bool dense_p = p->Sigma0.ptr != nullptr;
#define LQG_ASP_RUN(R_, DP_) \
  run_asp<R_, PAT, NX, NB, NU, NY, ND, DP_>(p, x, g, (long)g_sb, (long)g_sn, ll, (long)ll_sb, (long)ll_sn, grad, (long)ld, workspace, \
                                            workspace_bytes, phases, st)
  if (p->dtype == LQG_F64) return dense_p ? LQG_ASP_RUN(double, true) : LQG_ASP_RUN(double, false);
  return dense_p ? LQG_ASP_RUN(float, true) : LQG_ASP_RUN(float, false);
#undef LQG_ASP_RUN
}

}  // namespace host
}  // namespace lqg
