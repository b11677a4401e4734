// lqg_sp_entry.hpp — host side of a structure-specialised library (one per sparsity pattern, generated and compiled
// by lqg_amd/specialize.py).  Exposes the SAME contract as lqg_log_likelihood (include/lqg_hip.h) restricted to the
// case it is compiled for: time-invariant specs, no affine cost terms, dims fixed (one or two trials per system run
// fused in-lane, more go through the operator stream and the generic k_trial).
// Anything else is refused with LQG_ERR_ARG / LQG_ERR_DIMS before launching; the caller then uses the generic
// library.  The pattern's validity for the data (structural zeros really are zero) is the generator's contract.
#pragma once
#include <cstdint>
#include <cstdio>

#include "lqg_kernels_sp.hpp"
#include "lqg_launch.hpp"

namespace lqg {
namespace host {

// Checkpointed gains trade HBM traffic (and, per step, one dependent gain load) for VALU work in the forward kernel: a win
// for batches (2^20 systems: +5 %; 4096 systems of a small model, config 3: forward sweep 0.55 -> 0.32 ms — the per-step
// gain load's latency disappears), a loss when ONE wave walks a large model alone (every extra instruction is latency:
// config 2, m = 8, forward sweep 0.95 -> 1.26 ms).  Below this many systems the gains stream through HBM as in round 1.
#ifndef LQG_TRIAL_WIDE_RULE
#define LQG_TRIAL_WIDE_RULE 5      // geometry the default rule takes for many trials x many candidates (see trial_sweep_sp)
#endif
#ifndef LQG_SP_CHUNK_MIN_SYS
#define LQG_SP_CHUNK_MIN_SYS 1024
#endif

// The per-trial sweep with the operator's structural zeros compiled out: k_trial_sp in one pass, or — few trials over a
// long horizon — the time-chunked sweep (lqg_trial_chunk.hpp; its scratch follows the operator stream).  Trials-per-lane
// rule of launch_trial (lqg_launch.hpp).
template <typename R, typename PAT, int NX, int NB, int NU, int NY, int ND>
hipError_t trial_sweep_sp_main(const lqg_problem* p, lqg_traj x, void* ll, long ll_sb, long ll_sn, const void* ops, hipStream_t st,
                               const float* ops_lo, const int* hl) {
  const lqg_traj no_traj{nullptr, 0, 0, 0, 0};
  const dim3 block(LQG_BLOCK);
  constexpr auto FM_dense = lqg::trial_operator_mask<PAT, NX, NB, NU, NY, ND, true>();
  constexpr auto FM_noise = lqg::trial_operator_mask<PAT, NX, NB, NU, NY, ND, false>();
  const bool dense_p = p->Sigma0.ptr != nullptr;
  if (trial_chunks(p) > 1) {
    const size_t esz = traj_esz(p);
    const size_t ops_bytes = (size_t)p->n_sys * (size_t)(p->T + 1) * ops_reals(p->dims) * esz;
    void* scratch = static_cast<char*>(const_cast<void*>(ops)) + (ops_bytes + 255) / 256 * 256;
    if constexpr (sizeof(R) == 4) {
      if (hl) {            // MIXED: the chunk kernels apply hi + lo operators to the flagged systems (lqg_trial_chunk.hpp: HLC)
        if (dense_p)
          return launch_trial_chunked<R, NX + NB, ND, lqg::MaskPolicy<NX + NB, FM_dense>, true>(p, ops, x, ll, ll_sb, ll_sn, scratch,
                                                                                                st, ops_lo, hl);
        return launch_trial_chunked<R, NX + NB, ND, lqg::MaskPolicy<NX + NB, FM_noise>, true>(p, ops, x, ll, ll_sb, ll_sn, scratch, st,
                                                                                              ops_lo, hl);
      }
    }
    if (dense_p)
      return launch_trial_chunked<R, NX + NB, ND, lqg::MaskPolicy<NX + NB, FM_dense>>(p, ops, x, ll, ll_sb, ll_sn, scratch, st);
    return launch_trial_chunked<R, NX + NB, ND, lqg::MaskPolicy<NX + NB, FM_noise>>(p, ops, x, ll, ll_sb, ll_sn, scratch, st);
  }
  lqg::TrialArgs<R> tk{dt<R>(x), dt<R>(no_traj), static_cast<R*>(ll), ll_sb, ll_sn, (long)p->n_trials, p->T};
  tk.ops_lo = ops_lo;                 // (MIXED mode: this pass leaves the systems flagged in hl to the hi + lo pass below)
  tk.hl = hl;
  // many trials per candidate and many candidates: one 256-lane workgroup per 1024 trials of a candidate, the operator stream
  // staged in LDS (read once per candidate instead of once per 128 trials; lqg_kernels_sp.hpp k_trial_lds)
  // Many trials per candidate and many candidates: k_trial_sp on WIDER workgroups, so that 1024 trials of a candidate walk its
  // operator stream together through one CU's scalar cache instead of as eight independent 64-lane workgroups.  Measured on
  // BASELINE config 3 (4096 x 1024, per-trial sweep, ms; bitwise identical results): 64 x 2: 3.63 | 512 x 2: 3.37 | 1024 x 2: 6.27 |
  // LDS-staged operators (k_trial_lds, 256 x 4): 4.33.  tuning.trial_lds: 0 rule (256 x 2 from 768 trials per system and 256
  // systems), -1 64-lane workgroups, 1 k_trial_lds, 2..5 the A/B geometries below (second box: 64 x 2: 3.53, 512 x 2: 3.42,
  // 256 x 4: 3.40, 256 x 2: 3.35 -> the rule takes 256 x 2).
  {
    int geo = p->tuning.trial_lds;
    if (geo == 0 && p->n_trials >= 768 && p->n_sys >= 256) geo = LQG_TRIAL_WIDE_RULE;
    if (geo >= 2) {
      const R* o = static_cast<const R*>(ops);
#define LQG_TRIAL_GEO(BLK_, TPL_)                                                                                          \
  do {                                                                                                                    \
    const long per_ = (long)(BLK_) * (TPL_);                                                                              \
    const dim3 gg((unsigned)((p->n_trials + per_ - 1) / per_), (unsigned)p->n_sys);                                       \
    if (dense_p) hipLaunchKernelGGL((lqg::k_trial_sp<R, NX + NB, ND, TPL_, FM_dense, 0, BLK_>), gg, dim3(BLK_), 0, st, o, tk); \
    else hipLaunchKernelGGL((lqg::k_trial_sp<R, NX + NB, ND, TPL_, FM_noise, 0, BLK_>), gg, dim3(BLK_), 0, st, o, tk);    \
  } while (0)
      if (geo == 2) LQG_TRIAL_GEO(512, 2);
      else if (geo == 3) LQG_TRIAL_GEO(1024, 2);
      else if (geo == 4) LQG_TRIAL_GEO(256, 4);
      else LQG_TRIAL_GEO(256, 2);
#undef LQG_TRIAL_GEO
      return hipGetLastError();
    }
  }
  if (p->tuning.trial_lds == 1) {
    const long per = (long)LQG_TRIAL_LDS_BLOCK * LQG_TRIAL_LDS_TPL;
    const dim3 lgrid((unsigned)((p->n_trials + per - 1) / per), (unsigned)p->n_sys), lblock(LQG_TRIAL_LDS_BLOCK);
    const R* o = static_cast<const R*>(ops);
    if (dense_p) hipLaunchKernelGGL((lqg::k_trial_lds<R, NX + NB, ND, LQG_TRIAL_LDS_TPL, FM_dense>), lgrid, lblock, 0, st, o, tk);
    else hipLaunchKernelGGL((lqg::k_trial_lds<R, NX + NB, ND, LQG_TRIAL_LDS_TPL, FM_noise>), lgrid, lblock, 0, st, o, tk);
    return hipGetLastError();
  }
  const long lanes4 = (long)p->n_sys * ((p->n_trials + 4 * LQG_BLOCK - 1) / (4 * LQG_BLOCK)) * LQG_BLOCK;
  const bool wide = lanes4 >= 2L * 1024 * 64;
  const long per_block = (long)LQG_BLOCK * (wide ? LQG_TRIALS_PER_LANE : 1);
  const dim3 tgrid((unsigned)((p->n_trials + per_block - 1) / per_block), (unsigned)p->n_sys);
  const R* o = static_cast<const R*>(ops);
  if (dense_p) {
    if (wide) hipLaunchKernelGGL((lqg::k_trial_sp<R, NX + NB, ND, LQG_TRIALS_PER_LANE, FM_dense>), tgrid, block, 0, st, o, tk);
    else hipLaunchKernelGGL((lqg::k_trial_sp<R, NX + NB, ND, 1, FM_dense>), tgrid, block, 0, st, o, tk);
  } else {
    if (wide) hipLaunchKernelGGL((lqg::k_trial_sp<R, NX + NB, ND, LQG_TRIALS_PER_LANE, FM_noise>), tgrid, block, 0, st, o, tk);
    else hipLaunchKernelGGL((lqg::k_trial_sp<R, NX + NB, ND, 1, FM_noise>), tgrid, block, 0, st, o, tk);
  }
  return hipGetLastError();
}

// ops_lo / hl (MIXED mode, run_sp_mixed): a second launch walks the systems the builder flagged, with hi + lo operators
// (lqg_kernels_sp.hpp: LQG_HILO_MIN); its geometry follows the default rule.
template <typename R, typename PAT, int NX, int NB, int NU, int NY, int ND>
hipError_t trial_sweep_sp(const lqg_problem* p, lqg_traj x, void* ll, long ll_sb, long ll_sn, const void* ops, hipStream_t st,
                          const float* ops_lo = nullptr, const int* hl = nullptr) {
  const hipError_t e = trial_sweep_sp_main<R, PAT, NX, NB, NU, NY, ND>(p, x, ll, ll_sb, ll_sn, ops, st, ops_lo, hl);
  if (e != hipSuccess || !hl || trial_chunks(p) > 1) return e;       // (the time-chunked sweep handled its flagged systems itself)
  if constexpr (sizeof(R) == 4) {
    constexpr auto FM_dense = lqg::trial_operator_mask<PAT, NX, NB, NU, NY, ND, true>();
    constexpr auto FM_noise = lqg::trial_operator_mask<PAT, NX, NB, NU, NY, ND, false>();
    const lqg_traj no_traj{nullptr, 0, 0, 0, 0};
    const bool dense_p = p->Sigma0.ptr != nullptr;
    lqg::TrialArgs<R> tk{dt<R>(x), dt<R>(no_traj), static_cast<R*>(ll), ll_sb, ll_sn, (long)p->n_trials, p->T};
    tk.ops_lo = ops_lo;
    tk.hl = hl;
    const R* o = static_cast<const R*>(ops);
#define LQG_TRIAL_HL(BLK_, TPL_)                                                                                           \
  do {                                                                                                                    \
    const long per_ = (long)(BLK_) * (TPL_);                                                                              \
    const dim3 gg((unsigned)((p->n_trials + per_ - 1) / per_), (unsigned)p->n_sys);                                       \
    if (dense_p) hipLaunchKernelGGL((lqg::k_trial_sp<R, NX + NB, ND, TPL_, FM_dense, 0, BLK_, true>), gg, dim3(BLK_), 0, st, o, tk); \
    else hipLaunchKernelGGL((lqg::k_trial_sp<R, NX + NB, ND, TPL_, FM_noise, 0, BLK_, true>), gg, dim3(BLK_), 0, st, o, tk);    \
  } while (0)
    const long lanes4 = (long)p->n_sys * ((p->n_trials + 4 * LQG_BLOCK - 1) / (4 * LQG_BLOCK)) * LQG_BLOCK;
    if (p->tuning.trial_lds >= 0 && p->n_trials >= 768 && p->n_sys >= 256) LQG_TRIAL_HL(256, 2);
    else if (lanes4 >= 2L * 1024 * 64) LQG_TRIAL_HL(LQG_BLOCK, LQG_TRIALS_PER_LANE);
    else LQG_TRIAL_HL(LQG_BLOCK, 1);
#undef LQG_TRIAL_HL
  }
  return hipGetLastError();
}

template <typename R, typename PAT, int NX, int NB, int NU, int NY, int ND, int CK>
int run_sp_ck(const lqg_problem* p, lqg_traj x, void* ll, long ll_sb, long ll_sn, void* workspace, size_t workspace_bytes,
              hipStream_t st);

template <typename R, typename PAT, int NX, int NB, int NU, int NY, int ND>
int run_sp(const lqg_problem* p, lqg_traj x, void* ll, long ll_sb, long ll_sn, void* workspace, size_t workspace_bytes,
           hipStream_t st) {
  constexpr int CK = lqg::sp_chunk<R, NB, NU>();      // checkpointed gains (lqg_kernels_sp.hpp)
  if constexpr (CK > 0) {
    if (p->n_sys >= LQG_SP_CHUNK_MIN_SYS)
      return run_sp_ck<R, PAT, NX, NB, NU, NY, ND, CK>(p, x, ll, ll_sb, ll_sn, workspace, workspace_bytes, st);
  }
  return run_sp_ck<R, PAT, NX, NB, NU, NY, ND, 0>(p, x, ll, ll_sb, ll_sn, workspace, workspace_bytes, st);
}

template <typename R, typename PAT, int NX, int NB, int NU, int NY, int ND, int CK>
int run_sp_ck(const lqg_problem* p, lqg_traj x, void* ll, long ll_sb, long ll_sn, void* workspace, size_t workspace_bytes,
              hipStream_t st) {
  static_assert(CK == 0 || CK >= 4, "carve() sizes the checkpoint stream for chunks of at least 4 steps");
  const bool fused = p->n_trials <= 2;
  const Workspace w = carve(p, !fused);
  if (!workspace || workspace_bytes < w.total) return LQG_ERR_WORKSPACE;
  char* base = static_cast<char*>(workspace);
  R* Ls = reinterpret_cast<R*>(base + w.ls_off);
  R* ops = fused ? nullptr : reinterpret_cast<R*>(base + w.ops_off);
  auto mark = [&](int i) {
    if (p->phase_events[i]) (void)hipEventRecord(static_cast<hipEvent_t>(p->phase_events[i]), st);
  };
  const lqg_spec& a = p->actor;
  const lqg_spec& d = p->dynamics;
  const lqg_view none{nullptr, 0, 0, 0, 0};
  const lqg_traj no_traj{nullptr, 0, 0, 0, 0};
  const dim3 grid(blocks_for(p->n_sys)), block(LQG_BLOCK);
  mark(0);
  {
    lqg::RiccatiArgs<R> k{dv<R>(a.Q), dv<R>(a.q), dv<R>(a.Qf), dv<R>(a.qf), dv<R>(a.P), dv<R>(a.R), dv<R>(a.r),
                          dv<R>(a.A), dv<R>(a.B), dv<R>(none), dv<R>(none), dv<R>(none), Ls, w.ldb,
                          (long)p->n_sys, p->T, (R)p->eps};
    hipLaunchKernelGGL((lqg::k_riccati_sp<R, NB, NU, PAT, CK>), grid, block, 0, st, k);
  }
  mark(1);
  {
    lqg::ForwardArgs<R> k{dv<R>(a.A), dv<R>(a.B), dv<R>(a.F), dv<R>(a.V), dv<R>(a.W),
                          dv<R>(d.A), dv<R>(d.B), dv<R>(d.F), dv<R>(d.V), dv<R>(d.W),
                          dv<R>(p->Sigma0), Ls, w.ldb, dt<R>(x), static_cast<R*>(ll), ll_sb, ops, dv<R>(none),
                          dt<R>(no_traj), dv<R>(none), (long)p->n_sys, p->T,
                          p->dims.nva, p->dims.nwa, p->dims.nvd, p->dims.nwd};
    // DENSE_P: an explicit Sigma0 may be dense, so the Kalman covariance cannot carry the structural mask derived from V V'
    const lqg::RiccatiArgs<R> rc{dv<R>(a.Q), dv<R>(a.q), dv<R>(a.Qf), dv<R>(a.qf), dv<R>(a.P), dv<R>(a.R), dv<R>(a.r),
                                 dv<R>(a.A), dv<R>(a.B), dv<R>(none), dv<R>(none), dv<R>(none), Ls, w.ldb,
                                 (long)p->n_sys, p->T, (R)p->eps};
#define LQG_SP_FWD(NTR_, DP_)                                                                                      \
  hipLaunchKernelGGL((lqg::k_forward_sp<R, NX, NB, NU, NY, ND, PAT, NTR_, DP_, CK>), grid, block, 0, st,           \
                     k, ll_sn, rc)
    const bool dense_p = p->Sigma0.ptr != nullptr;
    // the lane's data row as one 16-byte vector (fp32, trials x d = 4, rows laid [T+1][system][trial][component])
    bool x4 = false;
    if constexpr (sizeof(R) == 4 && CK > 0) {
      if constexpr (ND == 2 || ND == 4) {
        x4 = !dense_p && p->n_trials * ND == 4 && x.sd == 1 && x.sn == ND && x.sb == 4 && x.st % 4 == 0 &&
             (reinterpret_cast<uintptr_t>(x.ptr) & 15u) == 0;
        if (x4) {
          if constexpr (ND == 2)
            hipLaunchKernelGGL((lqg::k_forward_sp<R, NX, NB, NU, NY, ND, PAT, 2, false, CK, R, true>), grid, block, 0, st, k, ll_sn, rc);
          else
            hipLaunchKernelGGL((lqg::k_forward_sp<R, NX, NB, NU, NY, ND, PAT, 1, false, CK, R, true>), grid, block, 0, st, k, ll_sn, rc);
        }
      }
    }
    if (x4) {}
    else if (p->n_trials == 1) { if (dense_p) LQG_SP_FWD(1, true); else LQG_SP_FWD(1, false); }
    else if (p->n_trials == 2) { if (dense_p) LQG_SP_FWD(2, true); else LQG_SP_FWD(2, false); }
    else { if (dense_p) LQG_SP_FWD(0, true); else LQG_SP_FWD(0, false); }
#undef LQG_SP_FWD
  }
  mark(2);
  if (!fused) {   // several trials per system: the per-trial sweep over the operator stream
    const hipError_t te = trial_sweep_sp<R, PAT, NX, NB, NU, NY, ND>(p, x, ll, ll_sb, ll_sn, ops, st);
    if (te != hipSuccess) return (int)te;
  }
  mark(3);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

// LQG_F32_SYS64 (include/lqg_hip.h): k_riccati_sp / k_forward_sp<NTR = 0> in fp64 over the double spec arrays, the
// operator stream written as float, then the fp32 per-trial sweep.
template <typename PAT, int NX, int NB, int NU, int NY, int ND>
int run_sp_mixed(const lqg_problem* p, lqg_traj x, void* ll, long ll_sb, long ll_sn, void* workspace, size_t workspace_bytes,
                 hipStream_t st) {
  using R = double;
  constexpr int CK = 0;                           // (fp64 sweeps stream their gains: sp_chunk<double>() is 0 too)
  if (p->n_trials < 1) return LQG_ERR_ARG;
  const Workspace w = carve(p, true);
  if (!workspace || workspace_bytes < w.total) return LQG_ERR_WORKSPACE;
  char* base = static_cast<char*>(workspace);
  R* Ls = reinterpret_cast<R*>(base + w.ls_off);
  R* ops = reinterpret_cast<R*>(base + w.ops_off);  // (written as float by the kernel: OT)
  // hi + lo operators for the systems whose Fj - I block is large (lqg_kernels_sp.hpp: LQG_HILO_MIN)
  float* ops_lo = w.lo_off ? reinterpret_cast<float*>(base + w.lo_off) : nullptr;      // (tuning.hilo = -1: no residual stream)
  int* hl = w.lo_off ? reinterpret_cast<int*>(base + w.hl_off) : nullptr;
  auto mark = [&](int i) {
    if (p->phase_events[i]) (void)hipEventRecord(static_cast<hipEvent_t>(p->phase_events[i]), st);
  };
  const lqg_spec& a = p->actor;
  const lqg_spec& d = p->dynamics;
  const lqg_view none{nullptr, 0, 0, 0, 0};
  const lqg_traj no_traj{nullptr, 0, 0, 0, 0};
  const dim3 grid(blocks_for(p->n_sys)), block(LQG_BLOCK);
  const lqg::RiccatiArgs<R> rc{dv<R>(a.Q), dv<R>(a.q), dv<R>(a.Qf), dv<R>(a.qf), dv<R>(a.P), dv<R>(a.R), dv<R>(a.r),
                               dv<R>(a.A), dv<R>(a.B), dv<R>(none), dv<R>(none), dv<R>(none), Ls, w.ldb,
                               (long)p->n_sys, p->T, (R)p->eps};
  mark(0);
  hipLaunchKernelGGL((lqg::k_riccati_sp<R, NB, NU, PAT, CK>), grid, block, 0, st, rc);
  mark(1);
  {
    lqg::ForwardArgs<R> k{dv<R>(a.A), dv<R>(a.B), dv<R>(a.F), dv<R>(a.V), dv<R>(a.W),
                          dv<R>(d.A), dv<R>(d.B), dv<R>(d.F), dv<R>(d.V), dv<R>(d.W),
                          dv<R>(p->Sigma0), Ls, w.ldb, dt<R>(no_traj), nullptr, 0, ops, dv<R>(none),
                          dt<R>(no_traj), dv<R>(none), (long)p->n_sys, p->T,
                          p->dims.nva, p->dims.nwa, p->dims.nvd, p->dims.nwd, ops_lo, hl};
    if (p->Sigma0.ptr)
      hipLaunchKernelGGL((lqg::k_forward_sp<R, NX, NB, NU, NY, ND, PAT, 0, true, CK, float>), grid, block, 0, st, k, ll_sn, rc);
    else
      hipLaunchKernelGGL((lqg::k_forward_sp<R, NX, NB, NU, NY, ND, PAT, 0, false, CK, float>), grid, block, 0, st, k, ll_sn, rc);
  }
  mark(2);
  const hipError_t te = trial_sweep_sp<float, PAT, NX, NB, NU, NY, ND>(p, x, ll, ll_sb, ll_sn, ops, st, ops_lo, hl);
  if (te != hipSuccess) return (int)te;
  mark(3);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

// Specs that vary in TIME under the pattern the library is compiled for (round 6; the reference's data model IS (T, ...)-stacked,
// lqg/spec.py:5-19, lqg/utils.py:10-35): k_riccati_tv_sp -> k_forward_tv_sp load the structurally non-zero entries of every step.
// One trial per system is swept in-lane; several trials (and the mixed mode LQG_F32_SYS64: fp64 system sweeps over the double
// spec arrays, operators rounded once to float, no residual stream) go through the operator stream and the pattern's per-trial
// sweep.  The cross cost P enters G = P + B'SA (lqr.py:23); q, qf, r only move the affine gain l and the offset s of the
// cost-to-go (lqr.py:24, 31, 34), neither of which the moments or the likelihood read (system.py:169-181 use gains.L) — ignored.
// Leading dimension of the CANONICAL storage [T][row][col][system] (workload.pack_systems) when every spec field the two sweeps read
// has it — sb = 1, sc = ld, sr = cols ld, st = rows cols ld (or 0: time-invariant) with one ld — and the 32-bit offsets of the
// canonical loaders (lqg_sparse.hpp) cannot overflow; 0 otherwise (then the strided loaders serve the call).  No cross cost P.
inline long canon_layout(const lqg_problem* p) {
  const lqg_spec& a = p->actor;
  const lqg_spec& d = p->dynamics;
  const lqg_dims& m = p->dims;
  if (a.P.ptr) return 0;
  long ld = 0;
  auto ok = [&](const lqg_view& v, long rows, long cols) {
    if (!v.ptr || v.sb != 1 || v.sc <= 0) return false;
    if (ld == 0) ld = v.sc;
    return v.sc == ld && v.sr == cols * ld && (v.st == rows * cols * ld || v.st == 0);
  };
  const bool all = ok(a.A, m.b, m.b) && ok(a.B, m.b, m.u) && ok(a.F, m.y, m.b) && ok(a.V, m.b, m.nva) && ok(a.W, m.y, m.nwa) &&
                   ok(a.Q, m.b, m.b) && ok(a.R, m.u, m.u) && ok(d.A, m.x, m.x) && ok(d.B, m.x, m.u) && ok(d.F, m.y, m.x) &&
                   ok(d.V, m.x, m.nvd) && ok(d.W, m.y, m.nwd);
  if (!all) return 0;
  const long esz = p->dtype == LQG_F32 ? 4 : 8;
  long widest = 1;
  for (long e : {(long)m.b * m.b, (long)m.b * m.nva, (long)m.y * m.nwa, (long)m.x * m.x, (long)m.x * m.nvd, (long)m.y * m.nwd, (long)m.y * m.b})
    widest = e > widest ? e : widest;
  if (widest * ld * esz >= (1L << 31) || (long)p->n_sys * esz >= (1L << 32) || round_up(p->n_sys, 64) * esz * m.u * m.b >= (1L << 31)) return 0;
  return ld;
}

template <typename PAT, int NX, int NB, int NU, int NY, int ND>
int run_sp_tv(const lqg_problem* p, lqg_traj x, void* ll, long ll_sb, long ll_sn, void* workspace, size_t workspace_bytes,
              hipStream_t st) {
  const bool mixed = p->dtype == LQG_F32_SYS64;
  const long ldc = canon_layout(p);
  const bool fused = p->n_trials == 1 && !mixed;
  const Workspace w = carve(p, !fused);
  if (!workspace || workspace_bytes < w.total) return LQG_ERR_WORKSPACE;
  auto mark = [&](int i) {
    if (p->phase_events[i]) (void)hipEventRecord(static_cast<hipEvent_t>(p->phase_events[i]), st);
  };
  const lqg_spec& a = p->actor;
  const lqg_spec& d = p->dynamics;
  const lqg_view none{nullptr, 0, 0, 0, 0};
  const lqg_traj no_traj{nullptr, 0, 0, 0, 0};
  const dim3 grid(blocks_for(p->n_sys)), block(LQG_BLOCK);
  char* base = static_cast<char*>(workspace);
  auto go = [&](auto tag, auto otag) -> hipError_t {
    using R = decltype(tag);
    using OT = decltype(otag);
    R* Ls = reinterpret_cast<R*>(base + w.ls_off);
    R* ops = fused ? nullptr : reinterpret_cast<R*>(base + w.ops_off);       // (written as OT by the kernel)
    mark(0);
    lqg::RiccatiArgs<R> rk{dv<R>(a.Q), dv<R>(none), dv<R>(a.Qf), dv<R>(none), dv<R>(a.P), dv<R>(a.R), dv<R>(none),
                           dv<R>(a.A), dv<R>(a.B), dv<R>(none), dv<R>(none), dv<R>(none), Ls, w.ldb, (long)p->n_sys,
                           p->T, (R)p->eps};
    if (ldc) hipLaunchKernelGGL((lqg::k_riccati_tv_sp<R, NB, NU, PAT, true>), grid, block, 0, st, rk, ldc);
    else hipLaunchKernelGGL((lqg::k_riccati_tv_sp<R, NB, NU, PAT, false>), grid, block, 0, st, rk, 0L);
    mark(1);
    const lqg::DView<R> Lv{Ls, 1, (long)(NU * NB) * w.ldb, (long)NB * w.ldb, w.ldb};
    lqg::ForwardArgs<R> fk{dv<R>(a.A), dv<R>(a.B), dv<R>(a.F), dv<R>(a.V), dv<R>(a.W),
                           dv<R>(d.A), dv<R>(d.B), dv<R>(d.F), dv<R>(d.V), dv<R>(d.W),
                           dv<R>(p->Sigma0), Ls, w.ldb, dt<R>(fused ? x : no_traj), fused ? static_cast<R*>(ll) : nullptr, ll_sb, ops,
                           dv<R>(none), dt<R>(no_traj), dv<R>(none), (long)p->n_sys, p->T, p->dims.nva, p->dims.nwa,
                           p->dims.nvd, p->dims.nwd, nullptr, nullptr};
    // (the canonical-layout instantiations: the default-Sigma0 ones only — with an explicit Sigma0 the strided loaders serve the call)
#define LQG_TV_FWD(DP_, FU_, OT_, CN_) \
  hipLaunchKernelGGL((lqg::k_forward_tv_sp<R, NX, NB, NU, NY, ND, PAT, DP_, FU_, OT_, CN_>), grid, block, 0, st, fk, Lv, (CN_) ? ldc : 0L)
    if (fused) {
      if constexpr (std::is_same_v<R, OT>) {
        if (p->Sigma0.ptr) LQG_TV_FWD(true, true, R, false);
        else if (ldc) LQG_TV_FWD(false, true, R, true);
        else LQG_TV_FWD(false, true, R, false);
      }
      mark(2);
      mark(3);
      return hipGetLastError();
    }
    if (p->Sigma0.ptr) LQG_TV_FWD(true, false, OT, false);
    else if (ldc) LQG_TV_FWD(false, false, OT, true);
    else LQG_TV_FWD(false, false, OT, false);
#undef LQG_TV_FWD
    mark(2);
    const hipError_t te = trial_sweep_sp<OT, PAT, NX, NB, NU, NY, ND>(p, x, ll, ll_sb, ll_sn, ops, st);
    mark(3);
    return te != hipSuccess ? te : hipGetLastError();
  };
  hipError_t e;
  if (mixed) e = go(double{}, float{});
  else if (p->dtype == LQG_F64) e = go(double{}, double{});
  else e = go(float{}, float{});
  return e == hipSuccess ? 0 : (int)e;
}

template <typename PAT, int NX, int NB, int NU, int NY, int ND>
int log_likelihood_sp(const lqg_problem* p, lqg_traj x, void* ll, int64_t ll_sb, int64_t ll_sn, void* workspace,
                      size_t workspace_bytes, void* stream) {
  if (!p || !x.ptr || !ll) return LQG_ERR_NULL;
  const lqg_dims& dm = p->dims;
  if (dm.x != NX || dm.b != NB || dm.u != NU || dm.y != NY || dm.d != ND) return LQG_ERR_DIMS;
  if (p->n_trials < 1 || p->T < 1) return LQG_ERR_ARG;
  if (p->dtype != LQG_F64 && p->dtype != LQG_F32 && p->dtype != LQG_F32_SYS64) return LQG_ERR_ARG;
  if (p->n_sys == 0) return 0;
  if (!forward_ti(p) || !actor_ti_riccati(p) || affine(p))       // specs that vary in time and / or affine cost terms
    return run_sp_tv<PAT, NX, NB, NU, NY, ND>(p, x, ll, (long)ll_sb, (long)ll_sn, workspace, workspace_bytes, (hipStream_t)stream);
  if (p->dtype == LQG_F64)
    return run_sp<double, PAT, NX, NB, NU, NY, ND>(p, x, ll, (long)ll_sb, (long)ll_sn, workspace, workspace_bytes,
                                                   (hipStream_t)stream);
  if (p->dtype == LQG_F32)
    return run_sp<float, PAT, NX, NB, NU, NY, ND>(p, x, ll, (long)ll_sb, (long)ll_sn, workspace, workspace_bytes,
                                                  (hipStream_t)stream);
  if (p->dtype == LQG_F32_SYS64)
    return run_sp_mixed<PAT, NX, NB, NU, NY, ND>(p, x, ll, (long)ll_sb, (long)ll_sn, workspace, workspace_bytes,
                                                 (hipStream_t)stream);
  return LQG_ERR_ARG;
}

// lqg_solve_materialised (include/lqg_hip.h) for specs that may vary in TIME under the pattern the library is compiled for:
// one trial per system, no affine cost terms; every output view optional.  Refusals as log_likelihood_sp.
template <typename PAT, int NX, int NB, int NU, int NY, int ND>
int solve_materialised_sp(const lqg_problem* p, lqg_traj x, lqg_view L, lqg_view l, lqg_view H, lqg_view K, lqg_traj mu,
                          lqg_view Sigma, void* ll, int64_t ll_sb, void* workspace, size_t workspace_bytes, void* stream) {
  if (!p || !x.ptr) return LQG_ERR_NULL;
  const lqg_dims& dm = p->dims;
  if (dm.x != NX || dm.b != NB || dm.u != NU || dm.y != NY || dm.d != ND) return LQG_ERR_DIMS;
  if (p->n_trials != 1 || p->T < 1 || affine(p)) return LQG_ERR_ARG;
  if (p->dtype != LQG_F32 && p->dtype != LQG_F64) return LQG_ERR_ARG;
  if (p->n_sys == 0) return 0;
  const Workspace w = carve(p, false);
  if (!workspace || workspace_bytes < w.total) return LQG_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  auto mark = [&](int i) {
    if (p->phase_events[i]) (void)hipEventRecord(static_cast<hipEvent_t>(p->phase_events[i]), st);
  };
  const lqg_spec& a = p->actor;
  const lqg_spec& d = p->dynamics;
  const dim3 grid(blocks_for(p->n_sys)), block(LQG_BLOCK);
  auto go = [&](auto tag) {
    using R = decltype(tag);
    R* Ls = reinterpret_cast<R*>(static_cast<char*>(workspace) + w.ls_off);
    mark(0);
    // the materialised L doubles as the forward sweep's gain stream (no second copy in the workspace) when it is requested
    const bool own_l = L.ptr != nullptr;
    lqg::RiccatiArgs<R> rk{dv<R>(a.Q), dv<R>(a.q), dv<R>(a.Qf), dv<R>(a.qf), dv<R>(a.P), dv<R>(a.R), dv<R>(a.r),
                           dv<R>(a.A), dv<R>(a.B), dv<R>(L), dv<R>(l), dv<R>(H), own_l ? nullptr : Ls, w.ldb, (long)p->n_sys,
                           p->T, (R)p->eps};
    const lqg::DView<R> Lv = own_l ? dv<R>(L)
                                   : lqg::DView<R>{Ls, 1, (long)(NU * NB) * w.ldb, (long)NB * w.ldb, w.ldb};
    hipLaunchKernelGGL((lqg::k_riccati_tv_sp<R, NB, NU, PAT>), grid, block, 0, st, rk, 0L);
    mark(1);
    lqg::ForwardArgs<R> fk{dv<R>(a.A), dv<R>(a.B), dv<R>(a.F), dv<R>(a.V), dv<R>(a.W),
                           dv<R>(d.A), dv<R>(d.B), dv<R>(d.F), dv<R>(d.V), dv<R>(d.W),
                           dv<R>(p->Sigma0), Ls, w.ldb, dt<R>(x), static_cast<R*>(ll), (long)ll_sb, nullptr, dv<R>(Sigma),
                           dt<R>(mu), dv<R>(K), (long)p->n_sys, p->T, p->dims.nva, p->dims.nwa, p->dims.nvd, p->dims.nwd};
    if (p->Sigma0.ptr) hipLaunchKernelGGL((lqg::k_forward_tv_sp<R, NX, NB, NU, NY, ND, PAT, true>), grid, block, 0, st, fk, Lv, 0L);
    else hipLaunchKernelGGL((lqg::k_forward_tv_sp<R, NX, NB, NU, NY, ND, PAT, false>), grid, block, 0, st, fk, Lv, 0L);
    mark(2);
    mark(3);
  };
  if (p->dtype == LQG_F64) go(double{});
  else go(float{});
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

// lqg_trial_sweep_fn (include/lqg_hip.h): the sweep alone, over an operator stream some other library produced (the
// time-parallel system sweeps of the main library).  Same refusals as log_likelihood_sp.
template <typename PAT, int NX, int NB, int NU, int NY, int ND>
int trial_sweep_entry(const lqg_problem* p, lqg_traj x, void* ll, int64_t ll_sb, int64_t ll_sn, const void* ops, void* stream) {
  if (!p || !x.ptr || !ll || !ops) return LQG_ERR_NULL;
  const lqg_dims& dm = p->dims;
  if (dm.x != NX || dm.b != NB || dm.u != NU || dm.y != NY || dm.d != ND) return LQG_ERR_DIMS;
  if (p->n_trials < 1 || p->T < 1) return LQG_ERR_ARG;
  if (!forward_ti(p) || !actor_ti_riccati(p) || affine(p)) return LQG_ERR_ARG;
  if (p->n_sys == 0) return 0;
  hipError_t e;
  if (p->dtype == LQG_F64)
    e = trial_sweep_sp<double, PAT, NX, NB, NU, NY, ND>(p, x, ll, (long)ll_sb, (long)ll_sn, ops, (hipStream_t)stream);
  else if (p->dtype == LQG_F32)
    e = trial_sweep_sp<float, PAT, NX, NB, NU, NY, ND>(p, x, ll, (long)ll_sb, (long)ll_sn, ops, (hipStream_t)stream);
  else
    return LQG_ERR_ARG;
  return e == hipSuccess ? 0 : (int)e;
}

}  // namespace host
}  // namespace lqg
