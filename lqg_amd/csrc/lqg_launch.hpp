// lqg_launch.hpp — host-side launchers: lqg_problem -> kernel argument structs -> hipLaunchKernelGGL.
// One function template per kernel family; lqg_inst.hip instantiates them per (dtype, dims) in separate
// translation units (parallel build), lqg_abi.hip dispatches to them.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/lqg_hip.h"
#include "lqg_kernels.hpp"
#include "lqg_trial_chunk.hpp"

namespace lqg {
namespace host {

template <typename R>
lqg::DView<R> dv(const lqg_view& v) {
  return lqg::DView<R>{static_cast<const R*>(v.ptr), (long)v.sb, (long)v.st, (long)v.sr, (long)v.sc};
}
template <typename R>
lqg::DTraj<R> dt(const lqg_traj& v) {
  return lqg::DTraj<R>{static_cast<const R*>(v.ptr), (long)v.sb, (long)v.sn, (long)v.st, (long)v.sd};
}

inline long round_up(long v, long m) { return (v + m - 1) / m * m; }
inline unsigned blocks_for(long n) { return (unsigned)((n + LQG_BLOCK - 1) / LQG_BLOCK); }

inline bool time_invariant(const lqg_view& v, int T) { return v.ptr == nullptr || v.st == 0 || T <= 1; }

inline bool actor_ti_riccati(const lqg_problem* p) {
  const lqg_spec& a = p->actor;
  return time_invariant(a.Q, p->T) && time_invariant(a.q, p->T) && time_invariant(a.P, p->T) &&
         time_invariant(a.R, p->T) && time_invariant(a.r, p->T) && time_invariant(a.A, p->T) &&
         time_invariant(a.B, p->T);
}
inline bool actor_ti_kalman(const lqg_problem* p) {
  const lqg_spec& a = p->actor;
  return time_invariant(a.A, p->T) && time_invariant(a.F, p->T) && time_invariant(a.V, p->T) &&
         time_invariant(a.W, p->T);
}
inline bool forward_ti(const lqg_problem* p) {
  const lqg_spec& a = p->actor;
  const lqg_spec& d = p->dynamics;
  return actor_ti_kalman(p) && time_invariant(a.B, p->T) && time_invariant(d.A, p->T) &&
         time_invariant(d.B, p->T) && time_invariant(d.F, p->T) && time_invariant(d.V, p->T) &&
         time_invariant(d.W, p->T);
}
inline bool affine(const lqg_problem* p) {
  const lqg_spec& a = p->actor;
  return a.q.ptr || a.qf.ptr || a.P.ptr || a.r.ptr;
}

// ---------------------------------------------------------------- workspace carving
struct Workspace {
  size_t ls_off, ls_bytes, ops_off, ops_bytes, total;
  long ldb;
  size_t lo_off, hl_off;      // LQG_F32_SYS64 only (0 otherwise): residual stream of the operator's F block, per-system first step + 1
};
// reals per step of the residual stream (the dense m x m image of the Fj - I block's rounding residual)
inline size_t hilo_reals(const lqg_dims& d) {
  const size_t m = d.x + d.b;
  return (m * m + 3) / 4 * 4;
}
inline size_t ops_reals(const lqg_dims& d) {
  const size_t m = d.x + d.b, o = d.d, rr = m - o;
  const size_t raw = m * m + rr * o + o * (o + 1) / 2 + 1;
  return (raw + 3) / 4 * 4;
}
// element sizes: trajectories / results / operator stream, and spec arrays / gain scratch (they differ for LQG_F32_SYS64)
inline size_t traj_esz(const lqg_problem* p) { return p->dtype == LQG_F64 ? 8 : 4; }
inline size_t spec_esz(const lqg_problem* p) { return p->dtype == LQG_F32 ? 4 : 8; }
inline Workspace carve(const lqg_problem* p, bool need_ops) {
  Workspace w{};
  const size_t esz = traj_esz(p);
  w.ldb = round_up(p->n_sys, 64);
  w.ls_off = 0;
  // gain scratch: L_t per step, or (structure-specialised libraries with checkpointed gains, chunk >= 4) the packed
  // cost-to-go S every chunk — whichever is larger, so that every library agrees on the layout of the workspace
  const size_t per_step = (size_t)p->T * p->dims.u * p->dims.b;
  const size_t ckpt = ((size_t)p->T / 4 + 1) * (size_t)(p->dims.b * (p->dims.b + 1) / 2);
  w.ls_bytes = (per_step > ckpt ? per_step : ckpt) * (size_t)w.ldb * spec_esz(p);
  w.ops_off = (w.ls_bytes + 255) / 256 * 256;
  w.ops_bytes = need_ops ? (size_t)p->n_sys * (size_t)(p->T + 1) * ops_reals(p->dims) * esz : 0;
  // (+ the scratch of the time-chunked per-trial sweep, lqg_trial_chunk.hpp: by convention it FOLLOWS the operator stream)
  w.total = w.ops_off + (w.ops_bytes + 255) / 256 * 256 + (need_ops ? trial_chunk_scratch(p).total : 0);
  if (need_ops && p->dtype == LQG_F32_SYS64 && p->tuning.hilo >= 0) {
    // (tuning.hilo = -1: no residual stream — the caller runs the generic kernels, which never read it, or the stream with the
    // residual would not fit its workspace limit and rounded operators are the better fall-back than leaving the mixed mode)
    // operators rounded ONCE to fp32 carry a systematic error of eps32 |F - I| |state| per step into every trial; where the
    // block is large (point-mass models: whitening gains of 10 .. 70) the per-trial sweep adds the residual back (hi + lo)
    w.lo_off = (w.total + 255) / 256 * 256;
    w.hl_off = w.lo_off + ((size_t)p->n_sys * (size_t)(p->T + 1) * hilo_reals(p->dims) * 4 + 255) / 256 * 256;
    w.total = w.hl_off + ((size_t)p->n_sys * sizeof(int) + 255) / 256 * 256;
  }
  return w;
}

// ---------------------------------------------------------------- launchers (one per kernel family)
template <typename R, int NB, int NU>
hipError_t launch_riccati(const lqg_problem* p, lqg_view L, lqg_view l, lqg_view H, void* Ls, long ldb,
                          hipStream_t st) {
  const lqg_spec& a = p->actor;
  lqg::RiccatiArgs<R> k{dv<R>(a.Q), dv<R>(a.q), dv<R>(a.Qf), dv<R>(a.qf), dv<R>(a.P), dv<R>(a.R), dv<R>(a.r),
                        dv<R>(a.A), dv<R>(a.B), dv<R>(L), dv<R>(l), dv<R>(H), static_cast<R*>(Ls), ldb,
                        (long)p->n_sys, p->T, (R)p->eps};
  const dim3 grid(blocks_for(p->n_sys)), block(LQG_BLOCK);
  const bool ti = actor_ti_riccati(p), af = affine(p);
  if (ti && !af) hipLaunchKernelGGL((lqg::k_riccati<R, NB, NU, true, false>), grid, block, 0, st, k);
  else if (ti && af) hipLaunchKernelGGL((lqg::k_riccati<R, NB, NU, true, true>), grid, block, 0, st, k);
  else if (!ti && !af) hipLaunchKernelGGL((lqg::k_riccati<R, NB, NU, false, false>), grid, block, 0, st, k);
  else hipLaunchKernelGGL((lqg::k_riccati<R, NB, NU, false, true>), grid, block, 0, st, k);
  return hipGetLastError();
}

template <typename R, int NB, int NY>
hipError_t launch_kalman(const lqg_problem* p, lqg_view K, hipStream_t st) {
  const lqg_spec& a = p->actor;
  lqg::KalmanArgs<R> k{dv<R>(a.A), dv<R>(a.F), dv<R>(a.V), dv<R>(a.W), dv<R>(p->Sigma0), dv<R>(K),
                       (long)p->n_sys, p->T, p->dims.nva, p->dims.nwa};
  const dim3 grid(blocks_for(p->n_sys)), block(LQG_BLOCK);
  if (actor_ti_kalman(p)) hipLaunchKernelGGL((lqg::k_kalman<R, NB, NY, true>), grid, block, 0, st, k);
  else hipLaunchKernelGGL((lqg::k_kalman<R, NB, NY, false>), grid, block, 0, st, k);
  return hipGetLastError();
}

// k_forward has 8 instantiations per (dtype, dims): TI x FUSED x MAT.  They are split over four functions (by FUSED
// and TI) so that lqg_inst.hip can compile them in separate translation units (the m = 20 kernels take ~90 s each).
template <typename R, int NX, int NB, int NU, int NY, int ND, bool FUSED, bool TI>
hipError_t launch_forward_v(const lqg::ForwardArgs<R>& k, long n_sys, bool mat, hipStream_t st) {
  const dim3 grid(blocks_for(n_sys)), block(LQG_BLOCK);
  if (mat) hipLaunchKernelGGL((lqg::k_forward<R, NX, NB, NU, NY, ND, TI, FUSED, true>), grid, block, 0, st, k);
  else hipLaunchKernelGGL((lqg::k_forward<R, NX, NB, NU, NY, ND, TI, FUSED, false>), grid, block, 0, st, k);
  return hipGetLastError();
}

template <typename R, int NX, int NB, int NU, int NY, int ND>
hipError_t launch_forward(const lqg_problem* p, const void* Ls, long ldb, bool fused, lqg_traj x, void* ll,
                          long ll_sb, void* ops, lqg_view Sig, lqg_traj mu, lqg_view Kout, hipStream_t st) {
  const lqg_spec& a = p->actor;
  const lqg_spec& d = p->dynamics;
  lqg::ForwardArgs<R> k{dv<R>(a.A), dv<R>(a.B), dv<R>(a.F), dv<R>(a.V), dv<R>(a.W),
                        dv<R>(d.A), dv<R>(d.B), dv<R>(d.F), dv<R>(d.V), dv<R>(d.W),
                        dv<R>(p->Sigma0), static_cast<const R*>(Ls), ldb, dt<R>(x), static_cast<R*>(ll), ll_sb,
                        static_cast<R*>(ops), dv<R>(Sig), dt<R>(mu), dv<R>(Kout), (long)p->n_sys, p->T,
                        p->dims.nva, p->dims.nwa, p->dims.nvd, p->dims.nwd};
  const bool ti = forward_ti(p);
  const bool mat = Sig.ptr || mu.ptr || Kout.ptr;
  const long n = (long)p->n_sys;
  if (fused) return ti ? launch_forward_v<R, NX, NB, NU, NY, ND, true, true>(k, n, mat, st)
                       : launch_forward_v<R, NX, NB, NU, NY, ND, true, false>(k, n, mat, st);
  return ti ? launch_forward_v<R, NX, NB, NU, NY, ND, false, true>(k, n, mat, st)
            : launch_forward_v<R, NX, NB, NU, NY, ND, false, false>(k, n, mat, st);
}

// LQG_F32_SYS64 (include/lqg_hip.h): the forward sweep in fp64 over double spec arrays, the operator stream written as
// float for the fp32 per-trial sweep.  Split by TI like launch_forward_v (separate translation units).
template <int NX, int NB, int NU, int NY, int ND, bool TI>
hipError_t launch_forward_ops32_v(const lqg::ForwardArgs<double>& k, long n_sys, hipStream_t st) {
  const dim3 grid(blocks_for(n_sys)), block(LQG_BLOCK);
  hipLaunchKernelGGL((lqg::k_forward<double, NX, NB, NU, NY, ND, TI, false, false, float>), grid, block, 0, st, k);
  return hipGetLastError();
}
template <int NX, int NB, int NU, int NY, int ND>
hipError_t launch_forward_ops32(const lqg_problem* p, const void* Ls, long ldb, void* ops, hipStream_t st) {
  using R = double;
  const lqg_spec& a = p->actor;
  const lqg_spec& d = p->dynamics;
  const lqg_view none{nullptr, 0, 0, 0, 0};
  const lqg_traj no_traj{nullptr, 0, 0, 0, 0};
  lqg::ForwardArgs<R> k{dv<R>(a.A), dv<R>(a.B), dv<R>(a.F), dv<R>(a.V), dv<R>(a.W),
                        dv<R>(d.A), dv<R>(d.B), dv<R>(d.F), dv<R>(d.V), dv<R>(d.W),
                        dv<R>(p->Sigma0), static_cast<const R*>(Ls), ldb, dt<R>(no_traj), nullptr, 0,
                        static_cast<R*>(ops), dv<R>(none), dt<R>(no_traj), dv<R>(none), (long)p->n_sys, p->T,
                        p->dims.nva, p->dims.nwa, p->dims.nvd, p->dims.nwd};
  return forward_ti(p) ? launch_forward_ops32_v<NX, NB, NU, NY, ND, true>(k, (long)p->n_sys, st)
                       : launch_forward_ops32_v<NX, NB, NU, NY, ND, false>(k, (long)p->n_sys, st);
}

#ifndef LQG_TRIALS_PER_LANE
#define LQG_TRIALS_PER_LANE 2
#endif
template <typename R, int M, int ND>
hipError_t launch_trial(const lqg_problem* p, const void* ops, lqg_traj x, lqg_traj mu, void* ll, long ll_sb,
                        long ll_sn, hipStream_t st) {
  if (!mu.ptr && ll && trial_chunks(p) > 1) {      // few trials, long horizon: the sweep split along time
    const size_t esz = p->dtype == LQG_F64 ? 8 : 4;
    const size_t ops_bytes = (size_t)p->n_sys * (size_t)(p->T + 1) * ops_reals(p->dims) * esz;
    void* scratch = static_cast<char*>(const_cast<void*>(ops)) + (ops_bytes + 255) / 256 * 256;
    return launch_trial_chunked<R, M, ND, lqg::FullMask>(p, ops, x, ll, ll_sb, ll_sn, scratch, st);
  }
  lqg::TrialArgs<R> k{dt<R>(x), dt<R>(mu), static_cast<R*>(ll), ll_sb, ll_sn, (long)p->n_trials, p->T};
  // Trials per lane: a few trials per lane amortise the per-step operator (scalar) loads when there are trials to spare
  // (2 measured best with the pipelined data loads: 2 / 4 / 8 -> 2.20 / 2.31 / 2.57 ms on config 5); with fewer than ~2
  // waves per SIMD at 4 per lane (one system with 10^4..10^5 trials: BASELINE configs 2 and 4) the sweep is
  // latency-bound and one trial per lane puts the most waves in flight.
  const long lanes4 = (long)p->n_sys * ((p->n_trials + 4 * LQG_BLOCK - 1) / (4 * LQG_BLOCK)) * LQG_BLOCK;
  const bool wide = lanes4 >= 2L * 1024 * 64;
  const long per_block = (long)LQG_BLOCK * (wide ? LQG_TRIALS_PER_LANE : 1);
  const dim3 grid((unsigned)((p->n_trials + per_block - 1) / per_block), (unsigned)p->n_sys), block(LQG_BLOCK);
  const R* o = static_cast<const R*>(ops);
  if (wide) {
    if (mu.ptr) hipLaunchKernelGGL((lqg::k_trial<R, M, ND, LQG_TRIALS_PER_LANE, true>), grid, block, 0, st, o, k);
    else hipLaunchKernelGGL((lqg::k_trial<R, M, ND, LQG_TRIALS_PER_LANE, false>), grid, block, 0, st, o, k);
  } else {
    if (mu.ptr) hipLaunchKernelGGL((lqg::k_trial<R, M, ND, 1, true>), grid, block, 0, st, o, k);
    else hipLaunchKernelGGL((lqg::k_trial<R, M, ND, 1, false>), grid, block, 0, st, o, k);
  }
  return hipGetLastError();
}

// eps.ptr == eta.ptr == NULL: the draws are made in-kernel from `seed` (lqg_simulate_rng)
template <typename R, int NX, int NB, int NU, int NY>
hipError_t launch_simulate(const lqg_problem* p, lqg_view L, lqg_view l, lqg_view K, lqg_traj eps, lqg_traj eta,
                           lqg_view x0, lqg_view xhat0, lqg_traj xs, lqg_traj xhat, lqg_traj ys, lqg_traj us,
                           hipStream_t st, unsigned long long seed = 0) {
  const lqg_spec& a = p->actor;
  const lqg_spec& d = p->dynamics;
  lqg::SimArgs<R> k{dv<R>(a.A), dv<R>(a.B), dv<R>(a.F), dv<R>(d.A), dv<R>(d.B), dv<R>(d.F), dv<R>(d.V), dv<R>(d.W),
                    dv<R>(L), dv<R>(l), dv<R>(K), dt<R>(eps), dt<R>(eta), dv<R>(x0), dv<R>(xhat0),
                    dt<R>(xs), dt<R>(xhat), dt<R>(ys), dt<R>(us), (long)p->n_sys, (long)p->n_trials, p->T,
                    p->dims.nvd, p->dims.nwd, seed};
  const dim3 grid(blocks_for(p->n_sys * p->n_trials)), block(LQG_BLOCK);
  if (!eps.ptr && !eta.ptr) hipLaunchKernelGGL((lqg::k_simulate<R, NX, NB, NU, NY, true>), grid, block, 0, st, k);
  else hipLaunchKernelGGL((lqg::k_simulate<R, NX, NB, NU, NY, false>), grid, block, 0, st, k);
  return hipGetLastError();
}

}  // namespace host
}  // namespace lqg
