// lqg_coop_inst.hip — the cooperative kernels of lqg_coop.hpp and their launch wrappers: ONE translation unit for every
// model shape (dimensions are run-time arguments), both dtypes, LDS and global-arena variants, 64- and 256-lane
// workgroups.
// With -DLQG_INST_COOPF="x,b,u,y,d" / -DLQG_INST_COOPR="b,u" (+ -DLQG_INST_F32 / _F64) this file instead compiles ONE
// fixed-dims instantiation of the forward / Riccati kernel (lqg_amd/build.py: one translation unit per shape of
// lqg_dims.def, built in parallel); the base unit dispatches to them and keeps the run-time-dims kernels for the rest.
#include <mutex>
#include <unordered_map>
#include "lqg_coop.hpp"
#include "lqg_coop_launch.hpp"
#include "lqg_launch.hpp"
#ifndef LQG_DIMS_DEF
#define LQG_DIMS_DEF "lqg_dims.def"
#endif
#include LQG_DIMS_DEF

namespace lqg {
namespace host {

namespace {
constexpr size_t kLdsLimit = 160 * 1024 - 512;     // gfx950: 160 KB LDS per CU
constexpr size_t kLdsDefault = 64 * 1024;          // above this the kernel needs the dynamic-LDS attribute raised
#ifndef LQG_COOP_BIG_BLOCK
#define LQG_COOP_BIG_BLOCK 1024
#endif
constexpr int kBigBlock = LQG_COOP_BIG_BLOCK;      // lanes per workgroup of the whole-workgroup (large-matrix) mode

inline size_t esz(const lqg_problem* p) { return p->dtype == LQG_F64 ? 8 : 4; }
inline long ric_reals(const lqg_problem* p) { return coop::riccati_arena_reals(p->dims.b, p->dims.u); }
inline long fwd_reals(const lqg_problem* p, bool kalman_only) {
  return kalman_only ? coop::kalman_arena_reals(p->dims.b, p->dims.y)
                     : coop::forward_arena_reals(p->dims.x, p->dims.b, p->dims.u, p->dims.y, p->dims.d);
}
// WAVES mode (the independent products of a stage on different waves, each spread over one wave's lanes) while a step's
// largest product has at most two elements per lane; larger systems spread every product over the whole workgroup
inline bool waves_for(int elems) { return elems <= 128; }

// (a property of the kernel, not of a launch: raised once per kernel and size, never again from inside a stream capture)
template <typename K>
hipError_t raise_lds(K kernel, size_t bytes) {
  return raise_dynamic_lds(reinterpret_cast<const void*>(kernel), bytes, kLdsDefault);      // lqg_coop_launch.hpp
}

template <typename R>
coop::Args<R> make_args(const lqg_problem* p) {
  const lqg_spec& a = p->actor;
  const lqg_spec& d = p->dynamics;
  coop::Args<R> k{};
  k.aQ = dv<R>(a.Q); k.aq = dv<R>(a.q); k.aQf = dv<R>(a.Qf); k.aqf = dv<R>(a.qf); k.aP = dv<R>(a.P); k.aR = dv<R>(a.R);
  k.ar = dv<R>(a.r); k.aA = dv<R>(a.A); k.aB = dv<R>(a.B); k.aF = dv<R>(a.F); k.aV = dv<R>(a.V); k.aW = dv<R>(a.W);
  k.dA = dv<R>(d.A); k.dB = dv<R>(d.B); k.dF = dv<R>(d.F); k.dV = dv<R>(d.V); k.dW = dv<R>(d.W);
  k.Sigma0 = dv<R>(p->Sigma0);
  k.n_sys = (long)p->n_sys;
  k.T = p->T;
  k.x = p->dims.x; k.b = p->dims.b; k.u = p->dims.u; k.y = p->dims.y; k.d = p->dims.d;
  k.nva = p->dims.nva; k.nwa = p->dims.nwa; k.nvd = p->dims.nvd; k.nwd = p->dims.nwd;
  k.nops = (int)ops_reals(p->dims);
  k.eps = (R)p->eps;
  return k;
}
}  // namespace

}  // namespace host
}  // namespace lqg

namespace lqg {
namespace host {
// fixed-dims launchers (LDS arena, WAVES mode): defined by the per-shape translation units
template <typename R, int CX, int CB, int CU, int CY, int CD>
hipError_t coop_forward_fixed(const coop::Args<R>& k, size_t lds, hipStream_t st) {
  auto kern = coop::k_coop_forward<R, 256, false, true, CX, CB, CU, CY, CD>;
  if (hipError_t e = raise_dynamic_lds(reinterpret_cast<const void*>(kern), lds); e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3((unsigned)k.n_sys), dim3(256), lds, st, k);
  return hipGetLastError();
}
template <typename R, int CB, int CU>
hipError_t coop_riccati_fixed(const coop::Args<R>& k, size_t lds, hipStream_t st) {
  auto kern = coop::k_coop_riccati<R, 256, false, true, CB, CU>;
  if (hipError_t e = raise_dynamic_lds(reinterpret_cast<const void*>(kern), lds); e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3((unsigned)k.n_sys), dim3(256), lds, st, k);
  return hipGetLastError();
}
#if defined(LQG_INST_COOPF)
#ifdef LQG_INST_F32
template hipError_t coop_forward_fixed<float, LQG_INST_COOPF>(const coop::Args<float>&, size_t, hipStream_t);
#endif
#ifdef LQG_INST_F64
template hipError_t coop_forward_fixed<double, LQG_INST_COOPF>(const coop::Args<double>&, size_t, hipStream_t);
#endif
#elif defined(LQG_INST_COOPR)
#ifdef LQG_INST_F32
template hipError_t coop_riccati_fixed<float, LQG_INST_COOPR>(const coop::Args<float>&, size_t, hipStream_t);
#endif
#ifdef LQG_INST_F64
template hipError_t coop_riccati_fixed<double, LQG_INST_COOPR>(const coop::Args<double>&, size_t, hipStream_t);
#endif
#endif
}  // namespace host
}  // namespace lqg

#if !defined(LQG_INST_COOPF) && !defined(LQG_INST_COOPR)
namespace lqg {
namespace host {
#define X(X_, B_, U_, Y_, D_)                                                                                      \
  extern template hipError_t coop_forward_fixed<float, X_, B_, U_, Y_, D_>(const coop::Args<float>&, size_t, hipStream_t); \
  extern template hipError_t coop_forward_fixed<double, X_, B_, U_, Y_, D_>(const coop::Args<double>&, size_t, hipStream_t);
LQG_FORWARD_DIMS(X)
#undef X
#define X(B_, U_)                                                                                              \
  extern template hipError_t coop_riccati_fixed<float, B_, U_>(const coop::Args<float>&, size_t, hipStream_t);  \
  extern template hipError_t coop_riccati_fixed<double, B_, U_>(const coop::Args<double>&, size_t, hipStream_t);
LQG_RICCATI_DIMS(X)
#undef X

bool coop_supported(const lqg_dims& d) {
  return d.x >= 1 && d.b >= 1 && d.u >= 1 && d.y >= 1 && d.u <= coop::kMaxSmall && d.y <= coop::kMaxSmall &&
         d.d <= coop::kMaxSmall;
}

bool coop_fits_lds(const lqg_problem* p, bool kalman_only, bool riccati_only) {
  const size_t ric = kalman_only ? 0 : (size_t)ric_reals(p) * esz(p);
  const size_t fwd = riccati_only ? 0 : (size_t)fwd_reals(p, kalman_only) * esz(p);
  return ric <= kLdsLimit && fwd <= kLdsLimit;
}

size_t coop_arena_bytes(const lqg_problem* p, bool kalman_only, bool riccati_only) {
  if (coop_fits_lds(p, kalman_only, riccati_only)) return 0;
  const long r = kalman_only ? 0 : ric_reals(p), f = riccati_only ? 0 : fwd_reals(p, kalman_only);
  return (size_t)p->n_sys * (size_t)(r > f ? r : f) * esz(p);
}

template <typename R>
hipError_t coop_riccati(const lqg_problem* p, lqg_view L, lqg_view l, lqg_view H, void* Ls, void* arena, hipStream_t st) {
  coop::Args<R> k = make_args<R>(p);
  k.L = dv<R>(L); k.l = dv<R>(l); k.H = dv<R>(H);
  k.Ls = static_cast<R*>(Ls);
  k.ti = actor_ti_riccati(p) ? 1 : 0;
  const long reals = ric_reals(p);
  size_t lds = (size_t)reals * sizeof(R);
  const bool global = lds > kLdsLimit;
  if (global && !arena) return hipErrorInvalidValue;
  k.arena = static_cast<R*>(arena);
  k.arena_reals = reals;
  const dim3 grid((unsigned)p->n_sys);
  const bool waves = waves_for(p->dims.b * p->dims.b);
  // run-time sparsity lists (large models only: the whole-workgroup mode), in LDS behind the arena / alone when it is global
  const bool sparse_off = p->tuning.coop_sparse < 0;
  const size_t lists = (size_t)coop::row_lists_bytes(p->dims.b, p->dims.b);
  k.sparse = (!waves && !sparse_off && p->dims.b <= 255 && (global ? lists : lds + lists) <= kLdsLimit) ? 1 : 0;
  k.lists_bytes = k.sparse ? (long)lists : 0;
  // hybrid placement when the working set exceeds LDS: as much of it as fits stays in LDS, the rest in the global arena
  k.lds_reals = global ? (long)((kLdsLimit - (size_t)k.lists_bytes) / sizeof(R)) : 0;
  const size_t lds_global = (size_t)k.lists_bytes + (size_t)k.lds_reals * sizeof(R);
  if (k.sparse && !global) lds += lists;
  if (waves && !global && k.ti) {             // a fixed-dims instantiation of this shape (time-invariant specs)
#define X(B_, U_) \
  if (p->dims.b == B_ && p->dims.u == U_) return coop_riccati_fixed<R, B_, U_>(k, lds, st);
    LQG_RICCATI_DIMS(X)
#undef X
  }
#define LQG_GO(G_, W_)                                                                     \
  {                                                                                         \
    /* whole-workgroup mode (large matrices): 1024 lanes = 4 waves per SIMD hide the LDS latency of the per-element */ \
    /* chains (one workgroup owns the CU anyway: its arena fills the LDS)                                            */ \
    constexpr int NTH = (W_) ? 256 : kBigBlock;                                             \
    auto kern = coop::k_coop_riccati<R, NTH, G_, W_>;                                       \
    { hipError_t e = raise_lds(kern, (G_) ? lds_global : lds); if (e != hipSuccess) return e; } \
    hipLaunchKernelGGL(kern, grid, dim3(NTH), (G_) ? lds_global : lds, st, k);              \
  }
  if (waves) { if (global) LQG_GO(true, true) else LQG_GO(false, true) }
  else { if (global) LQG_GO(true, false) else LQG_GO(false, false) }
#undef LQG_GO
  return hipGetLastError();
}

template <typename R>
hipError_t coop_forward(const lqg_problem* p, const void* Ls, void* ops, lqg_view Sig, lqg_view K, void* arena,
                        hipStream_t st) {
  coop::Args<R> k = make_args<R>(p);
  k.Sig = dv<R>(Sig); k.K = dv<R>(K);
  k.Ls = const_cast<R*>(static_cast<const R*>(Ls));
  k.ops = static_cast<R*>(ops);
  k.ti = forward_ti(p) ? 1 : 0;
  const bool kalman_only = !ops && !Sig.ptr;
  const long reals = fwd_reals(p, kalman_only);
  size_t lds = (size_t)reals * sizeof(R);
  const bool global = lds > kLdsLimit;
  if (global && !arena) return hipErrorInvalidValue;
  k.arena = static_cast<R*>(arena);
  k.arena_reals = reals;
  const dim3 grid((unsigned)p->n_sys);
  const int m = p->dims.x + p->dims.b;
  const bool waves = waves_for(kalman_only ? p->dims.b * p->dims.b : m * m);
  const bool sparse_off = p->tuning.coop_sparse < 0;
  const size_t lists = (size_t)coop::row_lists_bytes(p->dims.b, p->dims.b) + (size_t)coop::row_lists_bytes(m, m - p->dims.d);
  k.sparse = (!waves && !sparse_off && m <= 255 && (global ? lists : lds + lists) <= kLdsLimit) ? 1 : 0;
  k.lists_bytes = k.sparse ? (long)lists : 0;
  k.lds_reals = global ? (long)((kLdsLimit - (size_t)k.lists_bytes) / sizeof(R)) : 0;
  const size_t lds_global = (size_t)k.lists_bytes + (size_t)k.lds_reals * sizeof(R);
  if (k.sparse && !global) lds += lists;
  if (waves && !global && k.ti && !kalman_only) {
    const lqg_dims& d = p->dims;
#define X(X_, B_, U_, Y_, D_)                                                  \
  if constexpr ((X_ + B_) * (X_ + B_) <= 128) {                                 \
    if (d.x == X_ && d.b == B_ && d.u == U_ && d.y == Y_ && d.d == D_)         \
      return coop_forward_fixed<R, X_, B_, U_, Y_, D_>(k, lds, st);             \
  }
    LQG_FORWARD_DIMS(X)
#undef X
  }
#define LQG_GO(G_, W_)                                                                     \
  {                                                                                         \
    constexpr int NTH = (W_) ? 256 : kBigBlock;                                             \
    auto kern = coop::k_coop_forward<R, NTH, G_, W_>;                                       \
    { hipError_t e = raise_lds(kern, (G_) ? lds_global : lds); if (e != hipSuccess) return e; } \
    hipLaunchKernelGGL(kern, grid, dim3(NTH), (G_) ? lds_global : lds, st, k);              \
  }
  if (waves) { if (global) LQG_GO(true, true) else LQG_GO(false, true) }
  else { if (global) LQG_GO(true, false) else LQG_GO(false, false) }
#undef LQG_GO
  return hipGetLastError();
}

template <typename R>
hipError_t coop_trial(const lqg_problem* p, const void* ops, lqg_traj x, lqg_traj mu, void* ll, long ll_sb, long ll_sn,
                      hipStream_t st) {
  const int m = p->dims.x + p->dims.b, o = p->dims.d, rr = m - o;
  coop::TrialArgsRT<R> k{dt<R>(x), dt<R>(mu), static_cast<R*>(ll), ll_sb, ll_sn, (long)p->n_trials, p->T, m, o,
                         (int)ops_reals(p->dims)};
  // large joint dimension: a group of threads per trial splits the rows of the mean update (k_coop_trial_rows); trials
  // per block: the smallest power of two that keeps the grid within ~2048 workgroups, at most 64 and what LDS holds —
  // and, while a trial's group would have more threads than the update has rows, twice as many again: every workgroup streams
  // the whole operator block of every step (m^2 reals: 35 KB at m = 65 in fp64), so trials that share a block share that
  // traffic (13 delay-12 systems x 50 trials: 650 workgroups of one trial, 65 of 256 threads busy, 5.0 ms -> 325 of two)
  {
    constexpr int BR = 256, MAXPF = 24;
    const int nops = (int)ops_reals(p->dims);
    const size_t lists_lds = ((size_t)m + (size_t)m * m + 7) / 8 * 8;
    auto rows_lds = [&](int t) { return ((size_t)2 * nops + (size_t)t * 2 * m) * sizeof(R) + lists_lds; };
    // most trials per workgroup: 64 (measured, delay-12 model, per-trial sweep in ms at a cap of 16 / 32 / 64 / 128 — 4096 candidates x
    // 120 trials fp32: 135 / 135 / 108 / 172; 512 x 120 fp64: 47.5 / 33 / 33 / 33; smaller batches never reach the cap:
    // scripts/coop_trial_tpb.py)
    // — and 128 on 1024-thread workgroups only where even 64 per workgroup leave more than ~4096 of them (4096 x 120: 108 -> 83 ms;
    // on every smaller batch the wide workgroups lose: 512 x 120 17.1 -> 22.7 ms)
    const long groups = (long)p->n_sys * p->n_trials;
    const int cap = p->tuning.coop_trial_tpb > 0 ? p->tuning.coop_trial_tpb : (groups / 64 > 4096 ? 128 : 64);
    auto trials_per_block = [&](long groups_of_one, long trials) {
      int t = 1;
      while (t < cap && rows_lds(2 * t) <= kLdsLimit && (groups_of_one / t > 2048 || (2L * t <= trials && BR / (2 * t) >= m))) t *= 2;
      return t;
    };
    const int tpb = trials_per_block((long)p->n_sys * p->n_trials, p->n_trials);
    // (LDS: two operator blocks, the trials' vectors, the row lists of the mean-update block: m + m^2 bytes)
    const size_t lds_r = rows_lds(tpb);
    const bool rows_off = p->tuning.coop_trial_rows < 0;
    if (!rows_off && m >= 16 && o <= 6 && (nops + BR - 1) / BR <= MAXPF && lds_r <= kLdsLimit) {
      auto kr = coop::k_coop_trial_rows<R, BR>;
      hipError_t er = raise_lds(kr, lds_r);
      if (er != hipSuccess) return er;
      // the scratch of this sweep follows the operator stream (carve(), scan_plan()): chunk states, then the row lists
      const TrialChunkScratch sc = trial_chunk_scratch(p);
      const size_t ops_bytes = ((size_t)p->n_sys * (size_t)(p->T + 1) * (size_t)nops * sizeof(R) + 255) / 256 * 256;
      char* base = const_cast<char*>(static_cast<const char*>(ops)) + ops_bytes;
      // run-time sparsity of the mean update (delay augmentations: 9 % of m^2): listed once per call from the whole stream
      const long lists_stride = (long)trial_row_lists_bytes(p->dims);
      unsigned char* lists = nullptr;
      if (lists_stride > 0 && p->tuning.coop_sparse >= 0 && p->T >= 1) {
        lists = reinterpret_cast<unsigned char*>(base + sc.lists_off);
        er = hipMemsetAsync(lists, 0, (size_t)p->n_sys * (size_t)lists_stride, st);
        if (er != hipSuccess) return er;
        // (the horizon in slices while the systems alone leave the chip empty: one system of T = 500 in 63 slices of 8 steps)
        long slices = 2048 / (long)p->n_sys;
        slices = slices < 1 ? 1 : slices > (p->T + 7) / 8 ? (p->T + 7) / 8 : slices;
        const int per = (int)((p->T + slices - 1) / slices);
        hipLaunchKernelGGL((coop::k_coop_trial_flags<R, BR>), dim3((unsigned)p->n_sys, (unsigned)((p->T + per - 1) / per)), dim3(BR), 0, st,
                           static_cast<const R*>(ops), lists, lists_stride, p->T, m, nops, per);
        hipLaunchKernelGGL((coop::k_coop_trial_lists<128>), dim3((unsigned)p->n_sys), dim3(128), 0, st, lists, lists_stride, m);
      }
      // time-chunked (few trials, long horizon, log-likelihood only)
      const int nc = m > kLaneTrialMaxJoint ? trial_chunks(p) : 1;
      if (nc > 1 && ll && !mu.ptr) {
        coop::TrialChunkRT<R> ch{1, nc, trial_chunk_len(p, nc), reinterpret_cast<R*>(base + sc.state_off),
                                 reinterpret_cast<R*>(base + sc.phi_off), reinterpret_cast<double*>(base + sc.part_off)};
        // trials per block of the chunked passes (same rule, their grids)
        const int tz = trials_per_block((long)p->n_sys * (nc - 1) * (p->n_trials + m), p->n_trials + m);
        const size_t lds_z = rows_lds(tz);
        er = raise_lds(kr, lds_z > lds_r ? lds_z : lds_r);
        if (er != hipSuccess) return er;
        const R* o_ = static_cast<const R*>(ops);
        hipLaunchKernelGGL(kr, dim3((unsigned)((p->n_trials + m + tz - 1) / tz), (unsigned)p->n_sys, (unsigned)(nc - 1)), dim3(BR),
                           lds_z, st, o_, k, tz, ch, lists, lists_stride);
        if (nc > 2)
          hipLaunchKernelGGL(coop::k_coop_trial_fix<R>, dim3((unsigned)p->n_trials, (unsigned)p->n_sys), dim3(128),
                             (size_t)2 * m * sizeof(R), st, static_cast<const R*>(ch.phi), ch.state, (long)p->n_trials, nc - 1, m);
        ch.mode = 2;
        hipLaunchKernelGGL(kr, dim3((unsigned)((p->n_trials + tz - 1) / tz), (unsigned)p->n_sys, (unsigned)nc), dim3(BR), lds_z, st,
                           o_, k, tz, ch, lists, lists_stride);
        hipLaunchKernelGGL((lqg::k_trial_sum<R>), dim3((unsigned)((p->n_trials + LQG_BLOCK - 1) / LQG_BLOCK), (unsigned)p->n_sys),
                           dim3(LQG_BLOCK), 0, st, static_cast<const double*>(ch.part), static_cast<R*>(ll), ll_sb, ll_sn,
                           (long)p->n_trials, nc);
        return hipGetLastError();
      }
      const dim3 gr((unsigned)((p->n_trials + tpb - 1) / tpb), (unsigned)p->n_sys);
      // 128 trials per workgroup (the largest batches; fp64: from 32): the same sweep on 1024 threads with two operator blocks in
      // flight — 8 instead of 2 threads per trial, four times the waves per SIMD behind the dependent LDS reads of the list walk
      // (512 candidates x 120 trials fp64: 33.2 -> 23.2 ms; fp32 17.2 -> 18.4: not taken there)
      const int wide_mode = p->tuning.coop_trial_wide;               // 0 rule, 1 always, -1 never
      if (wide_mode >= 0 && (wide_mode > 0 || tpb >= 128 || (sizeof(R) == 8 && tpb >= 32))) {
        constexpr int BW = 1024;
        auto kw = coop::k_coop_trial_rows<R, BW>;
        er = raise_lds(kw, lds_r);
        if (er != hipSuccess) return er;
        hipLaunchKernelGGL(kw, gr, dim3(BW), lds_r, st, static_cast<const R*>(ops), k, tpb, coop::TrialChunkRT<R>{0, 1, 0, nullptr, nullptr, nullptr},
                           lists, lists_stride);
        return hipGetLastError();
      }
      hipLaunchKernelGGL(kr, gr, dim3(BR), lds_r, st, static_cast<const R*>(ops), k, tpb, coop::TrialChunkRT<R>{0, 1, 0, nullptr, nullptr, nullptr},
                         lists, lists_stride);
      return hipGetLastError();
    }
  }
  constexpr int B = 64;
  const size_t lds = (size_t)B * (size_t)(4 * o + 2 * rr + m) * sizeof(R);
  if (lds > kLdsLimit) return hipErrorInvalidValue;
  auto kern = coop::k_coop_trial<R, B>;
  hipError_t e = raise_lds(kern, lds);
  if (e != hipSuccess) return e;
  const dim3 grid((unsigned)((p->n_trials + B - 1) / B), (unsigned)p->n_sys);
  hipLaunchKernelGGL(kern, grid, dim3(B), lds, st, static_cast<const R*>(ops), k);
  return hipGetLastError();
}

template <typename R>
hipError_t coop_simulate(const lqg_problem* p, lqg_view L, lqg_view l, lqg_view K, lqg_traj eps, lqg_traj eta, lqg_view x0,
                         lqg_view xhat0, lqg_traj xs, lqg_traj xhat, lqg_traj ys, lqg_traj us, hipStream_t st,
                         unsigned long long seed) {
  const lqg_spec& a = p->actor;
  const lqg_spec& d = p->dynamics;
  lqg::SimArgs<R> k{dv<R>(a.A), dv<R>(a.B), dv<R>(a.F), dv<R>(d.A), dv<R>(d.B), dv<R>(d.F), dv<R>(d.V), dv<R>(d.W),
                    dv<R>(L), dv<R>(l), dv<R>(K), dt<R>(eps), dt<R>(eta), dv<R>(x0), dv<R>(xhat0),
                    dt<R>(xs), dt<R>(xhat), dt<R>(ys), dt<R>(us), (long)p->n_sys, (long)p->n_trials, p->T,
                    p->dims.nvd, p->dims.nwd, seed};
  constexpr int B = 64;
  const lqg_dims& dm = p->dims;
  const size_t lds = (size_t)B * (size_t)(2 * dm.x + 2 * dm.b + dm.u + dm.y) * sizeof(R);
  if (lds > kLdsLimit) return hipErrorInvalidValue;
  const bool rng = !eps.ptr && !eta.ptr;                     // draws made in-kernel from `seed` (lqg_simulate_rng)
  auto kern = rng ? coop::k_coop_simulate<R, B, true> : coop::k_coop_simulate<R, B, false>;
  hipError_t e = raise_lds(kern, lds);
  if (e != hipSuccess) return e;
  const long lanes = (long)p->n_sys * (long)p->n_trials;
  hipLaunchKernelGGL(kern, dim3((unsigned)((lanes + B - 1) / B)), dim3(B), lds, st, k, dm.x, dm.b, dm.u, dm.y);
  return hipGetLastError();
}

template hipError_t coop_simulate<float>(const lqg_problem*, lqg_view, lqg_view, lqg_view, lqg_traj, lqg_traj, lqg_view,
                                         lqg_view, lqg_traj, lqg_traj, lqg_traj, lqg_traj, hipStream_t, unsigned long long);
template hipError_t coop_simulate<double>(const lqg_problem*, lqg_view, lqg_view, lqg_view, lqg_traj, lqg_traj, lqg_view,
                                          lqg_view, lqg_traj, lqg_traj, lqg_traj, lqg_traj, hipStream_t, unsigned long long);
template hipError_t coop_riccati<float>(const lqg_problem*, lqg_view, lqg_view, lqg_view, void*, void*, hipStream_t);
template hipError_t coop_riccati<double>(const lqg_problem*, lqg_view, lqg_view, lqg_view, void*, void*, hipStream_t);
template hipError_t coop_forward<float>(const lqg_problem*, const void*, void*, lqg_view, lqg_view, void*, hipStream_t);
template hipError_t coop_forward<double>(const lqg_problem*, const void*, void*, lqg_view, lqg_view, void*, hipStream_t);
template hipError_t coop_trial<float>(const lqg_problem*, const void*, lqg_traj, lqg_traj, void*, long, long, hipStream_t);
template hipError_t coop_trial<double>(const lqg_problem*, const void*, lqg_traj, lqg_traj, void*, long, long, hipStream_t);

}  // namespace host
}  // namespace lqg
#endif  // base translation unit

#if defined(LQG_COOP_STAMP) && !defined(LQG_INST_COOPF) && !defined(LQG_INST_COOPR)
extern "C" int lqg_debug_coop_stamps(unsigned long long* out16, int reset) {
  (void)hipDeviceSynchronize();
  hipError_t e = hipMemcpyFromSymbol(out16, HIP_SYMBOL(lqg::coop::g_coop_stamps), 16 * sizeof(unsigned long long));
  if (reset) {
    unsigned long long z[16] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(lqg::coop::g_coop_stamps), z, sizeof(z));
  }
  return (int)e;
}
#endif
