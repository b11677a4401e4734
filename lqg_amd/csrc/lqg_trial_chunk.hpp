// lqg_trial_chunk.hpp — the per-trial sweep (lqg/system.py:219-221 mean recursion + :244-248 log_prob) split along TIME.
//
// k_trial walks one trial per lane through all T steps.  With few trials (one parameter vector x tens..thousands of
// trials: the inner loop of MLE / NUTS, BASELINE configs 2 and 4) that is one dependent chain of T steps on a handful of
// waves — a latency-bound sweep on a nearly empty chip.  The recursion of the conditional mean is AFFINE in its state
// s_t = (dO_t, muR_t) with operators that do not depend on the trial,
//     s_{t+1} = Phi_t s_t + g_t(x_t, x_{t-1}),        Phi_t = [0 ; I_r] [ -U2 Li | I ] + Fm[:, O:] [ -U2 Li | I ],  Fm = Fj - I,
// so the horizon is cut into chunks that run side by side:
//   1. k_trial_zs   one lane per (trial, chunk): the chunk's recursion from s = 0 (zero-state response z_c); one extra
//                   block per (system, chunk) pushes the M unit vectors through the chunk with the data set to zero —
//                   the chunk's transition matrix Phi_c, column by column.  No densities are evaluated.
//   2. k_trial_fix  one lane per trial walks the chunk boundaries: s_{c+1} = Phi_c s_c + z_c  (n_chunks - 1 mat-vecs).
//   3. k_trial_ll   one lane per (trial, chunk): the chunk again, now from its true start state, evaluating the
//                   densities of its steps exactly as k_trial does; per-chunk partial sums in fp64.
//   4. k_trial_sum  log-likelihood of a trial = sum of its chunks' partial sums, in chunk order (deterministic).
// The dependent chain is 2 T / n_chunks + n_chunks steps instead of T, and the lanes in flight grow n_chunks-fold for
// 1.6x the arithmetic.  Results differ from the one-pass sweep by rounding only (the start state of a chunk is
// Phi_c s_c + z_c instead of the step-by-step value).
// The structural zeros of the operator are a compile-time policy (FMP::at(i, j)): FullMask for the generic library,
// the pattern's operator mask in a structure-specialised one.
#pragma once
#include <cmath>
#include <cstdlib>

#include "lqg_kernels.hpp"

namespace lqg {

struct FullMask {
  static constexpr bool at(int, int) { return true; }
};

template <typename R>
struct TrialChunkArgs {
  DTraj<R> x;
  long n_trials;
  int T, n_chunks, chunk_len;
  R* state;          // [sys][n_chunks-1][M][n_trials]: zero-state ends, then (k_trial_fix) start states of chunk c+1
  R* phi;            // [sys][n_chunks-1][M][M]
  double* part;      // [sys][n_chunks][n_trials]
  // MIXED mode (ForwardArgs::ops_lo / hl in lqg_kernels.hpp), kernels compiled with HLC: the systems flagged in hl add the
  // rounding residual of the operator's Fj - I block to the mean updates from step hl[sys] - 1 on (hi + lo), unit-vector
  // pushes included
  const float* ops_lo;
  const int* hl;
};

template <typename R, int M, class FMP, int I, int J>
LQG_DEV void tc_lo_term(const float* __restrict__ lo, const R (&cv)[M], R& v) {
  if constexpr (J < M) {
    if constexpr (FMP::at(I, J)) v += (R)lo[I * M + J] * cv[J];
    tc_lo_term<R, M, FMP, I, J + 1>(lo, cv, v);
  }
}
template <typename R, int M, class FMP, int I>
LQG_DEV void tc_lo_rows(const float* __restrict__ lo, const R (&cv)[M], R (&mn)[M]) {
  if constexpr (I < M) {
    R v = R(0);
    tc_lo_term<R, M, FMP, I, 0>(lo, cv, v);
    mn[I] += v;
    tc_lo_rows<R, M, FMP, I + 1>(lo, cv, mn);
  }
}

template <typename R, int M, int ND, class FMP, int I, int J>
LQG_DEV void tc_mean_term(const R* __restrict__ op, const R (&cv)[M], R& v) {
  if constexpr (J < M) {
    if constexpr (FMP::at(I, J)) v += op[TrialOps<M, ND>::F_OFF + I * M + J] * cv[J];
    tc_mean_term<R, M, ND, FMP, I, J + 1>(op, cv, v);
  }
}
template <typename R, int M, int ND, class FMP, int I>
LQG_DEV void tc_mean_rows(const R* __restrict__ op, const R (&cv)[M], R (&mn)[M]) {
  if constexpr (I < M) {
    R v = R(0);
    tc_mean_term<R, M, ND, FMP, I, 0>(op, cv, v);
    mn[I] = v;
    tc_mean_rows<R, M, ND, FMP, I + 1>(op, cv, mn);
  }
}

// one step of the recursion: whitened innovation w (and its square zz), then — when `update` — the new state
// (lo: the step's residual block, or null — tested only in kernels compiled with HLC)
template <typename R, int M, int ND, class FMP, bool HLC = false>
LQG_DEV R tc_step(const R* __restrict__ op, const R (&xt)[ND], R (&xprev)[ND], R (&dO)[ND], R (&muR)[M - ND], bool update,
                  const float* __restrict__ lo = nullptr) {
  constexpr int O = ND, RR = M - ND;
  using Ops = TrialOps<M, ND>;
  R cv[M], w[O];
  LQG_UNROLL for (int i = 0; i < O; ++i) cv[i] = xt[i];
  R zz = R(0);
  {
    int e = 0;
    LQG_UNROLL for (int i = 0; i < O; ++i) {
      R v = R(0);
      LQG_UNROLL for (int j = 0; j <= i; ++j) v += op[Ops::L_OFF + (e++)] * ((cv[j] - xprev[j]) - dO[j]);
      w[i] = v;
      zz += v * v;
    }
  }
  if (update) {
    LQG_UNROLL for (int p = 0; p < RR; ++p) {
      R v = muR[p];
      LQG_UNROLL for (int j = 0; j < O; ++j) v += op[Ops::U_OFF + p * O + j] * w[j];
      cv[O + p] = v;
    }
    R mn[M];
    tc_mean_rows<R, M, ND, FMP, 0>(op, cv, mn);
    if constexpr (HLC) {
      if (lo) tc_lo_rows<R, M, FMP, 0>(lo, cv, mn);
    }
    LQG_UNROLL for (int i = 0; i < O; ++i) { dO[i] = mn[i]; xprev[i] = cv[i]; }
    LQG_UNROLL for (int p = 0; p < RR; ++p) muR[p] = cv[O + p] + mn[O + p];      // (the stream holds Fj - I)
  }
  return zz;
}

// grid: (trial blocks + 1, n_sys, n_chunks - 1); the last block in x carries the unit vectors.  TPL trials per lane share
// the step's operator loads (the passes wait on their scalar loads more than half of the time: SQ_WAIT_ANY / SQ_WAVE_CYCLES =
// 0.57 at one trial per lane, scripts/pmc_trial.sh).
template <typename R, int M, int ND, class FMP, int TPL, bool HLC = false>
__global__ void __launch_bounds__(LQG_BLOCK) k_trial_zs(const R* __restrict__ ops_all, const TrialChunkArgs<R> a) {
  constexpr int O = ND, RR = M - ND;
  using Ops = TrialOps<M, ND>;
  const long sys = blockIdx.y;
  const int c = blockIdx.z;
  const bool hom = blockIdx.x == gridDim.x - 1;
  const long n0 = hom ? (long)threadIdx.x : (long)blockIdx.x * (LQG_BLOCK * TPL) + threadIdx.x;
  const int t0 = c * a.chunk_len, t1 = t0 + a.chunk_len;               // (c <= n_chunks - 2: the chunk is complete)
  const R* __restrict__ op = ops_all + (sys * (long)(a.T + 1) + t0) * Ops::N;
  [[maybe_unused]] const float* lo = nullptr;
  [[maybe_unused]] int lo_from = 0;                      // (hl[sys] = 1 + the first step with a residual block, 0 = none)
  if constexpr (HLC) {
    if (a.hl && a.hl[sys] != 0) {
      lo = a.ops_lo + (sys * (long)(a.T + 1) + t0) * hilo_len<M>();
      lo_from = a.hl[sys] - 1;
    }
  }
  const R* xr[TPL];
  bool live[TPL];
  R xprev[TPL][O], dO[TPL][O], muR[TPL][RR], xq[TPL][O];
  LQG_UNROLL for (int k = 0; k < TPL; ++k) {
    const long n = n0 + (long)k * LQG_BLOCK;
    live[k] = hom ? (k == 0 && n < (long)M) : (n < a.n_trials);
    xr[k] = a.x.p + sys * a.x.sb + ((live[k] && !hom) ? n : 0) * a.x.sn;
    LQG_UNROLL for (int i = 0; i < O; ++i) {
      xprev[k][i] = hom ? R(0) : xr[k][(long)(t0 > 0 ? t0 - 1 : 0) * a.x.st + i * a.x.sd];
      xq[k][i] = hom ? R(0) : xr[k][(long)t0 * a.x.st + i * a.x.sd];
      dO[k][i] = (hom && k == 0 && n == i) ? R(1) : R(0);
    }
    LQG_UNROLL for (int p = 0; p < RR; ++p) muR[k][p] = (hom && k == 0 && n == O + p) ? R(1) : R(0);
  }
  for (int t = t0; t < t1; ++t) {
    const long row = (t + 1 < a.T) ? (long)(t + 1) : (long)a.T;          // data row one step ahead
    LQG_UNROLL for (int k = 0; k < TPL; ++k) {
      R xt[O];
      LQG_UNROLL for (int i = 0; i < O; ++i) xt[i] = xq[k][i];
      LQG_UNROLL for (int i = 0; i < O; ++i) xq[k][i] = hom ? R(0) : xr[k][row * a.x.st + i * a.x.sd];
      (void)tc_step<R, M, ND, FMP, HLC>(op, xt, xprev[k], dO[k], muR[k], true, t >= lo_from ? lo : nullptr);
    }
    op += Ops::N;
    if constexpr (HLC) {
      if (lo) lo += hilo_len<M>();
    }
  }
  const long slot = sys * (a.n_chunks - 1) + c;
  LQG_UNROLL for (int k = 0; k < TPL; ++k) {
    if (!live[k]) continue;
    const long n = n0 + (long)k * LQG_BLOCK;
    if (hom) {
      R* ph = a.phi + slot * (M * M) + n;
      LQG_UNROLL for (int i = 0; i < O; ++i) ph[i * M] = dO[k][i];
      LQG_UNROLL for (int p = 0; p < RR; ++p) ph[(O + p) * M] = muR[k][p];
    } else {
      R* sp = a.state + slot * M * a.n_trials + n;
      LQG_UNROLL for (int i = 0; i < O; ++i) sp[i * a.n_trials] = dO[k][i];
      LQG_UNROLL for (int p = 0; p < RR; ++p) sp[(O + p) * a.n_trials] = muR[k][p];
    }
  }
}

// grid: (trial blocks, n_sys).  Slot c of `state` holds the zero-state end of chunk c on entry and the start state of
// chunk c + 1 on return (slot 0 is both: chunk 0 starts from zero).
template <typename R, int M>
__global__ void __launch_bounds__(LQG_BLOCK) k_trial_fix(const R* __restrict__ phi_all, R* state, long n_trials, int n_slots) {
  const long sys = blockIdx.y;
  const long n = (long)blockIdx.x * LQG_BLOCK + threadIdx.x;
  if (n >= n_trials) return;
  R* sp = state + sys * n_slots * M * n_trials + n;
  const R* __restrict__ ph = phi_all + sys * n_slots * (M * M);
  R s[M];
  LQG_UNROLL for (int i = 0; i < M; ++i) s[i] = sp[i * n_trials];
  for (int c = 1; c < n_slots; ++c) {
    sp += M * n_trials;
    ph += M * M;
    R z[M];
    LQG_UNROLL for (int i = 0; i < M; ++i) z[i] = sp[i * n_trials];
    LQG_UNROLL for (int i = 0; i < M; ++i) {
      R v = z[i];
      LQG_UNROLL for (int j = 0; j < M; ++j) v += ph[i * M + j] * s[j];
      z[i] = v;
    }
    LQG_UNROLL for (int i = 0; i < M; ++i) { s[i] = z[i]; sp[i * n_trials] = z[i]; }
  }
}

// grid: (trial blocks, n_sys, n_chunks)
template <typename R, int M, int ND, class FMP, int TPL, bool HLC = false>
__global__ void __launch_bounds__(LQG_BLOCK) k_trial_ll(const R* __restrict__ ops_all, const TrialChunkArgs<R> a) {
  constexpr int O = ND, RR = M - ND;
  constexpr int kAccChunk = 8;
  using Ops = TrialOps<M, ND>;
  const long sys = blockIdx.y;
  const int c = blockIdx.z;
  const long n0 = (long)blockIdx.x * (LQG_BLOCK * TPL) + threadIdx.x;
  const int t0 = c * a.chunk_len;
  const bool last = c == a.n_chunks - 1;
  const int t1 = last ? a.T : t0 + a.chunk_len;
  const R* __restrict__ op = ops_all + (sys * (long)(a.T + 1) + t0) * Ops::N;
  [[maybe_unused]] const float* lo = nullptr;
  [[maybe_unused]] int lo_from = 0;                      // (hl[sys] = 1 + the first step with a residual block, 0 = none)
  if constexpr (HLC) {
    if (a.hl && a.hl[sys] != 0) {
      lo = a.ops_lo + (sys * (long)(a.T + 1) + t0) * hilo_len<M>();
      lo_from = a.hl[sys] - 1;
    }
  }
  const R* xr[TPL];
  bool live[TPL];
  R xprev[TPL][O], dO[TPL][O], muR[TPL][RR], xq[TPL][O];
  double acc[TPL];
  R part[TPL];
  LQG_UNROLL for (int k = 0; k < TPL; ++k) {
    const long n = n0 + (long)k * LQG_BLOCK;
    live[k] = n < a.n_trials;
    const long nn = live[k] ? n : a.n_trials - 1;
    xr[k] = a.x.p + sys * a.x.sb + nn * a.x.sn;
    LQG_UNROLL for (int i = 0; i < O; ++i) {
      xprev[k][i] = xr[k][(long)(t0 > 0 ? t0 - 1 : 0) * a.x.st + i * a.x.sd];
      xq[k][i] = xr[k][(long)t0 * a.x.st + i * a.x.sd];
    }
    if (c == 0) {
      LQG_UNROLL for (int i = 0; i < O; ++i) dO[k][i] = R(0);
      LQG_UNROLL for (int p = 0; p < RR; ++p) muR[k][p] = R(0);
    } else {
      const R* sp = a.state + (sys * (a.n_chunks - 1) + (c - 1)) * M * a.n_trials + nn;
      LQG_UNROLL for (int i = 0; i < O; ++i) dO[k][i] = sp[i * a.n_trials];
      LQG_UNROLL for (int p = 0; p < RR; ++p) muR[k][p] = sp[(O + p) * a.n_trials];
    }
    acc[k] = 0.0;
    part[k] = R(0);
  }
  const int tend = last ? t1 + 1 : t1;                                   // the last chunk also scores x_T
  for (int t = t0; t < tend; ++t) {
    const long row = (t + 1 < a.T) ? (long)(t + 1) : (long)a.T;
    const R hlc = op[Ops::H_OFF];
    const bool flush = ((t & (kAccChunk - 1)) == 0) || t + 1 == tend;
    LQG_UNROLL for (int k = 0; k < TPL; ++k) {
      R xt[O];
      LQG_UNROLL for (int i = 0; i < O; ++i) xt[i] = xq[k][i];
      LQG_UNROLL for (int i = 0; i < O; ++i) xq[k][i] = xr[k][row * a.x.st + i * a.x.sd];
      const R zz = tc_step<R, M, ND, FMP, HLC>(op, xt, xprev[k], dO[k], muR[k], t < a.T, t >= lo_from ? lo : nullptr);
      if (t > 0) part[k] += R(0.5) * zz + hlc;
      if (flush) { acc[k] -= (double)part[k]; part[k] = R(0); }
    }
    op += Ops::N;
    if constexpr (HLC) {
      if (lo) lo += hilo_len<M>();
    }
  }
  LQG_UNROLL for (int k = 0; k < TPL; ++k)
    if (live[k]) a.part[(sys * a.n_chunks + c) * a.n_trials + n0 + (long)k * LQG_BLOCK] = acc[k];
}

template <typename R>
__global__ void __launch_bounds__(LQG_BLOCK) k_trial_sum(const double* __restrict__ part, R* ll, long ll_sb, long ll_sn,
                                                         long n_trials, int n_chunks) {
  const long sys = blockIdx.y;
  const long n = (long)blockIdx.x * LQG_BLOCK + threadIdx.x;
  if (n >= n_trials) return;
  const double* p = part + sys * n_chunks * n_trials + n;
  double v = 0.0;
  for (int c = 0; c < n_chunks; ++c) v += p[c * n_trials];
  ll[sys * ll_sb + n * ll_sn] = (R)v;
}

namespace host {

// Number of chunks the per-trial sweep of this problem is cut into (1: the one-pass k_trial).  p->tuning.trial_chunks = -1
// disables, = k forces k chunks (clamped to chunks of >= 4 steps).  By default the sweep is chunked only while the trials
// alone leave the chip latency-bound — at most tuning.trial_chunk_max_waves (2048) waves, two per SIMD — into enough chunks
// to put tuning.trial_chunk_waves (16384) waves in flight, at most sqrt(2 T) chunks (the minimum of the dependent chain
// 2 T / k + k).  Measured, one system of m = 8, T = 500, fp32, per-trial sweep in ms, one pass / chunked
// (scripts/chunk_regime.py): 2^10 trials 0.49 / 0.06, 2^14: 0.50 / 0.11, 2^16 (1024 waves): 0.52 / 0.23, 2^17: 0.55 / 0.44,
// 2^18 (4096 waves): 0.63 / 0.81, 2^19: 0.93 / 1.5 — the two passes cost ~2x the arithmetic, which only pays while the
// one-pass sweep is bound by the latency of its dependent steps and not by instruction issue.
// Joint dimensions without lane kernels (x + b > 24: the delay-augmented models on the row-parallel k_coop_trial_rows, one
// WORKGROUP per few trials): chunked while the trials leave the chip empty — DelayedSubjectiveActor, T = 500, one trial: 1.72 ms
// in one pass (500 dependent steps of 3.4 us).  p->tuning.coop_trial_chunks = -1 disables, = k forces k chunks.
inline int coop_trial_chunks(const lqg_problem* p) {
  if (p->n_trials < 1 || p->T < 64) return 1;
  long nc = p->tuning.coop_trial_chunks;
  if (nc < 0) return 1;
  if (nc == 0) {
    // (every (system, chunk) also pushes m unit vectors through the chunk: pays for a handful of systems only — 64 systems of
    // one trial each: 2.5 ms in one pass, 16 ms chunked)
    // trial_ms one pass / chunked, fp32: 1 trial 1.88 / 0.35, 64: 1.87 / 0.45, 256: 1.88 / 0.87, 1024: 2.36 / 2.80
    if (p->n_sys > 8 || (long)p->n_sys * (long)p->n_trials > 512L) return 1;
    nc = (long)std::sqrt(2.0 * (double)p->T);
  }
  if (nc > p->T / 4) nc = p->T / 4;
  if (nc < 2) return 1;
  const int len = (int)((p->T + nc - 1) / nc);
  return (p->T + len - 1) / len;
}
constexpr int kLaneTrialMaxJoint = 24;    // beyond: no lane per-trial kernels exist (lqg_amd/_hip.py LANE_MAX_JOINT = 20)

inline int trial_chunks(const lqg_problem* p) {
  if (p->dims.x + p->dims.b > kLaneTrialMaxJoint) return coop_trial_chunks(p);
  if (p->n_trials <= 2 || p->T < 16) return 1;
  const int forced = p->tuning.trial_chunks;
  if (forced < 0) return 1;
  const long target = p->tuning.trial_chunk_waves > 0 ? p->tuning.trial_chunk_waves : 16384L;
  const long max_waves = p->tuning.trial_chunk_max_waves > 0 ? p->tuning.trial_chunk_max_waves : 2048L;
  long nc;
  if (forced > 0) {
    nc = forced;
  } else {
    const long waves = (long)p->n_sys * ((p->n_trials + LQG_BLOCK - 1) / LQG_BLOCK);
    if (waves > max_waves) return 1;
    nc = (target + waves - 1) / waves;
    const long cap = (long)std::sqrt(2.0 * (double)p->T);
    if (nc > cap) nc = cap;
  }
  if (nc > p->T / 4) nc = p->T / 4;
  if (nc < 2) return 1;
  const int len = (int)((p->T + nc - 1) / nc);
  return (p->T + len - 1) / len;                        // (every chunk non-empty)
}
inline int trial_chunk_len(const lqg_problem* p, int n_chunks) { return (p->T + n_chunks - 1) / n_chunks; }

// bytes per system of the row lists of the operator's mean-update block (large joint dimensions, k_coop_trial_rows): count per
// row | columns per row | one flag byte (lists in use), lqg_coop.hpp k_coop_trial_lists
inline size_t trial_row_lists_bytes(const lqg_dims& d) {
  const size_t m = (size_t)(d.x + d.b);
  return m > (size_t)kLaneTrialMaxJoint ? (m + m * m + 1 + 15) / 16 * 16 : 0;
}
struct TrialChunkScratch {
  size_t state_off, phi_off, part_off, lists_off, total;
};
inline TrialChunkScratch trial_chunk_scratch(const lqg_problem* p) {
  TrialChunkScratch s{};
  auto al = [](size_t v) { return (v + 255) / 256 * 256; };
  const int nc = trial_chunks(p);
  if (nc > 1) {
    const size_t esz = p->dtype == LQG_F64 ? 8 : 4;
    const size_t m = (size_t)(p->dims.x + p->dims.b), B = (size_t)p->n_sys, N = (size_t)p->n_trials;
    s.state_off = 0;
    s.phi_off = al(B * (nc - 1) * m * N * esz);
    s.part_off = s.phi_off + al(B * (nc - 1) * m * m * esz);
    s.total = s.part_off + al(B * nc * N * sizeof(double));
  }
  s.lists_off = s.total;
  s.total += al((size_t)p->n_sys * trial_row_lists_bytes(p->dims));
  return s;
}

// scratch: trial_chunk_scratch(p).total bytes (by convention it follows the operator stream in the workspace)
template <typename R, int M, int ND, class FMP, bool HLC = false>
hipError_t launch_trial_chunked(const lqg_problem* p, const void* ops, lqg_traj x, void* ll, long ll_sb, long ll_sn,
                                void* scratch, hipStream_t st, const float* ops_lo = nullptr, const int* hl = nullptr) {
  const int nc = trial_chunks(p);
  const TrialChunkScratch sc = trial_chunk_scratch(p);
  char* base = static_cast<char*>(scratch);
  lqg::TrialChunkArgs<R> k{lqg::DTraj<R>{static_cast<const R*>(x.ptr), (long)x.sb, (long)x.sn, (long)x.st, (long)x.sd},
                           (long)p->n_trials, p->T, nc, trial_chunk_len(p, nc),
                           reinterpret_cast<R*>(base + sc.state_off), reinterpret_cast<R*>(base + sc.phi_off),
                           reinterpret_cast<double*>(base + sc.part_off), ops_lo, hl};
  const unsigned tb = (unsigned)((p->n_trials + LQG_BLOCK - 1) / LQG_BLOCK), B = (unsigned)p->n_sys;
  const dim3 block(LQG_BLOCK);
  const R* o = static_cast<const R*>(ops);
  // two trials per lane once (trial, chunk) pairs alone over-fill the chip (> 8 waves per SIMD at one per lane)
  const int tpl_mode = p->tuning.trial_chunk_tpl;
  const bool two = tpl_mode == 2 || (tpl_mode == 0 && (long)tb * B * nc > 8192L);
  const unsigned tb2 = (unsigned)((p->n_trials + 2 * LQG_BLOCK - 1) / (2 * LQG_BLOCK));
  if (two) hipLaunchKernelGGL((lqg::k_trial_zs<R, M, ND, FMP, 2, HLC>), dim3(tb2 + 1, B, nc - 1), block, 0, st, o, k);
  else hipLaunchKernelGGL((lqg::k_trial_zs<R, M, ND, FMP, 1, HLC>), dim3(tb + 1, B, nc - 1), block, 0, st, o, k);
  if (nc > 2)
    hipLaunchKernelGGL((lqg::k_trial_fix<R, M>), dim3(tb, B), block, 0, st, static_cast<const R*>(k.phi), k.state,
                       (long)p->n_trials, nc - 1);
  if (two) hipLaunchKernelGGL((lqg::k_trial_ll<R, M, ND, FMP, 2, HLC>), dim3(tb2, B, nc), block, 0, st, o, k);
  else hipLaunchKernelGGL((lqg::k_trial_ll<R, M, ND, FMP, 1, HLC>), dim3(tb, B, nc), block, 0, st, o, k);
  hipLaunchKernelGGL((lqg::k_trial_sum<R>), dim3(tb, B), block, 0, st, static_cast<const double*>(k.part),
                     static_cast<R*>(ll), ll_sb, ll_sn, (long)p->n_trials, nc);
  return hipGetLastError();
}

}  // namespace host
}  // namespace lqg
