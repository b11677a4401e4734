// lqg_kernels_sp.hpp — STRUCTURE-SPECIALISED instantiations of the hot path (time-invariant specs, one trial per
// system, log-likelihood only): the same mathematics as k_riccati / k_forward<TI, FUSED> of lqg_kernels.hpp, written
// on the sparsity-typed matrices of lqg_sparse.hpp.  `PAT` carries, as compile-time masks, the structural zeros of
// every per-system constant of a model family (which entries of A, B, F, V V^T, W W^T, Q, R and of the hoisted
// products Fa Aa, Fd Ad, Fd Bd - Fa Ba, Fd Vd Vd^T, ... can ever be non-zero); lqg_amd/specialize.py derives the
// masks from the actual spec tensors (union over candidates), generates one translation unit per pattern and
// caches the compiled library.  With the all-true pattern this is the dense kernel; with the pattern of e.g.
// SubjectiveActor(dim=2) (A = I + 2 couplings, F = [I 0], diagonal noise, Fd Bd - Fa Ba == 0) a step shrinks from
// ~3700 to ~1500 instructions and the register working set from ~510 to ~270 per lane.
//
// Reference parity: lqg/control/lqr.py:16-42, lqg/belief/kf.py:6-21, lqg/system.py:167-248 (as lqg_kernels.hpp).
#pragma once
#ifndef LQG_SP_PREFETCH
#define LQG_SP_PREFETCH 1
#endif
#include <type_traits>
#include <utility>

#include "lqg_kernels.hpp"
#include "lqg_sparse.hpp"

// lambdas of the sweeps are called from several sites (first / whole / last chunk): without the attribute hipcc outlines them
// and every array they capture by reference goes to scratch memory (measured: 526 flat_load + 296 flat_store per chunk)
#define LQG_LAMBDA_INLINE __attribute__((always_inline))

namespace lqg {

#ifndef LQG_SP_RIC_WAVES
#define LQG_SP_RIC_WAVES 2
#endif
// waves per SIMD the forward kernel is allocated for: 2 in fp32 (256 registers per lane, measured 13 % faster than 1); in fp64
// 2 for the smallest joint dimensions (m = x + b <= 5: the 1-D tracking components — 255 registers with 4 spilled at the
// headline shape, round 4: 134.6 -> 160.5 M solves/s together with the checkpointed gains below), else 1 (the working set of
// m >= 8 needs the full 512-register file; at 2 it spills and runs 5x slower)
#ifndef LQG_SP_FWD_WAVES_F32
#define LQG_SP_FWD_WAVES_F32 2
#endif
#ifndef LQG_SP_FWD_WAVES_F64_SMALL
#define LQG_SP_FWD_WAVES_F64_SMALL 2
#endif
#ifndef LQG_SP_FWD_WAVES_F64
#define LQG_SP_FWD_WAVES_F64 1
#endif
template <typename R, int M>
constexpr int sp_fwd_waves() {
  return sizeof(R) == 4 ? LQG_SP_FWD_WAVES_F32 : (M <= 5 ? LQG_SP_FWD_WAVES_F64_SMALL : LQG_SP_FWD_WAVES_F64);
}

// ---------------------------------------------------------------- Riccati backward, TI, no affine terms
// LQG_SP_CHUNK = CK > 0: CHECKPOINTED gains.  The backward sweep keeps only the packed cost-to-go S every CK steps
// (NB (NB + 1) / 2 reals per CK steps instead of NU NB reals per step); the forward sweep re-runs the CK backward steps of
// each chunk in registers (the same riccati_step_sp, hence bitwise the same L_t) right before it consumes them.  Trades
// HBM traffic of the gain stream (written once, read once) for NU-row Riccati work on the forward kernel's VALU.
// Measured (MI355X, headline shape x=2 b=3 u=1, 2^20 systems, profiles/r02_a_chunk_*.json): fp32 CK = 0 / 4 / 8 / 16 ->
// 241 / 246 / 253 / 230 M solves/s (the Riccati kernel stops being HBM-write-bound: 1.02 -> 0.69 ms; the forward kernel
// pays 3.28 -> 3.41 ms for the recompute and drops from 4 to 3 waves per SIMD; at 16 the 48 gain registers cost more);
// fp64 CK = 0 / 8 -> 113 / 111 M (fp64 VALU runs at half rate: the recompute costs more than the traffic it saves).
// Round 4: with whole chunks run without per-step guards (k_forward_sp) the chunked fp64 sweep issues 249 instead of 305
// instructions per step (the per-step loop of CK = 0 carries 26 v_mov_b64 and is not unrolled) and moves half the bytes:
// CK = 0 / 4 / 8 (1 wave) / 8 (2 waves per SIMD) -> 134.6 / 151.2 / 142.0 / 160.5 M solves/s.
// Hence: 8 in both precisions, and only for gains of at most 4 reals per step (u b <= 4: every 1-D tracking model).
// LQG_SP_CHUNK overrides both (A/B builds: scripts/exp_chunk.sh).
#ifdef LQG_SP_CHUNK
#define LQG_SP_CHUNK_F32 LQG_SP_CHUNK
#define LQG_SP_CHUNK_F64 LQG_SP_CHUNK
#endif
#ifndef LQG_SP_CHUNK_F32
#define LQG_SP_CHUNK_F32 8
#endif
#ifndef LQG_SP_CHUNK_F64
#define LQG_SP_CHUNK_F64 8
#endif
#ifndef LQG_SP_CHUNK_MAX_GAIN
#define LQG_SP_CHUNK_MAX_GAIN 4
#endif
template <typename R, int NB, int NU>
constexpr int sp_chunk() {
  return (NU * NB <= LQG_SP_CHUNK_MAX_GAIN) ? (sizeof(R) == 4 ? LQG_SP_CHUNK_F32 : LQG_SP_CHUNK_F64) : 0;
}

// one backward step: S <- Q + A'SA + L'HL + L'G + G'L with L = -Ht^-1 G (lqr.py:22-34), L returned
template <typename R, int NB, int NU, typename MA, typename MB, typename MQ, typename MR>
LQG_DEV void riccati_step_sp(R (&S)[NB * NB], const MA& A, const MB& Bm, const MQ& Q, const MR& Rm, const R eps,
                             R (&L)[NU * NB], R* Ht_out = nullptr, const R* Pc = nullptr) {
  const auto Sm = from_dense<R, NB, NB>(S);
  const auto SA = mul(Sm, A);
  const auto SB = mul(Sm, Bm);
  R H[NU * NU], G[NU * NB];
  to_dense(mul_tn_sym_add(Bm, SB, Rm), H);                       // H = R + B^T S B     lqr.py:22
  to_dense(mul_tn(Bm, SA), G);                                   // G = B^T S A         lqr.py:23
  if (Pc) { LQG_UNROLL for (int i = 0; i < NU * NB; ++i) G[i] += Pc[i]; }   // G = P + B^T S A (cross cost; dense, time-varying sweeps only)
  R ev0 = min_eig_sym<R, NU>(H);
  R shift = eps - ev0;
  shift = (shift > R(0)) ? shift : R(0);
  R Ht[NU * NU];
  LQG_UNROLL for (int i = 0; i < NU * NU; ++i) Ht[i] = H[i];
  LQG_UNROLL for (int i = 0; i < NU; ++i) Ht[i * NU + i] += shift;  // lqr.py:27-28
  if (Ht_out) { LQG_UNROLL for (int i = 0; i < NU * NU; ++i) Ht_out[i] = Ht[i]; }
  R Lc[NU * NU], dinv[NU], Li[NU * NU], Hi[NU * NU];
  chol_lower<R, NU>(Ht, Lc, dinv);
  tri_inverse_lower<R, NU>(Lc, dinv, Li);
  spd_inverse_from_tri<R, NU>(Li, Hi);
  R W1[NU * NB];
  LQG_UNROLL for (int i = 0; i < NU; ++i)
    LQG_UNROLL for (int j = 0; j < NB; ++j) {
      R acc = R(0);
      LQG_UNROLL for (int k = 0; k < NU; ++k) acc -= Hi[i * NU + k] * G[k * NB + j];
      L[i * NB + j] = acc;                                       // L = -Ht^-1 G        lqr.py:30
    }
  LQG_UNROLL for (int i = 0; i < NU; ++i)
    LQG_UNROLL for (int j = 0; j < NB; ++j) {
      R acc = G[i * NB + j];
      LQG_UNROLL for (int k = 0; k < NU; ++k) acc += H[i * NU + k] * L[k * NB + j];
      W1[i * NB + j] = acc;                                      // H L + G (unregularised H)
    }
  R Sn[NB * NB];
  to_dense(mul_tn_sym_add(A, SA, Q), Sn);                        // Q + A^T S A
  LQG_UNROLL for (int i = 0; i < NB; ++i)
    LQG_UNROLL for (int j = i; j < NB; ++j) {
      R acc = Sn[i * NB + j];
      LQG_UNROLL for (int k = 0; k < NU; ++k) acc += L[k * NB + i] * W1[k * NB + j] + G[k * NB + i] * L[k * NB + j];
      S[i * NB + j] = acc;                                       // lqr.py:33
      S[j * NB + i] = acc;
    }
}

template <typename R, int NB, int NU, typename PAT, int CK>
__global__ void __launch_bounds__(LQG_BLOCK, LQG_SP_RIC_WAVES) k_riccati_sp(const RiccatiArgs<R> a) {
  const long s = blockIdx.x * (long)LQG_BLOCK + threadIdx.x;
  if (s >= a.n_sys) return;
  constexpr int NS = NB * (NB + 1) / 2;
  R S[NB * NB];
  load_sym<R, NB>(a.Qf.p + s * a.Qf.sb, a.Qf.sr, a.Qf.sc, S);
  const auto A = load_masked<R, NB, NB, PAT::Aa>(a.A.p + s * a.A.sb, a.A.sr, a.A.sc);
  const auto Bm = load_masked<R, NB, NU, PAT::Ba>(a.B.p + s * a.B.sb, a.B.sr, a.B.sc);
  const auto Q = load_sym_masked<R, NB, PAT::Q>(a.Q.p + s * a.Q.sb, a.Q.sr, a.Q.sc);
  const auto Rm = load_sym_masked<R, NU, PAT::Rr>(a.Rm.p + s * a.Rm.sb, a.Rm.sr, a.Rm.sc);

  for (int t = a.T - 1; t >= 0; --t) {
    if constexpr (CK > 0) {
      if ((t + 1) % CK == 0 || t == a.T - 1) {       // t is the last step of its chunk: keep S_{t+1}
        R* dst = a.Ls + (long)(t / CK) * NS * a.ldb + s;
        int e = 0;
        LQG_UNROLL for (int i = 0; i < NB; ++i)
          LQG_UNROLL for (int j = i; j < NB; ++j) dst[(e++) * a.ldb] = S[i * NB + j];
      }
    }
    R L[NU * NB];
    riccati_step_sp<R, NB, NU>(S, A, Bm, Q, Rm, a.eps, L);
    if constexpr (CK == 0) {
      R* dst = a.Ls + (long)t * (NU * NB) * a.ldb + s;
      LQG_UNROLL for (int e = 0; e < NU * NB; ++e) dst[e * a.ldb] = L[e];
    }
  }
}

// ---------------------------------------------------------------- forward sweep, TI, log-likelihood only
// Structural mask of the loop-carried Kalman covariance P_t: the least fixed point of the mask recursion of kf.py:10-14
// started from mask(V V^T) (the default Sigma0).  For the tracking models the cursor state is filtered independently of
// (target, velocity): P keeps 2 structural zeros, F P F' + W W' is DIAGONAL (inverted without a Cholesky), K has 3 of 6
// entries, and the sparsity propagates into the joint system and G G^T through the mask algebra.  DENSE (an explicit
// Sigma0 is supplied at run time): the mask is full.
template <typename PAT, int NB, int NY, bool DENSE>
constexpr Mask<NB, NB> kalman_state_mask() {
  if (DENSE) return mask_full<NB, NB>();
  Mask<NB, NB> P = mask_or(PAT::VVa, mask_t(PAT::VVa));
  for (int it = 0; it <= NB * NB; ++it) {
    const auto Pp = mask_or(mask_mul(mask_mul(PAT::Aa, P), mask_t(PAT::Aa)), PAT::VVa);
    const auto FP = mask_mul(PAT::Fa, Pp);
    const auto G = mask_or(mask_mul(FP, mask_t(PAT::Fa)), PAT::WWa);
    const auto Gi = mask_is_diag(G) ? G : mask_full<NY, NY>();
    const auto K = mask_mul(mask_t(FP), Gi);
    auto Pn = mask_or(Pp, mask_mul(K, FP));
    Pn = mask_or(mask_or(Pn, mask_t(Pn)), P);
    if (mask_eq(Pn, P)) break;
    P = Pn;
  }
  return P;
}

// Structural masks of the joint dynamics Fj (system.py:167-187) and of the per-step trial operator  Fj - I  that
// k_forward_sp<NTR = 0> writes to the operator stream: the mask algebra of the joint system on the pattern's masks, the
// loop-carried Kalman mask included.  k_forward_sp static_asserts that they equal the masks of the matrices it really
// builds.  DEV (the operator): the diagonal blocks start from A - I — Ad - I carries the pattern's numerically observed
// mask PAT::AdmI (structurally zero where the dynamics' A has a unit diagonal), Aa - I keeps its diagonal.
template <int N>
constexpr Mask<N, N> mask_sym(const Mask<N, N>& a) { return mask_or(a, mask_t(a)); }

template <typename PAT, int NX, int NB, int NU, int NY, bool DENSE_P, bool DEV>
constexpr Mask<NX + NB, NX + NB> joint_mask_impl() {
  const auto P = kalman_state_mask<PAT, NB, NY, DENSE_P>();
  const auto AP = mask_mul(PAT::Aa, P);
  const auto Pp = mask_or(mask_sym(mask_mul(AP, mask_t(PAT::Aa))), PAT::VVa);
  const auto FP = mask_mul(PAT::Fa, Pp);
  const auto G = mask_or(mask_sym(mask_mul(FP, mask_t(PAT::Fa))), PAT::WWa);
  const auto Gi = mask_is_diag(G) ? G : mask_full<NY, NY>();
  const auto K = mask_mul(mask_t(FP), Gi);
  const auto FAa = mask_and(mask_mul(PAT::Fa, PAT::Aa), PAT::FAa);
  const auto FAd = mask_and(mask_mul(PAT::Fd, PAT::Ad), PAT::FAd);
  const auto DB = mask_and(mask_or(mask_mul(PAT::Fd, PAT::Bd), mask_mul(PAT::Fa, PAT::Ba)), PAT::DB);
  const auto BK = mask_or(PAT::Ba, mask_mul(K, DB));
  const auto Lf = mask_full<NU, NB>();
  const auto A11 = DEV ? mask_and(mask_or(PAT::Ad, mask_eye<NX>()), PAT::AdmI) : PAT::Ad;
  const auto A22 = DEV ? mask_or(PAT::Aa, mask_eye<NB>()) : PAT::Aa;
  return mask_block(A11, mask_mul(PAT::Bd, Lf), mask_mul(K, FAd),
                    mask_or(mask_or(A22, mask_mul(K, FAa)), mask_mul(BK, Lf)));
}
template <typename PAT, int NX, int NB, int NU, int NY, bool DENSE_P>
constexpr Mask<NX + NB, NX + NB> joint_dynamics_mask() {
  return joint_mask_impl<PAT, NX, NB, NU, NY, DENSE_P, false>();
}
template <typename PAT, int NX, int NB, int NU, int NY, int ND, bool DENSE_P>
constexpr Mask<NX + NB, NX + NB> trial_operator_mask() {
  return joint_mask_impl<PAT, NX, NB, NU, NY, DENSE_P, true>();
}

// MIXED mode (k_forward_sp<double, ..., OT = float>): an operator rounded once to fp32 carries eps32 |F - I| |state| of
// SYSTEMATIC error per step into every trial — the same rounding every step of a stationary filter.  Measured (NumPy emulation
// of the fp32 per-trial sweep, PointMassBoundedActor over the bench's candidate ranges, T = 1067, error / max(|ll|, T d)):
// rounded operators 1.19e-6 worst, hi + lo operators 1.2e-7, exact operators 1.07e-7; the 1-D tracking models (|F - I| <= 0.6
// against 5 .. 74 for the point mass) sit at 1e-7 either way.  So from the FIRST step at which a system's block reaches
// LQG_HILO_MIN (looked at every 8th step) the builder also writes the residual block of every step (ForwardArgs::hl = that step
// + 1; 0 = never: such systems cost two integer instructions per entry every 8th step, no stores), and the per-trial sweeps
// apply hi + lo from that step on.
#ifndef LQG_HILO_MIN
#define LQG_HILO_MIN 2.0
#endif
// largest |entry| of an fp64 block, as the largest high dword of |v| (monotone in |v|; full-rate integer operations — the builder
// walks T dependent steps on a nearly empty chip, where every instruction costs its whole latency: an fp64 abs-max per entry
// and step cost 0.25 ms of config 3's 0.6 ms sweep)
template <int I, int M, int N, Mask<M, N> MK>
LQG_DEV void block_absmax_hi(const Mat<double, M, N, MK>& a, unsigned (&mx)[2]) {
  if constexpr (I < M * N) {
    if constexpr (MK.b[I]) {
      const unsigned h = (unsigned)__double2hiint(a.v[I]) & 0x7fffffffu;
      mx[I & 1] = h > mx[I & 1] ? h : mx[I & 1];
    }
    block_absmax_hi<I + 1>(a, mx);
  }
}
template <int I, typename OT, typename R, int M, int N, Mask<M, N> MK>
LQG_DEV void store_residual(const Mat<R, M, N, MK>& a, OT* __restrict__ p) {
  if constexpr (I < M * N) {
    if constexpr (MK.b[I]) p[I] = (OT)(a.v[I] - (R)(OT)a.v[I]);
    store_residual<I + 1, OT>(a, p);
  }
}

// OT: element type of the operator stream (NTR == 0), see k_forward.
// X4 (round 5, fp32, NTR x d = 4): the lane's data row is ONE 16-byte vector — trajectories laid [T+1][system][trial][component]
// (plan._trial_stack(rows=True)): one global_load_dwordx4 per step instead of four dword loads from four rows.
template <typename R, int NX, int NB, int NU, int NY, int ND, typename PAT, int NTR, bool DENSE_P, int CK, typename OT = R, bool X4 = false>
__global__ void __launch_bounds__(LQG_BLOCK, (sp_fwd_waves<R, NX + NB>()))
    k_forward_sp(const ForwardArgs<R> a, const long ll_sn, const RiccatiArgs<R> rc) {
  constexpr bool FUSED = NTR > 0;
  constexpr int NT = FUSED ? NTR : 1;
  constexpr int M = NX + NB, O = ND, RR = M - ND;
  using Ops = TrialOps<M, ND>;
  const long s = blockIdx.x * (long)LQG_BLOCK + threadIdx.x;
  if (s >= a.n_sys) return;

  // ---- per-system constants with their structural masks
  const auto Aa = load_masked<R, NB, NB, PAT::Aa>(a.aA.p + s * a.aA.sb, a.aA.sr, a.aA.sc);
  const auto Ba = load_masked<R, NB, NU, PAT::Ba>(a.aB.p + s * a.aB.sb, a.aB.sr, a.aB.sc);
  const auto Fa = load_masked<R, NY, NB, PAT::Fa>(a.aF.p + s * a.aF.sb, a.aF.sr, a.aF.sc);
  const auto VVa = load_gram_masked<R, NB, PAT::VVa>(a.aV.p + s * a.aV.sb, a.aV.sr, a.aV.sc, a.nva);
  const auto WWa = load_gram_masked<R, NY, PAT::WWa>(a.aW.p + s * a.aW.sb, a.aW.sr, a.aW.sc, a.nwa);
  const auto Ad = load_masked<R, NX, NX, PAT::Ad>(a.dA.p + s * a.dA.sb, a.dA.sr, a.dA.sc);
  const auto Bd = load_masked<R, NX, NU, PAT::Bd>(a.dB.p + s * a.dB.sb, a.dB.sr, a.dB.sc);
  const auto N1 = load_gram_masked<R, NX, PAT::N1>(a.dV.p + s * a.dV.sb, a.dV.sr, a.dV.sc, a.nvd);
  // hoisted products; restrict_to applies the numerically observed masks (e.g. Fd Bd - Fa Ba == 0)
  const auto FAa = restrict_to<PAT::FAa>(mul(Fa, Aa));
  const auto [FAd, DB, N2, N3] = [&]() {
    const auto Fd = load_masked<R, NY, NX, PAT::Fd>(a.dF.p + s * a.dF.sb, a.dF.sr, a.dF.sc);
    const auto WWd = load_gram_masked<R, NY, PAT::WWd>(a.dW.p + s * a.dW.sb, a.dW.sr, a.dW.sc, a.nwd);
    const auto fad = restrict_to<PAT::FAd>(mul(Fd, Ad));
    const auto db = restrict_to<PAT::DB>(sub(mul(Fd, Bd), mul(Fa, Ba)));
    const auto n2 = restrict_to<PAT::N2>(mul(Fd, N1));
    const auto n3 = restrict_to<PAT::N3>(mul_nt_sym_add(mul(Fd, N1), Fd, WWd));
    struct Out { decltype(fad) a; decltype(db) b; decltype(n2) c; decltype(n3) d; };
    return Out{fad, db, n2, n3};
  }();

  constexpr auto PMASK = kalman_state_mask<PAT, NB, NY, DENSE_P>();
  Mat<R, NB, NB, PMASK> Pm;                     // loop-carried Kalman covariance, structural zeros in the type
  {
    R P0[NB * NB];
    if (DENSE_P && a.Sigma0.p) load_sym<R, NB>(a.Sigma0.p + s * a.Sigma0.sb, a.Sigma0.sr, a.Sigma0.sc, P0);
    else load_gram<R, NB>(a.aV.p + s * a.aV.sb, a.aV.sr, a.aV.sc, a.nva, P0);
    LQG_UNROLL for (int i = 0; i < NB * NB; ++i)
      if (PMASK.b[i]) Pm.v[i] = P0[i];
  }

  R Sg[M * M], xprev[NT][O], dO[NT][O], muR[NT][RR];   // observed mean = xprev + dO (deviation form, see k_forward)
  double acc[NT];
  const R* xp = nullptr;
  LQG_UNROLL for (int k = 0; k < NT; ++k) acc[k] = 0.0;
  static_assert(!X4 || (FUSED && sizeof(R) == 4 && NT * O == 4), "X4: four floats per lane and row");
  if (FUSED) {
    xp = a.x.p + s * a.x.sb;
    if constexpr (X4) {
      const float4 v = *reinterpret_cast<const float4*>(xp);
      const float vv[4] = {v.x, v.y, v.z, v.w};
      LQG_UNROLL for (int k = 0; k < NT; ++k)
        LQG_UNROLL for (int i = 0; i < O; ++i) xprev[k][i] = vv[k * O + i];
    }
    LQG_UNROLL for (int k = 0; k < NT; ++k) {
      LQG_UNROLL for (int i = 0; i < O; ++i) { if constexpr (!X4) xprev[k][i] = xp[k * a.x.sn + i * a.x.sd]; dO[k][i] = R(0); }
      LQG_UNROLL for (int i = 0; i < RR; ++i) muR[k][i] = R(0);
    }
  }
  const R kLogNorm = R(0.5 * ND * 1.8378770664093453);

  R Li[O * O], U2[RR * O], hl = R(0), pd = R(1);
  unsigned pois = 0u;                             // largest pos_finite_key of the pivot products (lqg_small.hpp)
  [[maybe_unused]] int hl_first = -1;             // MIXED: first step whose |Fj - I| block reaches LQG_HILO_MIN (ForwardArgs::hl)
  // FUSED scoring (round 4): log N(x_t; mu, S) = -1/2 |w|^2 + log(prod of the Cholesky pivots' reciprocal roots) - (d/2) log 2 pi.
  // Per step only  part[n] += 1/2 |w|^2  (one fma per trial) and the log-determinant term run: fp32 takes ONE v_log_f32
  // (log2, 1 ulp; the ln 2 factor is applied in fp64 at the flush), fp64 multiplies the pivot products of the block together
  // and takes one log per flush (software log: ~40 instructions).  Every 8 steps the partial sums are flushed into the fp64
  // accumulators; the constant is added once at the end.  Before: a full logf (range checks, extended-precision ln 2 product),
  // three adds, a conversion and an fp64 add per trial per step — 23 of the step's 266 instructions.
  R part[NT], ldet = (sizeof(R) == 4) ? R(0) : R(1);
  LQG_UNROLL for (int k = 0; k < NT; ++k) part[k] = R(0);
  auto flush = [&]() LQG_LAMBDA_INLINE {
    double ld;
    if constexpr (sizeof(R) == 4) { ld = 0.6931471805599453 * (double)ldet; ldet = R(0); }
    else {
      // (the PRODUCT of the block's pivot products can leave (0, inf) although every factor is inside — |log10| beyond ~38 per
      // step: it is keyed like the factors, so that such a block poisons the result instead of returning +-inf silently)
      const unsigned key = pos_finite_key(ldet);
      pois = key > pois ? key : pois;
      ld = (double)log_<R>(ldet);
      ldet = R(1);
    }
    LQG_UNROLL for (int k = 0; k < NT; ++k) { acc[k] += ld - (double)part[k]; part[k] = R(0); }
  };
  auto condition = [&]() LQG_LAMBDA_INLINE {
    R Soo[O * O], Lc[O * O], dinv[O];
    LQG_UNROLL for (int i = 0; i < O; ++i)
      LQG_UNROLL for (int j = 0; j < O; ++j) Soo[i * O + j] = Sg[i * M + j];
    chol_lower<R, O>(Soo, Lc, dinv);
    tri_inverse_lower<R, O>(Lc, dinv, Li);
    pd = dinv[0];
    LQG_UNROLL for (int i = 1; i < O; ++i) pd *= dinv[i];
    {
      const unsigned key = pos_finite_key(pd);
      pois = key > pois ? key : pois;
    }
    if constexpr (!FUSED) hl = -log_<R>(pd);
    LQG_UNROLL for (int p = 0; p < RR; ++p)
      LQG_UNROLL for (int j = 0; j < O; ++j) {
        R v = R(0);
        LQG_UNROLL for (int k = 0; k <= j; ++k) v += Sg[(O + p) * M + k] * Li[j * O + k];
        U2[p * O + j] = v;
      }
  };
  auto score_logdet = [&]() LQG_LAMBDA_INLINE {                      // the step's log-determinant term (shared by the NT trials)
    if constexpr (sizeof(R) == 4) ldet += __builtin_amdgcn_logf(pd);       // v_log_f32: log2
    else ldet *= pd;
  };
  R w[O], xt[O];
#if LQG_SP_PREFETCH
  // software pipeline of the two per-step HBM streams: the rows of step t+1 are requested at the top of step t
  R xnx[NT][O], Lnx[NU * NB];
  // (streamed gains are requested TWO steps ahead: the counter that orders a wave's memory operations is in-order, so a wait for
  // a gain row requested in the same iteration as the operator stores also waits for those stores — on the MIXED builder, one
  // wave per SIMD, that exposed ~200 ns per step once the compiler moved the row's first use next to its request)
  [[maybe_unused]] R Lnx2[CK == 0 ? NU * NB : 1];
  if (FUSED) {
    LQG_UNROLL for (int k = 0; k < NT; ++k)
      LQG_UNROLL for (int i = 0; i < O; ++i) xnx[k][i] = xprev[k][i];
  }
  if constexpr (CK == 0) {
    LQG_UNROLL for (int e = 0; e < NU * NB; ++e) Lnx[e] = a.Ls[e * a.ldb + s];
    const R* src1 = a.Ls + (long)(a.T > 1 ? 1 : 0) * (NU * NB) * a.ldb + s;
    LQG_UNROLL for (int e = 0; e < NU * NB; ++e) Lnx2[e] = src1[e * a.ldb];
  }
#endif
  // ---- checkpointed gains: the chunk's L_t are recomputed backward from the kept S into registers
  constexpr int CKN = CK > 0 ? CK : 1;
  constexpr int NS = NB * (NB + 1) / 2;
  R Lbuf[CKN][NU * NB], Snx[NS];
  [[maybe_unused]] const auto rQ = [&]() {
    if constexpr (CK > 0) return load_sym_masked<R, NB, PAT::Q>(rc.Q.p + s * rc.Q.sb, rc.Q.sr, rc.Q.sc);
    else return 0;
  }();
  [[maybe_unused]] const auto rR = [&]() {
    if constexpr (CK > 0) return load_sym_masked<R, NU, PAT::Rr>(rc.Rm.p + s * rc.Rm.sb, rc.Rm.sr, rc.Rm.sc);
    else return 0;
  }();
  auto request_ckpt = [&](int c) LQG_LAMBDA_INLINE {                       // issue the loads of checkpoint c (S at the END of chunk c)
    const R* src = rc.Ls + (long)c * NS * rc.ldb + s;
    LQG_UNROLL for (int e = 0; e < NS; ++e) Snx[e] = src[e * rc.ldb];
  };
  auto refill = [&]<bool WHOLE = false>(int t0) LQG_LAMBDA_INLINE {        // gains of steps t0 .. t0 + CK - 1 from the requested checkpoint
    if constexpr (CK > 0) {
      R Sr[NB * NB];
      {
        int e = 0;
        LQG_UNROLL for (int i = 0; i < NB; ++i)
          LQG_UNROLL for (int j = i; j < NB; ++j) { Sr[i * NB + j] = Snx[e]; Sr[j * NB + i] = Snx[e]; ++e; }
      }
      if (t0 + CK < a.T) request_ckpt(t0 / CK + 1);      // next chunk's checkpoint: a whole chunk of work hides it
      LQG_UNROLL for (int j = CK - 1; j >= 0; --j)
        if (WHOLE || t0 + j < a.T) riccati_step_sp<R, NB, NU>(Sr, Aa, Ba, rQ, rR, rc.eps, Lbuf[j]);
    }
  };
  auto innovate = [&](int n, int row, bool score) LQG_LAMBDA_INLINE {        // trial n, data row `row`
#if LQG_SP_PREFETCH
    (void)row;
    LQG_UNROLL for (int i = 0; i < O; ++i) xt[i] = xnx[n][i];
#else
    const R* xr = xp + n * a.x.sn + (long)row * a.x.st;
    LQG_UNROLL for (int i = 0; i < O; ++i) xt[i] = xr[i * a.x.sd];
#endif
    R zz = R(0);
    LQG_UNROLL for (int i = 0; i < O; ++i) {
      R v = R(0);
      LQG_UNROLL for (int j = 0; j <= i; ++j) v += Li[i * O + j] * ((xt[j] - xprev[n][j]) - dO[n][j]);
      w[i] = v;
      zz += v * v;
    }
    if (score) part[n] += R(0.5) * zz;
  };

  // The first step is peeled (templated lambda): it alone initialises Sigma := G_0 G_0^T and skips the score of x_0;
  // inside one loop the compiler turns those two `t == 0` tests into ~30 v_cndmask per step.
  auto step = [&]<bool FIRST, int J>(int t) LQG_LAMBDA_INLINE {
    // ---- Kalman step                                                   kf.py:10-14
    const auto AP = mul(Aa, Pm);
    const auto Pp = mul_nt_sym_add(AP, Aa, VVa);
    const auto FP = mul(Fa, Pp);
    const auto Gi = spd_inverse_masked(mul_nt_sym_add(FP, Fa, WWa));     // (F Pp F' + W W')^-1
    const auto K = mul_tn(FP, Gi);                                       // K = (F Pp)^T Gk^-1
    assign_state(Pm, sym_sub_mul(Pp, K, FP));                            // P = Pp - K F Pp
    // ---- control gain L_t
    Mat<R, NU, NB> L;
    if constexpr (CK > 0) {
      LQG_UNROLL for (int e = 0; e < NU * NB; ++e) L.v[e] = Lbuf[J][e];
    } else {
#if LQG_SP_PREFETCH
      LQG_UNROLL for (int e = 0; e < NU * NB; ++e) L.v[e] = Lnx[e];
#else
      const R* src = a.Ls + (long)t * (NU * NB) * a.ldb + s;
      LQG_UNROLL for (int e = 0; e < NU * NB; ++e) L.v[e] = src[e * a.ldb];
#endif
    }
    // ---- joint dynamics                                                system.py:167-187
    const auto BK = add(Ba, mul(K, DB));
    const auto Fj = block2x2(Ad, mul(Bd, L), mul(K, FAd), add(sub(Aa, mul(K, FAa)), mul(BK, L)));
    static_assert(mask_eq(decltype(Fj)::mask, joint_dynamics_mask<PAT, NX, NB, NU, NY, DENSE_P>()),
                  "joint_dynamics_mask() must mirror the mask algebra of the joint system built here");
    // ---- joint noise covariance                                        system.py:190-207
    const auto KN2 = mul(K, N2);
    const auto GG = block2x2(N1, transpose(KN2), KN2, mul_nt_sym_add(mul(K, N3), K, Mat<R, NB, NB, mask_none<NB, NB>()>{}));
    if constexpr (FIRST) to_dense(GG, Sg);                               // Sigma0 := G[0] G[0]^T  system.py:212
    // ---- condition on x_t, score it, propagate the mean                system.py:219-221, 244-248
    condition();
    if (FUSED) {
      if constexpr (!FIRST) score_logdet();
      LQG_UNROLL for (int n = 0; n < NT; ++n) {
        innovate(n, t, !FIRST);
        R cvec[M];
        LQG_UNROLL for (int j = 0; j < O; ++j) cvec[j] = xt[j];
        LQG_UNROLL for (int p = 0; p < RR; ++p) {
          R v = muR[n][p];
          LQG_UNROLL for (int j = 0; j < O; ++j) v += U2[p * O + j] * w[j];
          cvec[O + p] = v;
        }
        R mn[M];
        dev_matvec_row<O, 0>(Fj, cvec, mn);                    // rows < O as deviation from x_t: ((Fj - I) cvec)[i]
        LQG_UNROLL for (int i = 0; i < O; ++i) { dO[n][i] = mn[i]; xprev[n][i] = xt[i]; }
        LQG_UNROLL for (int p = 0; p < RR; ++p) muR[n][p] = mn[O + p];
      }
    } else {
      OT* op = reinterpret_cast<OT*>(a.ops) + ((long)s * (a.T + 1) + t) * Ops::N;
      // Fj - I assembled as a deviation (the diagonal blocks start from A - I, then the small terms): see k_forward
      const auto FjD = block2x2(restrict_to<PAT::AdmI>(minus_identity(Ad)), mul(Bd, L), mul(K, FAd),
                                add(sub(minus_identity(Aa), mul(K, FAa)), mul(BK, L)));
      static_assert(mask_eq(decltype(FjD)::mask, trial_operator_mask<PAT, NX, NB, NU, NY, ND, DENSE_P>()),
                    "trial_operator_mask() must mirror the mask algebra of the operator built here");
      store_dense<0>(FjD, op + Ops::F_OFF);
      if constexpr (!std::is_same_v<OT, R>) {
        // MIXED: what the rounding to OT dropped from the block, and the block's largest entry (round 5)
        if (a.ops_lo) {
          if (hl_first < 0 && (t & 7) == 0) {          // (looked at every 8th step: a crossing is seen at most 7 steps late)
            static_assert(std::is_same_v<R, double>, "the MIXED builder runs in fp64");
            unsigned mx[2] = {0u, 0u};
            block_absmax_hi<0>(FjD, mx);
            if ((mx[0] > mx[1] ? mx[0] : mx[1]) >= (unsigned)__double2hiint((double)LQG_HILO_MIN)) hl_first = t;
          }
          if (hl_first >= 0) store_residual<0, OT>(FjD, a.ops_lo + ((long)s * (a.T + 1) + t) * (long)hilo_len<M>());
        }
      }
      LQG_UNROLL for (int i = 0; i < RR * O; ++i) op[Ops::U_OFF + i] = (OT)U2[i];
      {
        int e = 0;
        LQG_UNROLL for (int i = 0; i < O; ++i)
          LQG_UNROLL for (int j = 0; j <= i; ++j) op[Ops::L_OFF + (e++)] = (OT)Li[i * O + j];
      }
      store_or_nan(&op[Ops::H_OFF], (OT)(hl + kLogNorm), pois >= kPosFiniteLimit<R>);
    }
    // ---- Sigma' = F2 C F2^T + GG,  C = Srr - U2 U2^T                    system.py:223-230
    Mat<R, RR, RR> C;
    LQG_UNROLL for (int p = 0; p < RR; ++p)
      LQG_UNROLL for (int q = p; q < RR; ++q) {
        R v = Sg[(O + p) * M + O + q];
        LQG_UNROLL for (int j = 0; j < O; ++j) v -= U2[p * O + j] * U2[q * O + j];
        C.v[p * RR + q] = v;
        C.v[q * RR + p] = v;
      }
    const auto F2 = cols<O, RR>(Fj);
#if LQG_SP_PREFETCH
    {   // requests for step t+1 (row t+1 of x always exists; the last step re-reads its own gain row), issued BEFORE the
        // Sigma update below so that ~100 FMAs stand between the loads and the loop back-edge
      if constexpr (CK == 0) {
        const int tn = (t + 2 < a.T) ? t + 2 : a.T - 1;
        const R* src = a.Ls + (long)tn * (NU * NB) * a.ldb + s;
        LQG_UNROLL for (int e = 0; e < NU * NB; ++e) { Lnx[e] = Lnx2[e]; Lnx2[e] = src[e * a.ldb]; }
      }
      if (FUSED) {
        if constexpr (X4) {
          const float4 v = *reinterpret_cast<const float4*>(xp + (long)(t + 1) * a.x.st);
          const float vv[4] = {v.x, v.y, v.z, v.w};
          LQG_UNROLL for (int k = 0; k < NT; ++k)
            LQG_UNROLL for (int i = 0; i < O; ++i) xnx[k][i] = vv[k * O + i];
        } else {
          LQG_UNROLL for (int k = 0; k < NT; ++k) {
            const R* xr = xp + k * a.x.sn + (long)(t + 1) * a.x.st;
            LQG_UNROLL for (int i = 0; i < O; ++i) xnx[k][i] = xr[i * a.x.sd];
          }
        }
      }
    }
#endif
    to_dense(mul_nt_sym_add(mul(F2, C), F2, GG), Sg);
  };
  if constexpr (CK == 0) {
    step.template operator()<true, 0>(0);
    for (int t = 1; t < a.T; ++t) {
      step.template operator()<false, 0>(t);
      if (FUSED && (t & 7) == 0) flush();
    }
  } else {
    request_ckpt(0);
    // Whole chunks run WITHOUT a per-step `t < T` test (round 4): guarded steps are separate basic blocks, and the values the
    // software pipeline carries from one step to the next (data rows, previous row, mean state) were copied at every merge —
    // 14 v_mov + a scalar compare / branch per step in the ISA (scripts/isa_mix.py).  The first chunk (its first step
    // initialises Sigma and skips the score) and the last, partial chunk keep the guarded form.
    auto chunk_guarded = [&]<int... J>(int t0, std::integer_sequence<int, J...>) LQG_LAMBDA_INLINE {
      ((t0 + J < a.T ? ((J == 0 && t0 == 0) ? step.template operator()<true, J>(0)
                                             : step.template operator()<false, J>(t0 + J))
                     : (void)0), ...);
    };
    auto chunk_whole = [&]<int... J>(int t0, std::integer_sequence<int, J...>) LQG_LAMBDA_INLINE {
      (step.template operator()<false, J>(t0 + J), ...);
    };
    refill(0);
    chunk_guarded(0, std::make_integer_sequence<int, CKN>{});
    if (FUSED) flush();
    int t0 = CK;
    for (; t0 + CK <= a.T; t0 += CK) {
      refill.template operator()<true>(t0);
      chunk_whole(t0, std::make_integer_sequence<int, CKN>{});
      if (FUSED) flush();
    }
    if (t0 < a.T) {
      refill(t0);
      chunk_guarded(t0, std::make_integer_sequence<int, CKN>{});
    }
  }
  condition();
  if (FUSED) {
    score_logdet();
    LQG_UNROLL for (int n = 0; n < NT; ++n) innovate(n, a.T, true);
    flush();
    LQG_UNROLL for (int n = 0; n < NT; ++n) {
      acc[n] -= (double)a.T * (0.5 * ND * 1.8378770664093453);           // T scored rows x (d / 2) log 2 pi
      store_or_nan(&a.ll[s * a.ll_sb + n * ll_sn], (R)acc[n], pois >= kPosFiniteLimit<R>);
    }
  } else {
    OT* op = reinterpret_cast<OT*>(a.ops) + ((long)s * (a.T + 1) + a.T) * Ops::N;
    int e = 0;
    LQG_UNROLL for (int i = 0; i < O; ++i)
      LQG_UNROLL for (int j = 0; j <= i; ++j) op[Ops::L_OFF + (e++)] = (OT)Li[i * O + j];
    store_or_nan(&op[Ops::H_OFF], (OT)(hl + kLogNorm), pois >= kPosFiniteLimit<R>);
    if constexpr (!std::is_same_v<OT, R>) {
      if (a.hl) a.hl[s] = hl_first + 1;
    }
  }
}

// ---------------------------------------------------------------- TIME-VARYING specs, everything materialised (mode M2)
// lqg_solve_materialised through the generic dense kernels is bound by its ~3700 instructions per step at one wave per SIMD
// (DESIGN.md §6b).  When the time-varying specs keep ONE sparsity pattern (a zoo model whose entries move in time: the masks are
// the union over systems and steps, lqg_amd/specialize.py), the same two sweeps run on the pattern's masks: only the
// structurally non-zero entries are loaded per step, the step is the ~250 instructions of the headline path, and L, H, K,
// mu, Sigma are stored dense (structural zeros as zeros).  One trial per system, swept in-lane.
#ifndef LQG_TV_UNIFORM_BASE
#define LQG_TV_UNIFORM_BASE 1     // spec loads of the time-varying sweeps: wave-uniform entry pointer + lane offset (0: one per-lane pointer expression)
#endif
// CANON (round 6): every time-varying field is stored [T][row][col][system] with one leading dimension `ldc` (elements): the loads
// are scalar-base + lane-offset (lqg_sparse.hpp: load_masked_c) and none of the per-field strides but st is read.
template <typename R, int NB, int NU, typename PAT, bool CANON = false>
__global__ void __launch_bounds__(LQG_BLOCK, 2) k_riccati_tv_sp(const RiccatiArgs<R> a, const long ldc = 0) {
  const long s = blockIdx.x * (long)LQG_BLOCK + threadIdx.x;
  if (s >= a.n_sys) return;
  [[maybe_unused]] const unsigned ldb = (unsigned)(ldc * (long)sizeof(R)), lob = (unsigned)(s * (long)sizeof(R));
  R S[NB * NB];
  load_sym<R, NB>(a.Qf.p + s * a.Qf.sb, a.Qf.sr, a.Qf.sc, S);
  for (int t = a.T - 1; t >= 0; --t) {
#if LQG_TV_UNIFORM_BASE
    if constexpr (CANON) {
      const auto A = load_masked_c<R, NB, NB, PAT::Aa>(a.A.p + (long)t * a.A.st, ldb, lob);
      const auto Bm = load_masked_c<R, NB, NU, PAT::Ba>(a.B.p + (long)t * a.B.st, ldb, lob);
      const auto Q = load_sym_masked_c<R, NB, PAT::Q>(a.Q.p + (long)t * a.Q.st, ldb, lob);
      const auto Rm = load_sym_masked_c<R, NU, PAT::Rr>(a.Rm.p + (long)t * a.Rm.st, ldb, lob);
      R L[NU * NB];
      riccati_step_sp<R, NB, NU>(S, A, Bm, Q, Rm, a.eps, L);
      R* dst = a.Ls + (long)t * (NU * NB) * a.ldb + s;
      LQG_UNROLL for (int e = 0; e < NU * NB; ++e) dst[e * a.ldb] = L[e];
      continue;                                    // (the canonical instantiation serves the log-likelihood only: no P, no L / l / H outputs)
    }
    const auto A = load_masked_u<R, NB, NB, PAT::Aa>(a.A.p + (long)t * a.A.st, a.A.sr, a.A.sc, s * a.A.sb);
    const auto Bm = load_masked_u<R, NB, NU, PAT::Ba>(a.B.p + (long)t * a.B.st, a.B.sr, a.B.sc, s * a.B.sb);
    const auto Q = load_sym_masked_u<R, NB, PAT::Q>(a.Q.p + (long)t * a.Q.st, a.Q.sr, a.Q.sc, s * a.Q.sb);
    const auto Rm = load_sym_masked_u<R, NU, PAT::Rr>(a.Rm.p + (long)t * a.Rm.st, a.Rm.sr, a.Rm.sc, s * a.Rm.sb);
#else
    const auto A = load_masked<R, NB, NB, PAT::Aa>(a.A.p + s * a.A.sb + t * a.A.st, a.A.sr, a.A.sc);
    const auto Bm = load_masked<R, NB, NU, PAT::Ba>(a.B.p + s * a.B.sb + t * a.B.st, a.B.sr, a.B.sc);
    const auto Q = load_sym_masked<R, NB, PAT::Q>(a.Q.p + s * a.Q.sb + t * a.Q.st, a.Q.sr, a.Q.sc);
    const auto Rm = load_sym_masked<R, NU, PAT::Rr>(a.Rm.p + s * a.Rm.sb + t * a.Rm.st, a.Rm.sr, a.Rm.sc);
#endif
    R L[NU * NB], Ht[NU * NU], Pc[NU * NB];
    if (a.P.p) {                                   // cross cost u' P x (lqr.py:23).  q, qf, r only move the affine gain l and the
      const R* pp = a.P.p + s * a.P.sb + t * a.P.st;   // cost-to-go offset s (lqr.py:24, 31, 34): neither enters L, S or the likelihood
      LQG_UNROLL for (int i = 0; i < NU; ++i)
        LQG_UNROLL for (int j = 0; j < NB; ++j) Pc[i * NB + j] = pp[i * a.P.sr + j * a.P.sc];
    }
    riccati_step_sp<R, NB, NU>(S, A, Bm, Q, Rm, a.eps, L, Ht, a.P.p ? Pc : nullptr);
    if (a.Ls) {                                    // (null when the caller's L output doubles as the forward sweep's gain stream)
      R* dst = a.Ls + (long)t * (NU * NB) * a.ldb + s;
      LQG_UNROLL for (int e = 0; e < NU * NB; ++e) dst[e * a.ldb] = L[e];
    }
    if (a.L.p) store_mat<R, NU, NB>(const_cast<R*>(a.L.p) + s * a.L.sb + t * a.L.st, a.L.sr, a.L.sc, L);
    if (a.l.p) {
      R* ld = const_cast<R*>(a.l.p) + s * a.l.sb + t * a.l.st;
      LQG_UNROLL for (int i = 0; i < NU; ++i) ld[i * a.l.sr] = R(0);
    }
    if (a.H.p) store_mat<R, NU, NU>(const_cast<R*>(a.H.p) + s * a.H.sb + t * a.H.st, a.H.sr, a.H.sc, Ht);
  }
}

// FUSED = false (round 6: lqg_log_likelihood_sp for time-varying specs with several trials per system, or in the mixed mode): no
// trial is swept in-lane; the step's trial operator goes to the operator stream (element type OT, as k_forward_sp) for k_trial_sp.
#ifndef LQG_TV_PREFETCH
#define LQG_TV_PREFETCH 0
#endif
#ifndef LQG_TV_WAVES_F64
#define LQG_TV_WAVES_F64 1
#endif
template <typename R, int NX, int NB, int NU, int NY, int ND, typename PAT, bool DENSE_P, bool FUSED = true, typename OT = R, bool CANON = false>
__global__ void __launch_bounds__(LQG_BLOCK, sizeof(R) == 4 ? 2 : LQG_TV_WAVES_F64) k_forward_tv_sp(const ForwardArgs<R> a, const DView<R> Lv,
                                                                                               const long ldc = 0) {
  constexpr int M = NX + NB, O = ND, RR = M - ND;
  using Ops = TrialOps<M, ND>;
  const long s = blockIdx.x * (long)LQG_BLOCK + threadIdx.x;
  if (s >= a.n_sys) return;
  [[maybe_unused]] const unsigned ldb = (unsigned)(ldc * (long)sizeof(R)), lob = (unsigned)(s * (long)sizeof(R));
  constexpr auto PMASK = kalman_state_mask<PAT, NB, NY, DENSE_P>();
  Mat<R, NB, NB, PMASK> Pm;
  {
    R P0[NB * NB];
    if (DENSE_P && a.Sigma0.p) load_sym<R, NB>(a.Sigma0.p + s * a.Sigma0.sb, a.Sigma0.sr, a.Sigma0.sc, P0);
    else load_gram<R, NB>(a.aV.p + s * a.aV.sb, a.aV.sr, a.aV.sc, a.nva, P0);               // V[0] V[0]'   system.py:160
    LQG_UNROLL for (int i = 0; i < NB * NB; ++i)
      if (PMASK.b[i]) Pm.v[i] = P0[i];
  }
  R Sg[M * M], xprev[O], dO[O], muR[RR];
  double acc = 0.0;
  const R* xp = FUSED ? a.x.p + s * a.x.sb : nullptr;
  LQG_UNROLL for (int i = 0; i < O; ++i) { xprev[i] = FUSED ? xp[i * a.x.sd] : R(0); dO[i] = R(0); }
  LQG_UNROLL for (int i = 0; i < RR; ++i) muR[i] = R(0);
  const R kLogNorm = R(0.5 * ND * 1.8378770664093453);
  R Li[O * O], U2[RR * O], hl;
  unsigned pois = 0u;                             // largest pos_finite_key of the pivot products (lqg_small.hpp)
  auto condition = [&]() {
    R Soo[O * O], Lc[O * O], dinv[O];
    LQG_UNROLL for (int i = 0; i < O; ++i)
      LQG_UNROLL for (int j = 0; j < O; ++j) Soo[i * O + j] = Sg[i * M + j];
    chol_lower<R, O>(Soo, Lc, dinv);
    tri_inverse_lower<R, O>(Lc, dinv, Li);
    R pd = dinv[0];
    LQG_UNROLL for (int i = 1; i < O; ++i) pd *= dinv[i];
    {
      const unsigned key = pos_finite_key(pd);
      pois = key > pois ? key : pois;
    }
    hl = -log_<R>(pd);
    LQG_UNROLL for (int p = 0; p < RR; ++p)
      LQG_UNROLL for (int j = 0; j < O; ++j) {
        R v = R(0);
        LQG_UNROLL for (int k = 0; k <= j; ++k) v += Sg[(O + p) * M + k] * Li[j * O + k];
        U2[p * O + j] = v;
      }
  };
  R w[O], xt[O];
#if LQG_TV_PREFETCH
  // the data row of step t + 1 is requested at the start of step t as well (it is consumed in the middle of the step, by
  // `innovate`: requested there, its whole latency was exposed once per step)
  R xrow[O], xnx[O];                             // row of the step being run / row requested for the next one
  LQG_UNROLL for (int i = 0; i < O; ++i) { xnx[i] = FUSED ? xp[i * a.x.sd] : R(0); xrow[i] = xnx[i]; }
#endif
  auto innovate = [&](int row, bool score) {
#if LQG_TV_PREFETCH
    LQG_UNROLL for (int i = 0; i < O; ++i) xt[i] = xrow[i];
    (void)row;
#else
    const R* xr = xp + (long)row * a.x.st;
    LQG_UNROLL for (int i = 0; i < O; ++i) xt[i] = xr[i * a.x.sd];
#endif
    R zz = R(0);
    LQG_UNROLL for (int i = 0; i < O; ++i) {
      R v = R(0);
      LQG_UNROLL for (int j = 0; j <= i; ++j) v += Li[i * O + j] * ((xt[j] - xprev[j]) - dO[j]);
      w[i] = v;
      zz += v * v;
    }
    if (score) acc -= (double)(R(0.5) * zz + hl + kLogNorm);
  };
  // this step's specs: structurally non-zero entries only
#if LQG_TV_UNIFORM_BASE
  // wave-uniform entry pointers (scalar) + the lane's own element offset per field (lqg_sparse.hpp: load_masked_u)
  const long oAa = s * a.aA.sb, oBa = s * a.aB.sb, oFa = s * a.aF.sb, oVa = s * a.aV.sb, oWa = s * a.aW.sb;
  const long oAd = s * a.dA.sb, oBd = s * a.dB.sb, oFd = s * a.dF.sb, oVd = s * a.dV.sb, oWd = s * a.dW.sb;
  auto ldAa = [&](int t) { if constexpr (CANON) return load_masked_c<R, NB, NB, PAT::Aa>(a.aA.p + (long)t * a.aA.st, ldb, lob);
                           else return load_masked_u<R, NB, NB, PAT::Aa>(a.aA.p + (long)t * a.aA.st, a.aA.sr, a.aA.sc, oAa); };
  auto ldBa = [&](int t) { if constexpr (CANON) return load_masked_c<R, NB, NU, PAT::Ba>(a.aB.p + (long)t * a.aB.st, ldb, lob);
                           else return load_masked_u<R, NB, NU, PAT::Ba>(a.aB.p + (long)t * a.aB.st, a.aB.sr, a.aB.sc, oBa); };
  auto ldFa = [&](int t) { if constexpr (CANON) return load_masked_c<R, NY, NB, PAT::Fa>(a.aF.p + (long)t * a.aF.st, ldb, lob);
                           else return load_masked_u<R, NY, NB, PAT::Fa>(a.aF.p + (long)t * a.aF.st, a.aF.sr, a.aF.sc, oFa); };
  auto ldVVa = [&](int t) { if constexpr (CANON) return load_gram_masked_raw_c<R, NB, PAT::VVa, PAT::Va>(a.aV.p + (long)t * a.aV.st, a.nva, ldb, lob);
                            else return load_gram_masked_raw<R, NB, PAT::VVa, PAT::Va>(a.aV.p + (long)t * a.aV.st, a.aV.sr, a.aV.sc, a.nva, oVa); };
  auto ldWWa = [&](int t) { if constexpr (CANON) return load_gram_masked_raw_c<R, NY, PAT::WWa, PAT::Wa>(a.aW.p + (long)t * a.aW.st, a.nwa, ldb, lob);
                            else return load_gram_masked_raw<R, NY, PAT::WWa, PAT::Wa>(a.aW.p + (long)t * a.aW.st, a.aW.sr, a.aW.sc, a.nwa, oWa); };
  auto ldAd = [&](int t) { if constexpr (CANON) return load_masked_c<R, NX, NX, PAT::Ad>(a.dA.p + (long)t * a.dA.st, ldb, lob);
                           else return load_masked_u<R, NX, NX, PAT::Ad>(a.dA.p + (long)t * a.dA.st, a.dA.sr, a.dA.sc, oAd); };
  auto ldBd = [&](int t) { if constexpr (CANON) return load_masked_c<R, NX, NU, PAT::Bd>(a.dB.p + (long)t * a.dB.st, ldb, lob);
                           else return load_masked_u<R, NX, NU, PAT::Bd>(a.dB.p + (long)t * a.dB.st, a.dB.sr, a.dB.sc, oBd); };
  auto ldN1 = [&](int t) { if constexpr (CANON) return load_gram_masked_raw_c<R, NX, PAT::N1, PAT::Vd>(a.dV.p + (long)t * a.dV.st, a.nvd, ldb, lob);
                           else return load_gram_masked_raw<R, NX, PAT::N1, PAT::Vd>(a.dV.p + (long)t * a.dV.st, a.dV.sr, a.dV.sc, a.nvd, oVd); };
  auto ldFd = [&](int t) { if constexpr (CANON) return load_masked_c<R, NY, NX, PAT::Fd>(a.dF.p + (long)t * a.dF.st, ldb, lob);
                           else return load_masked_u<R, NY, NX, PAT::Fd>(a.dF.p + (long)t * a.dF.st, a.dF.sr, a.dF.sc, oFd); };
  auto ldWWd = [&](int t) { if constexpr (CANON) return load_gram_masked_raw_c<R, NY, PAT::WWd, PAT::Wd>(a.dW.p + (long)t * a.dW.st, a.nwd, ldb, lob);
                            else return load_gram_masked_raw<R, NY, PAT::WWd, PAT::Wd>(a.dW.p + (long)t * a.dW.st, a.dW.sr, a.dW.sc, a.nwd, oWd); };
#else
  auto ldAa = [&](int t) { return load_masked<R, NB, NB, PAT::Aa>(a.aA.p + s * a.aA.sb + t * a.aA.st, a.aA.sr, a.aA.sc); };
  auto ldBa = [&](int t) { return load_masked<R, NB, NU, PAT::Ba>(a.aB.p + s * a.aB.sb + t * a.aB.st, a.aB.sr, a.aB.sc); };
  auto ldFa = [&](int t) { return load_masked<R, NY, NB, PAT::Fa>(a.aF.p + s * a.aF.sb + t * a.aF.st, a.aF.sr, a.aF.sc); };
  auto ldVVa = [&](int t) { return load_gram_masked_raw<R, NB, PAT::VVa, PAT::Va>(a.aV.p + s * a.aV.sb + t * a.aV.st, a.aV.sr, a.aV.sc, a.nva); };
  auto ldWWa = [&](int t) { return load_gram_masked_raw<R, NY, PAT::WWa, PAT::Wa>(a.aW.p + s * a.aW.sb + t * a.aW.st, a.aW.sr, a.aW.sc, a.nwa); };
  auto ldAd = [&](int t) { return load_masked<R, NX, NX, PAT::Ad>(a.dA.p + s * a.dA.sb + t * a.dA.st, a.dA.sr, a.dA.sc); };
  auto ldBd = [&](int t) { return load_masked<R, NX, NU, PAT::Bd>(a.dB.p + s * a.dB.sb + t * a.dB.st, a.dB.sr, a.dB.sc); };
  auto ldN1 = [&](int t) { return load_gram_masked_raw<R, NX, PAT::N1, PAT::Vd>(a.dV.p + s * a.dV.sb + t * a.dV.st, a.dV.sr, a.dV.sc, a.nvd); };
  auto ldFd = [&](int t) { return load_masked<R, NY, NX, PAT::Fd>(a.dF.p + s * a.dF.sb + t * a.dF.st, a.dF.sr, a.dF.sc); };
  auto ldWWd = [&](int t) { return load_gram_masked_raw<R, NY, PAT::WWd, PAT::Wd>(a.dW.p + s * a.dW.sb + t * a.dW.st, a.dW.sr, a.dW.sc, a.nwd); };
#endif
  auto ldL = [&](int t) {
    Mat<R, NU, NB> L;                             // gains of step t: the caller's L array or the gain scratch, as a strided view
    if constexpr (CANON) {                        // (the gain scratch [T][NU NB][ldb] is canonical with its own leading dimension Lv.sc)
      LQG_UNROLL for (int e = 0; e < NU * NB; ++e)
        L.v[e] = canon_at<R>(Lv.p + (long)t * Lv.st, (unsigned)e, (unsigned)(Lv.sc * (long)sizeof(R)), lob);
      return L;
    }
    const R* src = Lv.p + s * Lv.sb + (long)t * Lv.st;
    LQG_UNROLL for (int i = 0; i < NU; ++i)
      LQG_UNROLL for (int j = 0; j < NB; ++j) L.v[i * NB + j] = src[i * Lv.sr + j * Lv.sc];
    return L;
  };
  // (measured on 2^17 systems — two waves per SIMD — and not kept: requesting the next step's rows one or four steps ahead
  // (17.6 -> 19.0 ms forward, 5.4 -> 6.0 ms Riccati) and running the Kalman step of t + 1 inside iteration t as a second
  // dependency chain (17.8 -> 18.9 ms))
#if LQG_TV_PREFETCH
  // the specs of step t + 1 are requested at the START of step t (one wave or two per SIMD have nothing else to hide a step's
  // ~25 dependent loads behind: the log-likelihood route of round 6 ran 5.7 us per step in fp64 without this)
  auto nAa = ldAa(0); auto nBa = ldBa(0); auto nFa = ldFa(0); auto nVVa = ldVVa(0); auto nWWa = ldWWa(0);
  auto nAd = ldAd(0); auto nBd = ldBd(0); auto nN1 = ldN1(0); auto nFd = ldFd(0); auto nWWd = ldWWd(0);
  Mat<R, NU, NB> nL = ldL(0);
#endif
  auto step = [&]<bool FIRST>(int t) {
#if LQG_TV_PREFETCH
    const auto Aa = nAa;  const auto Ba = nBa;  const auto Fa = nFa;  const auto VVa = nVVa;  const auto WWa = nWWa;
    const auto Ad = nAd;  const auto Bd = nBd;  const auto N1 = nN1;  const auto Fd = nFd;  const auto WWd = nWWd;
    const Mat<R, NU, NB> L = nL;
    if constexpr (FUSED) {                       // row t was requested a step ago: hand it to this step; request row t + 1 (rows run to T)
      LQG_UNROLL for (int i = 0; i < O; ++i) xrow[i] = xnx[i];
      const R* xr = xp + (long)(t + 1) * a.x.st;
      LQG_UNROLL for (int i = 0; i < O; ++i) xnx[i] = xr[i * a.x.sd];
    }
    {
      const int tn = t + 1 < a.T ? t + 1 : t;
      nAa = ldAa(tn); nBa = ldBa(tn); nFa = ldFa(tn); nVVa = ldVVa(tn); nWWa = ldWWa(tn);
      nAd = ldAd(tn); nBd = ldBd(tn); nN1 = ldN1(tn); nFd = ldFd(tn); nWWd = ldWWd(tn);
      nL = ldL(tn);
    }
#else
    const auto Aa = ldAa(t);
    const auto Ba = ldBa(t);
    const auto Fa = ldFa(t);
    const auto VVa = ldVVa(t);
    const auto WWa = ldWWa(t);
    const auto Ad = ldAd(t);
    const auto Bd = ldBd(t);
    const auto N1 = ldN1(t);
    const auto Fd = ldFd(t);
    const auto WWd = ldWWd(t);
    const Mat<R, NU, NB> L = ldL(t);
#endif
    const auto FAa = restrict_to<PAT::FAa>(mul(Fa, Aa));
    const auto FAd = restrict_to<PAT::FAd>(mul(Fd, Ad));
    const auto DB = restrict_to<PAT::DB>(sub(mul(Fd, Bd), mul(Fa, Ba)));
    const auto N2 = restrict_to<PAT::N2>(mul(Fd, N1));
    const auto N3 = restrict_to<PAT::N3>(mul_nt_sym_add(mul(Fd, N1), Fd, WWd));
    // ---- Kalman step                                                   kf.py:10-14
    const auto AP = mul(Aa, Pm);
    const auto Pp = mul_nt_sym_add(AP, Aa, VVa);
    const auto FP = mul(Fa, Pp);
    const auto Gi = spd_inverse_masked(mul_nt_sym_add(FP, Fa, WWa));
    const auto K = mul_tn(FP, Gi);
    assign_state(Pm, sym_sub_mul(Pp, K, FP));
    if (!CANON && a.Kout.p) {                    // (the canonical instantiation serves the log-likelihood: nothing materialised)
      R Kd[NB * NY];
      to_dense(K, Kd);
      store_mat<R, NB, NY>(const_cast<R*>(a.Kout.p) + s * a.Kout.sb + t * a.Kout.st, a.Kout.sr, a.Kout.sc, Kd);
    }
    // ---- joint dynamics and noise covariance                           system.py:167-207
    const auto BK = add(Ba, mul(K, DB));
    const auto Fj = block2x2(Ad, mul(Bd, L), mul(K, FAd), add(sub(Aa, mul(K, FAa)), mul(BK, L)));
    static_assert(mask_eq(decltype(Fj)::mask, joint_dynamics_mask<PAT, NX, NB, NU, NY, DENSE_P>()),
                  "joint_dynamics_mask() must mirror the mask algebra of the joint system built here");
    const auto KN2 = mul(K, N2);
    const auto GG = block2x2(N1, transpose(KN2), KN2, mul_nt_sym_add(mul(K, N3), K, Mat<R, NB, NB, mask_none<NB, NB>()>{}));
    if constexpr (FIRST) to_dense(GG, Sg);                               // Sigma0 := G[0] G[0]^T  system.py:212
    // ---- condition on x_t, score it, propagate the mean                system.py:219-221, 244-248
    condition();
    if constexpr (FUSED) {
      innovate(t, !FIRST);
      R cvec[M], mn[M];
      LQG_UNROLL for (int j = 0; j < O; ++j) cvec[j] = xt[j];
      LQG_UNROLL for (int p = 0; p < RR; ++p) {
        R v = muR[p];
        LQG_UNROLL for (int j = 0; j < O; ++j) v += U2[p * O + j] * w[j];
        cvec[O + p] = v;
      }
      dev_matvec_row<O, 0>(Fj, cvec, mn);                                  // rows < O as deviation from x_t
      LQG_UNROLL for (int i = 0; i < O; ++i) { dO[i] = mn[i]; xprev[i] = xt[i]; }
      LQG_UNROLL for (int p = 0; p < RR; ++p) muR[p] = mn[O + p];
      if (!CANON && a.mu.p) {
        R* dst = const_cast<R*>(a.mu.p) + s * a.mu.sb + (long)t * a.mu.st;
        LQG_UNROLL for (int i = 0; i < M; ++i) dst[i * a.mu.sd] = (i < O) ? xt[i] + mn[i] : mn[i];
      }
    } else {                                                               // the step's trial operator (as k_forward_sp)
      OT* op = reinterpret_cast<OT*>(a.ops) + ((long)s * (a.T + 1) + t) * Ops::N;
      const auto FjD = block2x2(restrict_to<PAT::AdmI>(minus_identity(Ad)), mul(Bd, L), mul(K, FAd),
                                add(sub(minus_identity(Aa), mul(K, FAa)), mul(BK, L)));
      static_assert(mask_eq(decltype(FjD)::mask, trial_operator_mask<PAT, NX, NB, NU, NY, ND, DENSE_P>()),
                    "trial_operator_mask() must mirror the mask algebra of the operator built here");
      store_dense<0>(FjD, op + Ops::F_OFF);
      LQG_UNROLL for (int i = 0; i < RR * O; ++i) op[Ops::U_OFF + i] = (OT)U2[i];
      int e = 0;
      LQG_UNROLL for (int i = 0; i < O; ++i)
        LQG_UNROLL for (int j = 0; j <= i; ++j) op[Ops::L_OFF + (e++)] = (OT)Li[i * O + j];
      store_or_nan(&op[Ops::H_OFF], (OT)(hl + kLogNorm), pois >= kPosFiniteLimit<R>);
    }
    // ---- Sigma' = F2 C F2^T + GG,  C = Srr - U2 U2^T                    system.py:223-230
    Mat<R, RR, RR> C;
    LQG_UNROLL for (int p = 0; p < RR; ++p)
      LQG_UNROLL for (int q = p; q < RR; ++q) {
        R v = Sg[(O + p) * M + O + q];
        LQG_UNROLL for (int j = 0; j < O; ++j) v -= U2[p * O + j] * U2[q * O + j];
        C.v[p * RR + q] = v;
        C.v[q * RR + p] = v;
      }
    const auto F2 = cols<O, RR>(Fj);
    to_dense(mul_nt_sym_add(mul(F2, C), F2, GG), Sg);
    if (!CANON && a.Sig.p) store_mat<R, M, M>(const_cast<R*>(a.Sig.p) + s * a.Sig.sb + t * a.Sig.st, a.Sig.sr, a.Sig.sc, Sg);
  };
  step.template operator()<true>(0);
  for (int t = 1; t < a.T; ++t) step.template operator()<false>(t);
  condition();
  if constexpr (FUSED) {
#if LQG_TV_PREFETCH
    LQG_UNROLL for (int i = 0; i < O; ++i) xrow[i] = xnx[i];           // row T, requested during the last step
#endif
    innovate(a.T, true);
    if (a.ll) store_or_nan(&a.ll[s * a.ll_sb], (R)acc, pois >= kPosFiniteLimit<R>);
  } else {
    OT* op = reinterpret_cast<OT*>(a.ops) + ((long)s * (a.T + 1) + a.T) * Ops::N;
    int e = 0;
    LQG_UNROLL for (int i = 0; i < O; ++i)
      LQG_UNROLL for (int j = 0; j <= i; ++j) op[Ops::L_OFF + (e++)] = (OT)Li[i * O + j];
    store_or_nan(&op[Ops::H_OFF], (OT)(hl + kLogNorm), pois >= kPosFiniteLimit<R>);
  }
}

// ---------------------------------------------------------------- per-trial sweep with the operator's structure
// k_trial (lqg_kernels.hpp) with the structural zeros of the operator Fj - I (FM) resolved at compile time:
// the mean update is ~half of a trial-step's instructions and the tracking models' operators are 40-60 % dense (unit
// diagonal of A, a single control row, selection-shaped K Fd Ad).  Same operator stream, same arithmetic order for the
// terms that remain; rows without any term make their deviation a compile-time zero.  (Compile-time recursion instead
// of `if` inside unrolled loops: see dev_matvec_row in lqg_sparse.hpp.)
// compile-time mask as the operator-structure policy of the time-chunked sweep (lqg_trial_chunk.hpp)
template <int M, Mask<M, M> FM>
struct MaskPolicy {
  static constexpr bool at(int i, int j) { return FM.b[i * M + j]; }
};

#ifndef LQG_TRIAL_OPS_PREFETCH_MAX
#define LQG_TRIAL_OPS_PREFETCH_MAX 36      // dwords per operator block held twice in SGPRs (m = 5 in fp32: config 5 per-trial sweep 1.67 -> 1.46 ms)
#endif
template <bool PF, int NPF, int IDX, typename R>
LQG_DEV R trial_op_at(const R (&opc)[NPF], const R* __restrict__ op) {
  if constexpr (PF) return opc[IDX];
  else return op[IDX];
}
template <typename R, int M, int ND, Mask<M, M> FM, bool PF, int NPF, int I, int J>
LQG_DEV void trial_mean_term(const R (&opc)[NPF], const R* __restrict__ op, const R (&cv)[M], R& v) {
  if constexpr (J < M) {
    if constexpr (FM.b[I * M + J]) v += trial_op_at<PF, NPF, TrialOps<M, ND>::F_OFF + I * M + J, R>(opc, op) * cv[J];
    trial_mean_term<R, M, ND, FM, PF, NPF, I, J + 1>(opc, op, cv, v);
  }
}
template <typename R, int M, int ND, Mask<M, M> FM, bool PF, int NPF, int I>
LQG_DEV void trial_mean_rows(const R (&opc)[NPF], const R* __restrict__ op, const R (&cv)[M], R (&mn)[M]) {
  if constexpr (I < M) {
    R v = R(0);
    trial_mean_term<R, M, ND, FM, PF, NPF, I, 0>(opc, op, cv, v);
    mn[I] = v;
    trial_mean_rows<R, M, ND, FM, PF, NPF, I + 1>(opc, op, cv, mn);
  }
}

// hi + lo operators (HL): mn += lo (Fj - I block's rounding residual, ForwardArgs::ops_lo) x cv, summed apart from the main
// terms and added last
template <typename R, int M, Mask<M, M> FM, int I, int J>
LQG_DEV void trial_lo_term(const float* __restrict__ lo, const R (&cv)[M], R& v) {
  if constexpr (J < M) {
    if constexpr (FM.b[I * M + J]) v += (R)lo[I * M + J] * cv[J];
    trial_lo_term<R, M, FM, I, J + 1>(lo, cv, v);
  }
}
template <typename R, int M, Mask<M, M> FM, int I>
LQG_DEV void trial_lo_rows(const float* __restrict__ lo, const R (&cv)[M], R (&mn)[M]) {
  if constexpr (I < M) {
    R v = R(0);
    trial_lo_term<R, M, FM, I, 0>(lo, cv, v);
    mn[I] += v;
    trial_lo_rows<R, M, FM, I + 1>(lo, cv, mn);
  }
}

// CKT > 0: the restart data of every CKT-th row is also KEPT (TrialArgs::tck, see keepc) — the forward pass of the reverse-mode sweep.
// BLK: lanes per workgroup.  64 by default; 512 / 1024 put 1024+ trials of ONE candidate into one workgroup, whose waves walk
// the candidate's operator stream together through the CU's scalar cache (one fetch per CU instead of one per 128 trials).
// HL (MIXED mode, TrialArgs::hl set): the launch with HL walks the systems flagged by the builder (hl[sys] = 1 + the first step
// whose |Fj - I| block is large) and applies hi + lo operators from that step on; the launch without walks the others — the
// unflagged systems' loop is untouched.
template <typename R, int M, int ND, int TPL, Mask<M, M> FM, int CKT = 0, int BLK = LQG_BLOCK, bool HL = false>
__global__ void __launch_bounds__(BLK) k_trial_sp(const R* __restrict__ ops_all, const TrialArgs<R> a) {
  constexpr int O = ND, RR = M - ND;
  constexpr int kAccChunk = 8;
  using Ops = TrialOps<M, ND>;
  static_assert(!HL || (sizeof(R) == 4 && CKT == 0), "hi + lo operators: the fp32 per-trial sweep of the MIXED mode");
  const long sys = blockIdx.y;
  if (a.hl) {
    if ((a.hl[sys] != 0) != HL) return;
  }
  [[maybe_unused]] const float* __restrict__ lo_sys = HL ? a.ops_lo + sys * (long)(a.T + 1) * hilo_len<M>() : nullptr;
  [[maybe_unused]] const int lo_from = HL ? a.hl[sys] - 1 : 0;    // residual blocks exist (and matter) from this step on
  const long n0 = (long)blockIdx.x * (BLK * TPL) + threadIdx.x;
  const R* __restrict__ op = ops_all + sys * (long)(a.T + 1) * Ops::N;
  const R* xr[TPL];
  bool live[TPL];
  R xprev[TPL][O], dO[TPL][O], muR[TPL][RR];
  double acc[TPL];
  R part[TPL];
  LQG_UNROLL for (int k = 0; k < TPL; ++k) {
    long n = n0 + (long)k * BLK;
    live[k] = n < a.n_trials;
    n = live[k] ? n : (a.n_trials - 1);
    xr[k] = a.x.p + sys * a.x.sb + n * a.x.sn;
    LQG_UNROLL for (int i = 0; i < O; ++i) { xprev[k][i] = xr[k][i * a.x.sd]; dO[k][i] = R(0); }
    LQG_UNROLL for (int i = 0; i < RR; ++i) muR[k][i] = R(0);
    acc[k] = 0.0;
    part[k] = R(0);
  }
  R xq[TPL][O];                                           // data rows one step ahead
  LQG_UNROLL for (int k = 0; k < TPL; ++k)
    LQG_UNROLL for (int i = 0; i < O; ++i) xq[k][i] = xprev[k][i];
  constexpr bool PF = LQG_TRIAL_OPS_PREFETCH && (Ops::N * (int)(sizeof(R) / 4) <= LQG_TRIAL_OPS_PREFETCH_MAX);
  constexpr int NPF = PF ? Ops::N : 1;
  // Two loop structures (round 4).  Small joint dimensions (M < 8: the 1-D tracking models of configs 3 and 5) keep the
  // two-step loop with run-time first / last tests below: the sweep lives on latency hiding at ~60 VGPRs, and blocks of 8
  // unguarded steps — 30 instead of 36 VALU instructions per trial-step — let the compiler hoist the block's data loads to
  // 117 VGPRs: config 3 3.54 -> 4.59 ms (measured).  Larger ones (M >= 8: the 2-D hand model of config 4,
  // m = 10) are arithmetic-heavy and already register-bound: first / last step peeled, blocks of 8 unguarded steps, one
  // flush of the fp32 partial sums per block: config 4 (262 144 trials) per-trial sweep 1.92 -> 1.35 ms.
  // CKT > 0: what the reverse-mode sweep restarts a chunk from.  The mean state (dO, muR) entering row t + 1 is a function of
  // this step's (x_t, c_t) and operator — so the RR reals of c_t are kept (not the M of the state: half the checkpoint traffic
  // for the tracking models), as record (t + 1) / CKT when t + 1 starts a chunk, and as record nckt for the last row.
  [[maybe_unused]] auto keepc = [&](int t, int k, const R (&cv)[M]) LQG_LAMBDA_INLINE {
    if constexpr (CKT > 0) {
      const int tn = t + 1;
      if ((tn % CKT) == 0 || tn == a.T) {
        if (live[k]) {
          const int rec = tn == a.T ? a.nckt : tn / CKT;
          R* dst = a.tck + ((sys * (a.nckt + 1) + rec) * RR) * a.npad + n0 + (long)k * BLK;
          LQG_UNROLL for (int i = 0; i < RR; ++i) dst[i * a.npad] = cv[O + i];
        }
      }
    }
  };
  constexpr bool BLOCK8 = M >= 8;
  if constexpr (BLOCK8) {
    auto body8 = [&]<bool FIRST, bool LAST>(int t, const R (&cur)[NPF], const R* __restrict__ opt) LQG_LAMBDA_INLINE {
#define LQG_OP(i_) (PF ? cur[PF ? (i_) : 0] : opt[i_])
      R Li[O * (O + 1) / 2];
      LQG_UNROLL for (int i = 0; i < O * (O + 1) / 2; ++i) Li[i] = LQG_OP(Ops::L_OFF + i);
      const R hlc = LQG_OP(Ops::H_OFF);
      LQG_UNROLL for (int k = 0; k < TPL; ++k) {
        R cv[M], w[O];                                      // cv = [x_t ; c]
        LQG_UNROLL for (int i = 0; i < O; ++i) cv[i] = xq[k][i];
        if constexpr (!LAST) {
          LQG_UNROLL for (int i = 0; i < O; ++i) xq[k][i] = xr[k][(long)(t + 1) * a.x.st + i * a.x.sd];
        }
        R zz = R(0);
        {
          int e = 0;
          LQG_UNROLL for (int i = 0; i < O; ++i) {
            R v = R(0);
            LQG_UNROLL for (int j = 0; j <= i; ++j) v += Li[e++] * ((cv[j] - xprev[k][j]) - dO[k][j]);
            w[i] = v;
            zz += v * v;
          }
        }
        if constexpr (!FIRST) part[k] += R(0.5) * zz + hlc;
        if constexpr (!LAST) {
          LQG_UNROLL for (int p = 0; p < RR; ++p) {
            R v = muR[k][p];
            LQG_UNROLL for (int j = 0; j < O; ++j) v += LQG_OP(Ops::U_OFF + p * O + j) * w[j];
            cv[O + p] = v;
          }
          keepc(t, k, cv);
          R mn[M];
          trial_mean_rows<R, M, ND, FM, PF, NPF, 0>(cur, opt, cv, mn);
          if constexpr (HL) {
            if (t >= lo_from) trial_lo_rows<R, M, FM, 0>(lo_sys + (long)t * hilo_len<M>(), cv, mn);
          }
          LQG_UNROLL for (int i = 0; i < O; ++i) { dO[k][i] = mn[i]; xprev[k][i] = cv[i]; }
          LQG_UNROLL for (int p = 0; p < RR; ++p) muR[k][p] = cv[O + p] + mn[O + p];   // (the stream holds Fj - I)
        }
      }
#undef LQG_OP
    };
    auto flush = [&]() LQG_LAMBDA_INLINE {
      LQG_UNROLL for (int k = 0; k < TPL; ++k) { acc[k] -= (double)part[k]; part[k] = R(0); }
    };
    static_assert(kAccChunk == 8, "the unguarded block is eight steps");
    if constexpr (PF) {
      // step t reads bufA when t is even, bufB when odd, and requests the block of step t + 1 into the other buffer first;
      // blocks start at odd t, so the eight steps of a block have fixed buffers
      R bufA[NPF], bufB[NPF];
      auto fetch = [&](R (&dst)[NPF], int row) LQG_LAMBDA_INLINE {
        const R* __restrict__ src = op + (long)(row <= a.T ? row : a.T) * Ops::N;
        LQG_UNROLL for (int i = 0; i < NPF; ++i) dst[i] = src[i];
      };
      fetch(bufA, 0);
      fetch(bufB, 1);
      body8.template operator()<true, false>(0, bufA, op);
      int t = 1;
      for (; t + 8 <= a.T; t += 8) {
        LQG_UNROLL for (int j = 0; j < 8; j += 2) {
          fetch(bufA, t + j + 1);
          body8.template operator()<false, false>(t + j, bufB, op);
          fetch(bufB, t + j + 2);
          body8.template operator()<false, false>(t + j + 1, bufA, op);
        }
        flush();
      }
      for (; t < a.T; ++t) {                                 // guarded tail (< 8 steps)
        if (t & 1) { fetch(bufA, t + 1); body8.template operator()<false, false>(t, bufB, op); }
        else { fetch(bufB, t + 1); body8.template operator()<false, false>(t, bufA, op); }
      }
      if (a.T & 1) body8.template operator()<false, true>(a.T, bufB, op);
      else body8.template operator()<false, true>(a.T, bufA, op);
    } else {
      const R none[1] = {R(0)};
      body8.template operator()<true, false>(0, none, op);
      int t = 1;
      for (; t + 8 <= a.T; t += 8) {
        LQG_UNROLL for (int j = 0; j < 8; ++j) body8.template operator()<false, false>(t + j, none, op + (long)(t + j) * Ops::N);
        flush();
      }
      for (; t < a.T; ++t) body8.template operator()<false, false>(t, none, op + (long)t * Ops::N);
      body8.template operator()<false, true>(a.T, none, op + (long)a.T * Ops::N);
    }
    flush();
  } else {
    // one step of the sweep; `cur` holds the step's operator block when it is prefetched into SGPRs (PF), else the block is
    // read through `opt`
    auto body = [&](int t, const R (&cur)[NPF], const R* __restrict__ opt) {
  #define LQG_OP(i_) (PF ? cur[PF ? (i_) : 0] : opt[i_])
      R Li[O * (O + 1) / 2];
      LQG_UNROLL for (int i = 0; i < O * (O + 1) / 2; ++i) Li[i] = LQG_OP(Ops::L_OFF + i);
      const R hlc = LQG_OP(Ops::H_OFF);
      const bool flush = ((t & (kAccChunk - 1)) == 0) || t == a.T;
      LQG_UNROLL for (int k = 0; k < TPL; ++k) {
        R cv[M], w[O];                                      // cv = [x_t ; c]
        LQG_UNROLL for (int i = 0; i < O; ++i) cv[i] = xq[k][i];
        {
          const long row = (t + 1 < a.T) ? (long)(t + 1) : (long)a.T;
          LQG_UNROLL for (int i = 0; i < O; ++i) xq[k][i] = xr[k][row * a.x.st + i * a.x.sd];
        }
        R zz = R(0);
        {
          int e = 0;
          LQG_UNROLL for (int i = 0; i < O; ++i) {
            R v = R(0);
            LQG_UNROLL for (int j = 0; j <= i; ++j) v += Li[e++] * ((cv[j] - xprev[k][j]) - dO[k][j]);
            w[i] = v;
            zz += v * v;
          }
        }
        if (t > 0) part[k] += R(0.5) * zz + hlc;
        if (flush) { acc[k] -= (double)part[k]; part[k] = R(0); }
        if (t < a.T) {
          LQG_UNROLL for (int p = 0; p < RR; ++p) {
            R v = muR[k][p];
            LQG_UNROLL for (int j = 0; j < O; ++j) v += LQG_OP(Ops::U_OFF + p * O + j) * w[j];
            cv[O + p] = v;
          }
          keepc(t, k, cv);
          R mn[M];
          trial_mean_rows<R, M, ND, FM, PF, NPF, 0>(cur, opt, cv, mn);
          if constexpr (HL) {
            if (t >= lo_from) trial_lo_rows<R, M, FM, 0>(lo_sys + (long)t * hilo_len<M>(), cv, mn);
          }
          LQG_UNROLL for (int i = 0; i < O; ++i) { dO[k][i] = mn[i]; xprev[k][i] = cv[i]; }
          LQG_UNROLL for (int p = 0; p < RR; ++p) muR[k][p] = cv[O + p] + mn[O + p];   // (the stream holds Fj - I)
        }
      }
  #undef LQG_OP
    };
    if constexpr (PF) {
      // Double-buffered operator blocks in SGPRs, the loop unrolled by two so that the buffers trade places instead of being
      // copied (the copy was Ops::N s_mov per step: 44 scalar instructions per step against 68 vector ones on config 3)
      R bufA[NPF], bufB[NPF];
      auto fetch = [&](R (&dst)[NPF], int row) {
        const R* __restrict__ src = op + (long)(row <= a.T ? row : a.T) * Ops::N;
        LQG_UNROLL for (int i = 0; i < NPF; ++i) dst[i] = src[i];
      };
      fetch(bufA, 0);
      int t = 0;
      for (; t + 1 <= a.T; t += 2) {
        fetch(bufB, t + 1);
        body(t, bufA, op);
        fetch(bufA, t + 2);
        body(t + 1, bufB, op);
      }
      if (t <= a.T) body(t, bufA, op);
    } else {
      const R none[1] = {R(0)};
      for (int t = 0; t <= a.T; ++t) body(t, none, op + (long)t * Ops::N);
    }
  }
  if (a.ll) {
    LQG_UNROLL for (int k = 0; k < TPL; ++k)
      if (live[k]) a.ll[sys * a.ll_sb + (n0 + (long)k * BLK) * a.ll_sn] = (R)acc[k];
  }
}

// ---------------------------------------------------------------- per-trial sweep, operator stream staged in LDS (round 5)
// k_trial_sp gives each 64-lane workgroup 64 x TPL trials and reads the step's operator block through the scalar cache: with
// 1024 trials per candidate (BASELINE config 3) eight workgroups re-stream every candidate's whole operator stream — 6.8 GB of
// HBM traffic per launch against 25 MB of data rows (round-4 PMC), neither VALU- nor HBM-bound: stalled on those loads.  Here ONE
// 256-lane workgroup owns up to 256 x TPL trials of one candidate; the operator blocks of CKL steps are fetched ONCE per
// workgroup by coalesced vector loads into LDS (double-buffered: the next chunk is requested while this one is walked) and
// read back as LDS broadcasts.  Same arithmetic, same order as k_trial_sp: results agree bitwise (tests/test_gpu_parity.py).
// CKT > 0 keeps the restart data of every CKT-th row (TrialArgs::tck), as k_trial_sp<..., CKT>.
#ifndef LQG_TRIAL_LDS_BLOCK
#define LQG_TRIAL_LDS_BLOCK 256
#endif
#ifndef LQG_TRIAL_LDS_TPL
#define LQG_TRIAL_LDS_TPL 4
#endif
#ifndef LQG_TRIAL_LDS_CHUNK
#define LQG_TRIAL_LDS_CHUNK 8
#endif
template <typename R, int M, int ND, int TPL, Mask<M, M> FM, int CKT = 0>
__global__ void __launch_bounds__(LQG_TRIAL_LDS_BLOCK) k_trial_lds(const R* __restrict__ ops_all, const TrialArgs<R> a) {
  constexpr int O = ND, RR = M - ND, BLK = LQG_TRIAL_LDS_BLOCK, CKL = LQG_TRIAL_LDS_CHUNK;
  constexpr int kAccChunk = 8;
  using Ops = TrialOps<M, ND>;
  constexpr int CKN = CKL * Ops::N, NLD = (CKN + BLK - 1) / BLK;
  __shared__ R lops[2][CKN];
  const long sys = blockIdx.y;
  if (a.hl) {
    if (a.hl[sys] != 0) return;                              // (flagged systems: k_trial_sp<..., HL>)
  }
  const long n0 = (long)blockIdx.x * (BLK * TPL) + threadIdx.x;
  const R* __restrict__ op = ops_all + sys * (long)(a.T + 1) * Ops::N;
  const long op_len = (long)(a.T + 1) * Ops::N;
  const R* xr[TPL];
  bool live[TPL];
  R xprev[TPL][O], dO[TPL][O], muR[TPL][RR], part[TPL];
  double acc[TPL];
  LQG_UNROLL for (int k = 0; k < TPL; ++k) {
    long n = n0 + (long)k * BLK;
    live[k] = n < a.n_trials;
    n = live[k] ? n : (a.n_trials - 1);
    xr[k] = a.x.p + sys * a.x.sb + n * a.x.sn;
    LQG_UNROLL for (int i = 0; i < O; ++i) { xprev[k][i] = xr[k][i * a.x.sd]; dO[k][i] = R(0); }
    LQG_UNROLL for (int i = 0; i < RR; ++i) muR[k][i] = R(0);
    acc[k] = 0.0;
    part[k] = R(0);
  }
  R xq[TPL][O];                                           // data rows one step ahead
  LQG_UNROLL for (int k = 0; k < TPL; ++k)
    LQG_UNROLL for (int i = 0; i < O; ++i) xq[k][i] = xprev[k][i];
  R nx[NLD];
  LQG_UNROLL for (int q = 0; q < NLD; ++q) nx[q] = R(0);
  auto request = [&](int c) LQG_LAMBDA_INLINE {
    const long base = (long)c * CKN;
    LQG_UNROLL for (int q = 0; q < NLD; ++q) {
      const int i = q * BLK + (int)threadIdx.x;
      if (i < CKN && base + i < op_len) nx[q] = op[base + i];
    }
  };
  auto publish = [&](int buf) LQG_LAMBDA_INLINE {
    LQG_UNROLL for (int q = 0; q < NLD; ++q) {
      const int i = q * BLK + (int)threadIdx.x;
      if (i < CKN) lops[buf][i] = nx[q];
    }
  };
  // CKT > 0: what the reverse-mode sweep restarts a chunk from.  The mean state (dO, muR) entering row t + 1 is a function of
  // this step's (x_t, c_t) and operator — so the RR reals of c_t are kept (not the M of the state: half the checkpoint traffic
  // for the tracking models), as record (t + 1) / CKT when t + 1 starts a chunk, and as record nckt for the last row.
  [[maybe_unused]] auto keepc = [&](int t, int k, const R (&cv)[M]) LQG_LAMBDA_INLINE {
    if constexpr (CKT > 0) {
      const int tn = t + 1;
      if ((tn % CKT) == 0 || tn == a.T) {
        if (live[k]) {
          const int rec = tn == a.T ? a.nckt : tn / CKT;
          R* dst = a.tck + ((sys * (a.nckt + 1) + rec) * RR) * a.npad + n0 + (long)k * BLK;
          LQG_UNROLL for (int i = 0; i < RR; ++i) dst[i * a.npad] = cv[O + i];
        }
      }
    }
  };
  const R none[1] = {R(0)};
  const int nchunk = (a.T + 1 + CKL - 1) / CKL;           // rows 0 .. T
  int obuf = 0;
  request(0);
  for (int c = 0; c < nchunk; ++c) {
    publish(obuf);
    __syncthreads();
    if (c + 1 < nchunk) request(c + 1);
    LQG_UNROLL for (int j = 0; j < CKL; ++j) {
      const int t = c * CKL + j;
      if (t <= a.T) {
        const R* __restrict__ opt = lops[obuf] + j * Ops::N;
        R Li[O * (O + 1) / 2];
        LQG_UNROLL for (int i = 0; i < O * (O + 1) / 2; ++i) Li[i] = opt[Ops::L_OFF + i];
        const R hlc = opt[Ops::H_OFF];
        const bool flush = ((t & (kAccChunk - 1)) == 0) || t == a.T;
        LQG_UNROLL for (int k = 0; k < TPL; ++k) {
          R cv[M], w[O];                                      // cv = [x_t ; c]
          LQG_UNROLL for (int i = 0; i < O; ++i) cv[i] = xq[k][i];
          {
            const long row = (t + 1 < a.T) ? (long)(t + 1) : (long)a.T;
            LQG_UNROLL for (int i = 0; i < O; ++i) xq[k][i] = xr[k][row * a.x.st + i * a.x.sd];
          }
          R zz = R(0);
          {
            int e = 0;
            LQG_UNROLL for (int i = 0; i < O; ++i) {
              R v = R(0);
              LQG_UNROLL for (int q = 0; q <= i; ++q) v += Li[e++] * ((cv[q] - xprev[k][q]) - dO[k][q]);
              w[i] = v;
              zz += v * v;
            }
          }
          if (t > 0) part[k] += R(0.5) * zz + hlc;
          if (flush) { acc[k] -= (double)part[k]; part[k] = R(0); }
          if (t < a.T) {
            LQG_UNROLL for (int p = 0; p < RR; ++p) {
              R v = muR[k][p];
              LQG_UNROLL for (int q = 0; q < O; ++q) v += opt[Ops::U_OFF + p * O + q] * w[q];
              cv[O + p] = v;
            }
            keepc(t, k, cv);
            R mn[M];
            trial_mean_rows<R, M, ND, FM, false, 1, 0>(none, opt, cv, mn);
            LQG_UNROLL for (int i = 0; i < O; ++i) { dO[k][i] = mn[i]; xprev[k][i] = cv[i]; }
            LQG_UNROLL for (int p = 0; p < RR; ++p) muR[k][p] = cv[O + p] + mn[O + p];   // (the stream holds Fj - I)
          }
        }
      }
    }
    obuf ^= 1;
  }
  if (a.ll) {
    LQG_UNROLL for (int k = 0; k < TPL; ++k)
      if (live[k]) a.ll[sys * a.ll_sb + (n0 + (long)k * BLK) * a.ll_sn] = (R)acc[k];
  }
}

// dense pattern (every constant may be non-zero everywhere): the specialised kernels reduce to the generic ones
template <int NX, int NB, int NU, int NY>
struct DensePattern {
  static constexpr auto Aa = mask_full<NB, NB>();
  static constexpr auto Ba = mask_full<NB, NU>();
  static constexpr auto Fa = mask_full<NY, NB>();
  static constexpr auto VVa = mask_full<NB, NB>();
  static constexpr auto WWa = mask_full<NY, NY>();
  static constexpr auto Q = mask_full<NB, NB>();
  static constexpr auto Rr = mask_full<NU, NU>();
  static constexpr auto Ad = mask_full<NX, NX>();
  static constexpr auto AdmI = mask_full<NX, NX>();
  static constexpr auto Bd = mask_full<NX, NU>();
  static constexpr auto Fd = mask_full<NY, NX>();
  static constexpr auto N1 = mask_full<NX, NX>();
  static constexpr auto WWd = mask_full<NY, NY>();
  static constexpr auto FAa = mask_full<NY, NB>();
  static constexpr auto FAd = mask_full<NY, NX>();
  static constexpr auto DB = mask_full<NY, NU>();
  static constexpr auto N2 = mask_full<NY, NX>();
  static constexpr auto N3 = mask_full<NY, NY>();
};

}  // namespace lqg
