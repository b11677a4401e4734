// lqg_scan.hpp — TIME-PARALLEL system sweeps: the three recursions of the path as associative scans.
//
// The Riccati recursion (lqr.py:16-42), the Kalman covariance recursion (kf.py:6-21) and the moment recursion
// (system.py:209-235) are T dependent steps each: with ONE system (one parameter vector x many trials — the inner loop
// of NUTS / Adam, BASELINE configs 2 and 4) no mapping of a step onto lanes removes that chain (lane-per-system: T x
// ~1000 scalar instructions; workgroup-per-system, lqg_coop.hpp: T x ~6 LDS stages).  All three are linear-fractional
// (Riccati-type) maps, and those compose associatively: an element (A, C, J) of n x n matrices stands for a WINDOW of
// steps, and two adjacent windows combine as (Sarkka & Garcia-Fernandez 2021, Temporal parallelization of Bayesian
// smoothers / of dynamic programming and LQ control)
//     M   = I + C1 J2
//     A   = A2 M^-1 A1
//     C   = A2 M^-1 C1 A2' + C2
//     J   = A1' J2 M^-1 A1 + J1
// so the state after EVERY step follows from a prefix scan: log2(T) dependent combines instead of T dependent steps,
// and the T windows of a level are independent — one wave each, spread over the whole chip.
//   Riccati : elements (A_t, B_t R_t^-1 B_t', Q_t), closed by (0, 0, Qf); the SUFFIX products' J are the costs-to-go S_t.
//   Kalman  : elements ((I - K H) A_t, (I - K H) V V', A_t' H' S^-1 H A_t) with S = H V V' H' + W W', K = V V' H' S^-1,
//             H = F_t; the first element carries the prior; the PREFIX products' C are the filtered covariances P_t|t.
//   moments : the same Kalman construction on the joint (state, belief) system F_j[t], G_j[t] G_j[t]' with the first d
//             components observed exactly (H = [I_d 0], no observation noise); the prefix products' C are the
//             conditional covariances, from which Sigma_t = F_j C F_j' + G_j G_j'.
// Order of the levels: Hillis-Steele (log2 T levels of T combines, ping-pong buffers) while a level fits the chip — the scans of
// small windows are bound by launches and latency, not by work; the windows of 25 .. 64 (one 1024-lane workgroup per combine, 256
// in flight) run the WORK-EFFICIENT Brent-Kung order in place once a level would need more than one round (run_scan,
// lqg_scan_inst.hip): ~2 T combines in 2 log2 T - 2 levels.
// Everything between the scans is independent per step and runs as one wave per (system, step): gains L_t from S_{t+1},
// K_t from P_{t-1|t-1}, the joint system, and finally the per-step trial operators (the same stream k_trial reads).
// Arithmetic is fp64 whatever the problem dtype (there are few systems: the cost is irrelevant, and the fp32 operator
// stream gets fp64-accurate operators); scripts/scan_prototype.py and tests/test_gpu_scan.py pin it: the scans reproduce
// the sequential fp64 recursions to 1e-15 .. 1e-11 on every model of the zoo.
// Preconditions (checked by the caller, lqg_amd/plan.py): no affine cost terms (q, r, P, qf: the likelihood ignores l,
// but P enters G), eigenvalue floor provably inactive (lqr.py:27-28 is not a linear-fractional map when active), u, y,
// d <= 4.  Time-varying specs are fine (elements are per step).
#pragma once
#include <type_traits>
#include <utility>

#include "lqg_coop.hpp"

namespace lqg {
namespace scan {

using D = double;
constexpr int kWave = 64;
constexpr int kStepMax = 1024;       // lanes per element of the per-step kernels: 64 (16 when packed), 1024 for m > 24
constexpr int kStepRk = 512;         // ... of the Riccati / Kalman builders and finalisers for m > 24 (two workgroups per CU)

template <typename F>
LQG_DEV void each(int n, F f) {
  for (int e = threadIdx.x; e < n; e += (int)blockDim.x) f(e);     // blockDim.x = lanes per element (see elem_index)
}
LQG_DEV void wsync() { __syncthreads(); }   // one wave per workgroup: a fence, no barrier instruction

// C[i*ldc + j] = init(i, j) + sum_k A(i,k) B(k,j),  A(i,k) = a[i*ars + k*acs], B(k,j) = b[k*brs + j*bcs]
template <typename Init>
LQG_DEV void mm(D* c, int ldc, int M, int N, int K, const D* a, int ars, int acs, const D* b, int brs, int bcs, Init init) {
  each(M * N, [&](int e) {
    const int i = e / N, j = e - i * N;
    c[i * ldc + j] = coop::dot4<D>(a + i * ars, acs, b + j * bcs, brs, K, init(i, j));
  });
}
LQG_DEV D zero_init(int, int) { return 0.0; }

// symmetric result of a product that is symmetric in exact arithmetic: average of the two mirror entries
template <typename Init>
LQG_DEV void mm_sym(D* c, int n, int K, const D* a, int ars, int acs, const D* b, int brs, int bcs, Init init) {
  each(n * n, [&](int e) {
    const int i = e / n, j = e - i * n;
    const D v1 = coop::dot4<D>(a + i * ars, acs, b + j * bcs, brs, K, init(i, j));
    const D v2 = coop::dot4<D>(a + j * ars, acs, b + i * bcs, brs, K, init(j, i));
    c[i * n + j] = 0.5 * (v1 + v2);
  });
}

// ---------------------------------------------------------------- one level of the scan
// Elements live in global memory as [system][index][A | C | J] (3 n^2 doubles).  Level d of Hillis-Steele:
//   prefix (left = 0): out[k] = in[k-d] (x) in[k]      suffix stored in reversed order (left = 1): out[k] = in[k] (x) in[k-d]
// One launch serves up to two independent scans of the same n (the Riccati and the Kalman scan run side by side).
// k_scan_level_rt also serves the levels of a WORK-EFFICIENT (Brent-Kung) scan run IN PLACE (in == out): a level combines only the
// elements k = k0 + i ks, i < cnt (cnt < 0: every element, the Hillis-Steele level above).
struct Seg {
  const D* in;
  D* out;
  int len, d, left;
  int k0 = 0, ks = 1, cnt = -1;
};
__host__ __device__ inline int seg_count(const Seg& s) { return s.cnt < 0 ? s.len : s.cnt; }

template <int NT, typename F>
LQG_DEV void each_t(int n, F f) {
  for (int e = threadIdx.x; e < n; e += NT) f(e);
}
template <int N>
LQG_DEV D dotn(const D* a, int as, const D* b, int bs, D acc) {
  LQG_UNROLL for (int k = 0; k < N; ++k) acc = fma(a[k * as], b[k * bs], acc);
  return acc;
}

// Everything about the window size is a compile-time constant (N): index arithmetic is multiply-shift, the loops unroll
// and the loads of a dot product are in flight together.  The linear solve M^-1 [A1 | C1] is Gauss-Jordan with partial
// pivoting where EVERY lane finds the pivot itself from N broadcast LDS reads (no cross-lane reduction) and the step
// reads the old matrix from one LDS buffer and writes the new one to another: one fence per column.
// EPB elements share one 256-thread workgroup, NT lanes each (sub-wave for small windows): one element per 64-lane workgroup
// left 48 of 64 lanes idle at n = 4 and made a level of S systems S x T single-wave workgroups — 27 us per level at 32
// systems against 4.7 us at one (the slope of ~10 us per system of scripts/small_batch.py).
template <int N, int NT, int EPB>
__global__ void __launch_bounds__(NT * EPB) k_scan_level(const Seg s0, const Seg s1) {
  constexpr int NN = N * N, W = 3 * N;
  constexpr int LDS_PER = 14 * NN + 8;
  extern __shared__ double lqg_coop_smem[];
  const int sub = (int)threadIdx.x / NT, tid = (int)threadIdx.x - sub * NT;
  D* sm = lqg_coop_smem + sub * LDS_PER;
  int k = (int)blockIdx.x * EPB + sub;
  const bool valid = k < s0.len + s1.len;
  const bool second = k >= s0.len;
  if (second) k -= s0.len;
  const D* in = second ? s1.in : s0.in;
  D* out = second ? s1.out : s0.out;
  const int len = second ? s1.len : s0.len, d = second ? s1.d : s0.d, left = second ? s1.left : s0.left;
  const long sys = blockIdx.y;
  constexpr long es = 3L * NN;
  const bool copy = valid && k < d, comb = valid && k >= d;
  const D* ek = in + (sys * len + (valid ? k : 0)) * es;
  D* eo = out + (sys * len + (valid ? k : 0)) * es;
  auto each = [&](int n, auto f) {
    for (int e = tid; e < n; e += NT) f(e);
  };
  if (copy) each(3 * NN, [&](int e) { eo[e] = ek[e]; });
  const D* ep = in + (sys * len + (comb ? k - d : 0)) * es;
  const D* e1 = left ? ek : ep;        // the window that comes FIRST in time
  const D* e2 = left ? ep : ek;
  D *A1 = sm, *C1 = A1 + NN, *J1 = C1 + NN, *A2 = J1 + NN, *C2 = A2 + NN, *J2 = C2 + NN, *Wa = J2 + NN, *Wb = Wa + 3 * NN,
    *T1 = Wb + 3 * NN, *U = T1 + NN;
  if (comb) each(3 * NN, [&](int e) { A1[e] = e1[e]; A2[e] = e2[e]; });     // (A, C, J are contiguous in both)
  __syncthreads();
  // Wa = [ I + C1 J2 | A1 | C1 ]
  if (comb) each(NN, [&](int e) {
    const int i = e / N, j = e - i * N;
    Wa[i * W + j] = dotn<N>(C1 + i * N, 1, J2 + j, N, (i == j) ? 1.0 : 0.0);
    Wa[i * W + N + j] = A1[e];
    Wa[i * W + 2 * N + j] = C1[e];
  });
  __syncthreads();
  D *src = Wa, *dst = Wb;
  LQG_UNROLL for (int c = 0; c < N; ++c) {
    if (comb) {
      D best = fabs(src[c * W + c]);
      int p = c;
      LQG_UNROLL for (int r = c + 1; r < N; ++r) {
        const D v = fabs(src[r * W + c]);
        if (v > best) { best = v; p = r; }
      }
      const D pinv = 1.0 / src[p * W + c];
      // rows c and p change places; columns <= c are never read again
      constexpr int PER = (N * W + NT - 1) / NT;
      LQG_UNROLL for (int q = 0; q < PER; ++q) {
        const int e = tid + q * NT;
        const int i = e / W, j = e - i * W;
        if (e < N * W && j > c) {
          const D piv = src[p * W + j] * pinv;
          if (i == c) {
            dst[e] = piv;
          } else {
            const int row = (i == p) ? c : i;
            dst[e] = fma(-src[row * W + c], piv, src[row * W + j]);
          }
        }
      }
    }
    __syncthreads();
    D* t = src; src = dst; dst = t;
  }
  const D* X = src;                                                  // [ . | X1 = M^-1 A1 | X2 = M^-1 C1 ]
  // A = A2 X1 ; T1 = A2 X2 ; U = J2 X1
  if (comb) each(3 * NN, [&](int e) {
    const int blk = e / NN, r = e - blk * NN, i = r / N, j = r - i * N;
    if (blk == 0) eo[r] = dotn<N>(A2 + i * N, 1, X + N + j, W, 0.0);
    else if (blk == 1) T1[r] = dotn<N>(A2 + i * N, 1, X + 2 * N + j, W, 0.0);
    else U[r] = dotn<N>(J2 + i * N, 1, X + N + j, W, 0.0);
  });
  __syncthreads();
  // C = T1 A2' + C2 ; J = A1' U + J1 (both symmetric in exact arithmetic: the mirror entries are averaged)
  if (comb) each(2 * NN, [&](int e) {
    const int blk = e / NN, r = e - blk * NN, i = r / N, j = r - i * N;
    if (blk == 0) {
      const D v1 = dotn<N>(T1 + i * N, 1, A2 + j * N, 1, C2[i * N + j]);
      const D v2 = dotn<N>(T1 + j * N, 1, A2 + i * N, 1, C2[j * N + i]);
      eo[NN + r] = 0.5 * (v1 + v2);
    } else {
      const D v1 = dotn<N>(A1 + i, N, U + j, N, J1[i * N + j]);
      const D v2 = dotn<N>(A1 + j, N, U + i, N, J1[j * N + i]);
      eo[2 * NN + r] = 0.5 * (v1 + v2);
    }
  });
}
// lanes per element: enough for ~3 entries of the n x 3n elimination matrix per lane (a 4 x 4 window keeps 16 lanes busy, not
// 64: four of them share a wave); elements per 256-thread workgroup
// (packed = true: many elements per level, throughput matters — 32 systems at n = 4: 27 -> ~9 us per level; packed = false:
// few elements, the latency of one element matters — a full wave per element up to n = 8: 4.7 us against 5.5 us)
constexpr int scan_level_threads(int n, bool packed) {
  return n <= 8 ? (packed ? (n <= 4 ? 16 : n <= 6 ? 32 : 64) : 64) : n <= 12 ? 128 : 256;
}
constexpr int scan_level_epb(int n, bool packed) { return 256 / scan_level_threads(n, packed); }
inline size_t scan_level_lds(int n, bool packed) { return (size_t)(14 * n * n + 8) * sizeof(D) * scan_level_epb(n, packed); }

// ---------------------------------------------------------------- the WHOLE scan in one launch, windows of 1 .. 3
// The dim-1 tracking models (and the decoupled components of the dim-2 ones: lqg_amd/decouple.py) have windows of 2 x 2
// (BoundedActor) or 3 x 3 (SubjectiveActor): one level of k_scan_level is then 4.5 us of launch, LDS staging and fences around
// ~100 multiply-adds — 18 such launches are 40 % of a one-vector evaluation (profiles/r03_w_timeline_config2.txt has the
// n = 4 / 6 picture).  Here ONE LANE owns one window: its element stays in registers through all log2(T) levels, the combine
// is straight-line register code (the same formulas, the same partial pivoting — by compare-and-select, every lane its own
// pivots), and only the partner window travels through LDS ([component][window]: conflict-free).  One workgroup per
// (sequence, system); the sequence (T + 1 <= 1024 windows, 3 n^2 T doubles <= 150 KB) never leaves the CU between levels.
template <int N>
LQG_DEV void lane_combine(const D (&e1)[3 * N * N], const D (&e2)[3 * N * N], D (&eo)[3 * N * N]) {
  constexpr int NN = N * N, W = 3 * N;
  const D *A1 = e1, *C1 = e1 + NN, *J1 = e1 + 2 * NN, *A2 = e2, *C2 = e2 + NN, *J2 = e2 + 2 * NN;
  D Wm[N * W];                                                       // [ I + C1 J2 | A1 | C1 ]
  LQG_UNROLL for (int i = 0; i < N; ++i)
    LQG_UNROLL for (int j = 0; j < N; ++j) {
      D acc = (i == j) ? 1.0 : 0.0;
      LQG_UNROLL for (int k = 0; k < N; ++k) acc = fma(C1[i * N + k], J2[k * N + j], acc);
      Wm[i * W + j] = acc;
      Wm[i * W + N + j] = A1[i * N + j];
      Wm[i * W + 2 * N + j] = C1[i * N + j];
    }
  LQG_UNROLL for (int c = 0; c < N; ++c) {                           // Gauss-Jordan, partial pivoting (lowest row on ties)
    LQG_UNROLL for (int r = c + 1; r < N; ++r) {
      // (bring the larger of rows c, r to row c: after the loop row c holds the column's largest entry, as k_scan_level picks)
      const bool sw = fabs(Wm[r * W + c]) > fabs(Wm[c * W + c]);
      LQG_UNROLL for (int j = c; j < W; ++j) {
        const D a = Wm[c * W + j], b = Wm[r * W + j];
        Wm[c * W + j] = sw ? b : a;
        Wm[r * W + j] = sw ? a : b;
      }
    }
    const D pinv = 1.0 / Wm[c * W + c];
    LQG_UNROLL for (int j = c + 1; j < W; ++j) Wm[c * W + j] *= pinv;
    LQG_UNROLL for (int r = 0; r < N; ++r) {
      if (r == c) continue;
      const D f = Wm[r * W + c];
      LQG_UNROLL for (int j = c + 1; j < W; ++j) Wm[r * W + j] = fma(-f, Wm[c * W + j], Wm[r * W + j]);
    }
  }
  D T1[NN], U[NN];                                                   // X1 = Wm[:, N:2N], X2 = Wm[:, 2N:3N]
  LQG_UNROLL for (int i = 0; i < N; ++i)
    LQG_UNROLL for (int j = 0; j < N; ++j) {
      D a = 0.0, t = 0.0, u = 0.0;
      LQG_UNROLL for (int k = 0; k < N; ++k) {
        a = fma(A2[i * N + k], Wm[k * W + N + j], a);
        t = fma(A2[i * N + k], Wm[k * W + 2 * N + j], t);
        u = fma(J2[i * N + k], Wm[k * W + N + j], u);
      }
      eo[i * N + j] = a;
      T1[i * N + j] = t;
      U[i * N + j] = u;
    }
  LQG_UNROLL for (int i = 0; i < N; ++i)
    LQG_UNROLL for (int j = i; j < N; ++j) {
      D c1 = C2[i * N + j], c2 = C2[j * N + i], j1 = J1[i * N + j], j2 = J1[j * N + i];
      LQG_UNROLL for (int k = 0; k < N; ++k) {
        c1 = fma(T1[i * N + k], A2[j * N + k], c1);
        c2 = fma(T1[j * N + k], A2[i * N + k], c2);
        j1 = fma(A1[k * N + i], U[k * N + j], j1);
        j2 = fma(A1[k * N + j], U[k * N + i], j2);
      }
      eo[NN + i * N + j] = eo[NN + j * N + i] = 0.5 * (c1 + c2);
      eo[2 * NN + i * N + j] = eo[2 * NN + j * N + i] = 0.5 * (j1 + j2);
    }
}

// grid (sequences, systems), block >= the longest sequence.  Sequence i: `len` windows from s.in, result into s.out.
template <int N, int MAXT>
__global__ void __launch_bounds__(MAXT) k_scan_lane(const Seg s0, const Seg s1) {
  constexpr int ES = 3 * N * N;
  extern __shared__ double lqg_coop_smem[];
  const bool second = blockIdx.x == 1;
  const D* in = second ? s1.in : s0.in;
  D* out = second ? s1.out : s0.out;
  const int len = second ? s1.len : s0.len, left = second ? s1.left : s0.left;
  const int k = (int)threadIdx.x;
  const bool live = k < len;
  const long base = ((long)blockIdx.y * len + (live ? k : 0)) * ES;
  D own[ES];
  LQG_UNROLL for (int e = 0; e < ES; ++e) own[e] = in[base + e];
  for (int d = 1; d < len; d *= 2) {
    if (live) {
      LQG_UNROLL for (int e = 0; e < ES; ++e) lqg_coop_smem[e * len + k] = own[e];
    }
    __syncthreads();
    if (live && k >= d) {
      D other[ES], res[ES];
      LQG_UNROLL for (int e = 0; e < ES; ++e) other[e] = lqg_coop_smem[e * len + k - d];
      if (left) lane_combine<N>(own, other, res);                    // (the window that comes FIRST in time is the left operand)
      else lane_combine<N>(other, own, res);
      LQG_UNROLL for (int e = 0; e < ES; ++e) own[e] = res[e];
    }
    __syncthreads();                                                 // every partner read before the next level's writes
  }
  if (live) {
    LQG_UNROLL for (int e = 0; e < ES; ++e) out[base + e] = own[e];
  }
}
constexpr int kScanLaneMaxN = 3;
// (3 x 3 windows need ~150 registers per lane: at most 512 lanes per workgroup, i.e. T <= 511)
inline int scan_lane_max_len(int n) { return n <= 2 ? 1024 : 512; }
inline size_t scan_lane_lds(int n, int len) { return (size_t)3 * n * n * len * sizeof(D); }

// ---------------------------------------------------------------- one level of the scan, windows of 25 .. 64
// The delay-augmented models (lqg/tracking/delay.py:9-51: b = 39, m - d = 64 for the reference's DelayedSubjectiveActor)
// have windows whose 14 n^2 doubles do not fit LDS.  Same combine, same pivoting rule, other data placement — run-time n,
// ONE element per workgroup of NW waves:
//   * the n x 3n elimination matrix [I + C1 J2 | A1 | C1] lives in REGISTERS: wave w owns rows w TI .. w TI + TI - 1
//     (TI = 64 / NW), lane l the columns l, n + l, 2n + l of them — an elimination step costs each lane 3 TI multiply-adds
//     and TI + 6 LDS reads (the pivot row, the row it displaces and the pivot column are published through LDS; every
//     wave finds the pivot itself by a wave reduction of the published column: two barriers per column);
//   * the six n x n x n products run on the fp64 matrix core: 16 x 16 output tiles dealt to the waves, operands straight from
//     the elements in global memory (L2) or from LDS panels (mfma_tile);
//   * LDS holds four n x (n + 1) panels: M, then X1 = M^-1 A1 | X2 = M^-1 C1 | T1 = A2 X2 | U = J2 X1, then the unsymmetrised
//     C and J whose mirror entries are averaged through LDS (an odd leading dimension keeps the strided reads conflict-free).
// 137 KB of LDS at n = 64, 54 KB at n = 39; ceil(n / TI) waves are launched.
#ifdef LQG_SCAN_STAMP
// developer build (-DLQG_SCAN_STAMP, variant library): cycles per phase of k_scan_level_rt, accumulated by lane 0 of the LAST
// window of system 0 (read back with lqg_debug_scan_stamps of lqg_scan_inst.hip)
__device__ unsigned long long g_scan_stamps[16];
#define LQG_SSTAMP(slot_)                                                        \
  do {                                                                           \
    if (threadIdx.x == 0 && blockIdx.x == gridDim.x - 1 && blockIdx.y == 0 && n > 48) { \
      const unsigned long long now_ = __builtin_readcyclecounter();              \
      g_scan_stamps[slot_] += now_ - stamp_prev_;                                \
      stamp_prev_ = now_;                                                        \
    }                                                                            \
  } while (0)
#else
#define LQG_SSTAMP(slot_) do { } while (0)
#endif
// maximum of a 32-bit key over the wave by DPP (row rotations, then the two row broadcasts of gfx9): six v_max_u32 with a
// DPP operand where six rounds of __shfl_xor on (double, int) cost 18 dependent ds_bpermute round trips (1850 cycles per
// column, measured).  Every lane of row 3 ends with the maximum; lane 63 is read.
template <int CTRL, int ROW_MASK>
LQG_DEV unsigned dpp_max_step(unsigned v) {
  const unsigned o = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROW_MASK, 0xF, false);
  return o > v ? o : v;
}
LQG_DEV unsigned wave_max_u32(unsigned v) {
  v = dpp_max_step<0x121, 0xF>(v);      // row_ror:1
  v = dpp_max_step<0x122, 0xF>(v);      // row_ror:2
  v = dpp_max_step<0x124, 0xF>(v);      // row_ror:4
  v = dpp_max_step<0x128, 0xF>(v);      // row_ror:8   -> every lane holds its row's maximum
  v = dpp_max_step<0x142, 0xA>(v);      // row_bcast:15 into rows 1, 3
  v = dpp_max_step<0x143, 0xC>(v);      // row_bcast:31 into rows 2, 3
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
// 1 / x to fp64 rounding: v_rcp_f64 and two Newton steps (the IEEE division sequence costs ~40 instructions per column on
// every wave; the quotient only scales the pivot row)
LQG_DEV D fast_rcp(D x) {
  D r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}

// ---- 16 x 16 output tiles of the window products on the fp64 matrix core (v_mfma_f64_16x16x4_f64) ------------------------
// Register layout, pinned on the hardware by scripts/micro/mfma_f64_layout.hip: A operand — lane l holds A(l % 16, l / 16);
// B operand — lane l holds B(l / 16, l % 16); accumulator — lane l, register r holds D(4 r + l / 16, l % 16).
// The reduction index is split so that lane group g = l / 16 owns the CONTIGUOUS quarter g Q .. g Q + Q - 1 of it (Q = padded
// n / 4): a left operand stored by rows is then Q consecutive doubles per lane.  Operand loaders return 0 outside the n x n
// matrix (the padding of the last tiles), so no accumulated term is ever NaN-poisoned by what lies behind an operand.
typedef double mfma_acc_t __attribute__((ext_vector_type(4)));
struct TileIdx { int i0, j0, li, g; };
template <int Q, typename LF, typename RF>
LQG_DEV mfma_acc_t mfma_tile(const TileIdx& t, LF left, RF right) {
  mfma_acc_t acc = {0.0, 0.0, 0.0, 0.0};
  D a[Q], b[Q];
  LQG_UNROLL for (int q = 0; q < Q; ++q) { a[q] = left(t.i0 + t.li, t.g * Q + q); b[q] = right(t.g * Q + q, t.j0 + t.li); }
  LQG_UNROLL for (int q = 0; q < Q; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], b[q], acc, 0, 0, 0);
  return acc;
}
// two products sharing their left operand
template <int Q, typename LF, typename RF1, typename RF2>
LQG_DEV void mfma_tile2(const TileIdx& t, LF left, RF1 right1, RF2 right2, mfma_acc_t& acc1, mfma_acc_t& acc2) {
  acc1 = mfma_acc_t{0.0, 0.0, 0.0, 0.0};
  acc2 = mfma_acc_t{0.0, 0.0, 0.0, 0.0};
  D a[Q], b1[Q], b2[Q];
  LQG_UNROLL for (int q = 0; q < Q; ++q) {
    a[q] = left(t.i0 + t.li, t.g * Q + q);
    b1[q] = right1(t.g * Q + q, t.j0 + t.li);
    b2[q] = right2(t.g * Q + q, t.j0 + t.li);
  }
  LQG_UNROLL for (int q = 0; q < Q; ++q) {
    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], b1[q], acc1, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], b2[q], acc2, 0, 0, 0);
  }
}
// M(r, c) = p[r * rs + c * cs] inside the n x n matrix, 0 outside (clamped address, masked value)
struct MatView {
  const D* p;
  int rs, cs, n;
  LQG_DEV D operator()(int r, int c) const {
    const bool in = r < n && c < n;
    const D v = p[(in ? r : 0) * rs + (in ? c : 0) * cs];
    return in ? v : 0.0;
  }
};

// (second launch bound = waves per SIMD.  Two workgroups per CU — 64 VGPRs at 16 waves — spill and run 9 % slower: 1)
#ifndef LQG_SCAN_RT_WGS_PER_CU
#define LQG_SCAN_RT_WGS_PER_CU 1
#endif
template <int NW>
__global__ void __launch_bounds__(NW * 64, LQG_SCAN_RT_WGS_PER_CU * NW / 4) k_scan_level_rt(const Seg s0, const Seg s1, const int n) {
  constexpr int TI = 64 / NW;
  extern __shared__ double lqg_coop_smem[];
  const int tid = (int)threadIdx.x, lane = tid & 63, NT = (int)blockDim.x;     // (ceil(n / TI) waves: scan_level_rt_threads)
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nn = n * n, ld = n | 1, panel = n * (n + 1);              // (odd leading dimension: transposed reads conflict-free)
  int k = (int)blockIdx.x;
  const int c0 = seg_count(s0);
  const bool second = k >= c0;
  if (second) k -= c0;
  const D* in = second ? s1.in : s0.in;
  D* out = second ? s1.out : s0.out;
  const int len = second ? s1.len : s0.len, d = second ? s1.d : s0.d, left = second ? s1.left : s0.left;
  k = (second ? s1.k0 : s0.k0) + k * (second ? s1.ks : s0.ks);        // (the element this workgroup produces)
  const long sys = blockIdx.y;
  const long es = 3L * nn;
  const D* ek = in + (sys * len + k) * es;
  D* eo = out + (sys * len + k) * es;
  if (k < d) {                                                        // (workgroup-uniform: no barrier is skipped by a part of it)
    if (eo != ek)
      for (int e = tid; e < 3 * nn; e += NT) eo[e] = ek[e];
    return;
  }
  const D* ep = in + (sys * len + (k - d)) * es;
  const D* __restrict__ e1 = left ? ek : ep;                          // the window that comes FIRST in time
  const D* __restrict__ e2 = left ? ep : ek;
  const D *A1 = e1, *C1 = e1 + nn, *J1 = e1 + 2 * nn, *A2 = e2, *C2 = e2 + nn, *J2 = e2 + 2 * nn;
  D *P0 = lqg_coop_smem, *P1 = P0 + panel, *P2 = P1 + panel, *P3 = P2 + panel;
  D *pcol = P3 + panel, *prow = pcol + 128, *crow = prow + 192;
  const bool lv = lane < n;
  const int lc = lv ? lane : n - 1;                                   // clamped lane for reads whose result is masked
  const int nwv = NT >> 6, nt = (n + 15) >> 4;                        // waves launched, tiles per matrix side
  const MatView mA1{A1, n, 1, n}, mC1{C1, n, 1, n}, mA2{A2, n, 1, n}, mJ2{J2, n, 1, n};
  const MatView mA1t{A1, 1, n, n}, mA2t{A2, 1, n, n};                 // transposed views
  // a product's tiles on this workgroup's waves: f(tile index struct) for tile = w, w + nwv, ...
  auto tiles = [&](auto f) {
    for (int tile = w; tile < nt * nt; tile += nwv) {
      const int ti = tile / nt;
      f(TileIdx{ti * 16, (tile - ti * nt) * 16, lane & 15, lane >> 4});
    }
  };
  // a wave owns at most KEEP tiles of a product — nwv = ceil(n / TI) waves over ceil(n / 16)^2 tiles: 2 at TI = 4, 3 at TI = 8
  // (n = 49 .. 56)
  constexpr int KEEP = NW >= 16 ? 2 : 3;
  // D(4 r + g, li) of a tile -> dst(row, col)
  auto scatter = [&](const TileIdx& t, const mfma_acc_t& acc, auto put) {
    LQG_UNROLL for (int r = 0; r < 4; ++r) {
      const int i = t.i0 + 4 * r + t.g, j = t.j0 + t.li;
      if (i < n && j < n) put(i, j, acc[r]);
    }
  };
  auto with_q = [&](auto f) {                                         // padded n / 4 as a compile-time constant
    if (nt == 4) f(std::integral_constant<int, 16>{});
    else if (nt == 3) f(std::integral_constant<int, 12>{});
    else f(std::integral_constant<int, 8>{});
  };
#ifdef LQG_SCAN_STAMP
  unsigned long long stamp_prev_ = __builtin_readcyclecounter();
#endif
  LQG_SSTAMP(0);
  // ---- M = I + C1 J2 -> P0; rows of [ M | A1 | C1 ] into registers
  with_q([&](auto qc) {
    constexpr int Q = decltype(qc)::value;
    tiles([&](const TileIdx& t) {
      const mfma_acc_t acc = mfma_tile<Q>(t, mC1, mJ2);
      scatter(t, acc, [&](int i, int j, D v) { P0[i * n + j] = v + ((i == j) ? 1.0 : 0.0); });
    });
  });
  __syncthreads();
  D own[TI][3];
  LQG_UNROLL for (int r = 0; r < TI; ++r) {
    const int i = w * TI + r;
    const bool live = lv && i < n;
    const int at = (i < n ? i : n - 1) * n + lc;
    own[r][0] = live ? P0[at] : 0.0;
    own[r][1] = live ? A1[at] : 0.0;
    own[r][2] = live ? C1[at] : 0.0;
  }
  LQG_SSTAMP(1);
  // ---- Gauss-Jordan with partial pivoting.  The pivot is the largest |entry| of the column compared on the sign-less
  // high word of the double with its low six bits replaced by the row (exponent + 14 mantissa bits decide, lowest row on
  // ties): a pivot within 2^-14 of the largest — the growth bound of partial pivoting is unchanged to that factor.
  for (int c = 0; c < n; ++c) {
    D* pc = pcol + (c & 1) * 64;
    if (lane == c) {
      LQG_UNROLL for (int r = 0; r < TI; ++r) pc[w * TI + r] = own[r][0];
    }
    __syncthreads();
    LQG_SSTAMP(2);
    const D mine = pc[lane];                                          // (lanes >= n read the zero rows' entries)
    D ci[TI];                                                         // this wave's column entries
    LQG_UNROLL for (int r = 0; r < TI; ++r) ci[r] = pc[w * TI + r];
    const unsigned hi = (unsigned)__double2hiint(mine) & 0x7fffffffu;
    const unsigned key = (lane >= c && lv) ? ((hi & ~63u) | (unsigned)(63 - lane)) : 0u;
    const int p = 63 - (int)(wave_max_u32(key) & 63u);
    LQG_SSTAMP(3);
    // the pivot and the entry of row c straight from the lanes that hold them (no second LDS round trip)
    const D pv = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(mine), p), __builtin_amdgcn_readlane(__double2loint(mine), p));
    const D colc = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(mine), c), __builtin_amdgcn_readlane(__double2loint(mine), c));
    const D pinv = fast_rcp(pv);
    const int wp = p / TI, wc = c / TI;
    if (w == wp) {
      LQG_UNROLL for (int r = 0; r < TI; ++r)
        if (r == p % TI) { LQG_UNROLL for (int q = 0; q < 3; ++q) prow[q * 64 + lane] = own[r][q]; }
    }
    if (w == wc) {
      LQG_UNROLL for (int r = 0; r < TI; ++r)
        if (r == c % TI) { LQG_UNROLL for (int q = 0; q < 3; ++q) crow[q * 64 + lane] = own[r][q]; }
    }
    __syncthreads();
    LQG_SSTAMP(4);
    D pr[3];
    LQG_UNROLL for (int q = 0; q < 3; ++q) pr[q] = prow[q * 64 + lane] * pinv;
    if (w != wp && w != wc) {                                         // (most waves: neither the pivot row nor row c)
      LQG_UNROLL for (int r = 0; r < TI; ++r)
        LQG_UNROLL for (int q = 0; q < 3; ++q) own[r][q] = fma(-ci[r], pr[q], own[r][q]);
    } else {
      D cr[3];
      LQG_UNROLL for (int q = 0; q < 3; ++q) cr[q] = crow[q * 64 + lane];
      LQG_UNROLL for (int r = 0; r < TI; ++r) {
        const int i = w * TI + r;
        const bool isp = i == p, isc = i == c;                        // (rows c and p change places)
        const D coef = isp ? colc : ci[r];
        LQG_UNROLL for (int q = 0; q < 3; ++q) {
          const D v = fma(-coef, pr[q], isp ? cr[q] : own[r][q]);
          own[r][q] = isc ? pr[q] : v;
        }
      }
    }
    LQG_SSTAMP(5);
  }
  // ---- X1 = M^-1 A1 -> P0, X2 = M^-1 C1 -> P1   (every wave passed the elimination's barriers after reading M from P0)
  LQG_UNROLL for (int r = 0; r < TI; ++r) {
    const int i = w * TI + r;
    if (i < n && lv) { P0[i * n + lane] = own[r][1]; P1[i * n + lane] = own[r][2]; }
  }
  __syncthreads();
  LQG_SSTAMP(6);
  // ---- A = A2 X1 (out), T1 = A2 X2 -> P2 (rows padded to ld), U = J2 X1 -> P3
  // (A stays in registers until every read of the operands is behind a barrier: the level may run in place, out == in)
  mfma_acc_t keepA[KEEP];
  with_q([&](auto qc) {
    constexpr int Q = decltype(qc)::value;
    const MatView mX1{P0, n, 1, n}, mX2{P1, n, 1, n};
    LQG_UNROLL for (int it = 0; it < KEEP; ++it) {
      const int tile = w + it * nwv;
      if (tile < nt * nt) {
        const int ti = tile / nt;
        const TileIdx t{ti * 16, (tile - ti * nt) * 16, lane & 15, lane >> 4};
        mfma_acc_t aT;
        mfma_tile2<Q>(t, mA2, mX1, mX2, keepA[it], aT);
        const mfma_acc_t aU = mfma_tile<Q>(t, mJ2, mX1);
        scatter(t, aT, [&](int i, int j, D v) { P2[i * ld + j] = v; });
        scatter(t, aU, [&](int i, int j, D v) { P3[i * n + j] = v; });
      }
    }
  });
  __syncthreads();                                                    // (X1, X2 dead)
  LQG_SSTAMP(7);
  // ---- unsymmetrised C = T1 A2' -> P0, J = A1' U -> P1 (rows padded to ld)
  with_q([&](auto qc) {
    constexpr int Q = decltype(qc)::value;
    const MatView mT1{P2, ld, 1, n}, mU{P3, n, 1, n};
    tiles([&](const TileIdx& t) {
      const mfma_acc_t gC = mfma_tile<Q>(t, mT1, mA2t);
      const mfma_acc_t gJ = mfma_tile<Q>(t, mA1t, mU);
      scatter(t, gC, [&](int i, int j, D v) { P0[i * ld + j] = v; });
      scatter(t, gJ, [&](int i, int j, D v) { P1[i * ld + j] = v; });
    });
  });
  __syncthreads();
  LQG_SSTAMP(8);
  // ---- the mirror entries of C and J are averaged (C2, J1 are exactly symmetric: they were made so).  In place, C2 or J1 IS the
  // entry being overwritten: read and written by the same lane.  A: no operand is read after the barrier above.
  LQG_UNROLL for (int r = 0; r < TI; ++r) {
    const int a = w * TI + r;
    if (a < n && lv) {
      const D c2 = C2[a * n + lane], j1 = J1[a * n + lane];
      eo[nn + a * n + lane] = 0.5 * (P0[a * ld + lane] + P0[lane * ld + a]) + c2;
      eo[2 * nn + a * n + lane] = 0.5 * (P1[a * ld + lane] + P1[lane * ld + a]) + j1;
    }
  }
  LQG_UNROLL for (int it = 0; it < KEEP; ++it) {
    const int tile = w + it * nwv;
    if (tile < nt * nt) {
      const int ti = tile / nt;
      scatter(TileIdx{ti * 16, (tile - ti * nt) * 16, lane & 15, lane >> 4}, keepA[it], [&](int i, int j, D v) { eo[i * n + j] = v; });
    }
  }
  LQG_SSTAMP(9);
}
constexpr int kScanRtMax = 64;            // largest window of k_scan_level_rt (one lane per column)
constexpr long kScanRtConcurrent = 256;   // combines of k_scan_level_rt in flight on the chip: one workgroup per CU
inline size_t scan_level_rt_lds(int n) { return (size_t)(4 * n * (n + 1) + 128 + 192 + 192 + 64) * sizeof(D); }
inline int scan_level_rt_threads(int n, int nw) { const int ti = 64 / nw; return (n + ti - 1) / ti * 64; }

// ---------------------------------------------------------------- per-step kernels
template <typename R>
struct Args {
  DView<R> aQ, aQf, aR, aA, aB, aF, aV, aW;
  DView<R> dA, dB, dF, dV, dW;
  DView<R> Sigma0;
  DView<R> Sig;            // optional output Sigma[B,T,m,m]
  D* elems;                // scan input buffer [n_sys][len][3 n^2]
  const D* res;            // scan result buffer
  D* elems2;               // the Kalman scan's buffers (it runs side by side with the Riccati scan)
  const D* res2;
  D* Lbuf;                 // [n_sys][T][u*b]
  D* Kbuf;                 // [n_sys][T][b*y]
  D* FG;                   // [n_sys][T][2 m^2]   Fj | GG
  R* ops;                  // [n_sys][T+1][nops]  (problem dtype)
  long n_sys;
  int T, x, b, u, y, d, nva, nwa, nvd, nwd, nops;
  D eps;
  int lds_elem;            // doubles of LDS per element of the launch (set per launch by the host)
};

template <typename R>
LQG_DEV void ld(const DView<R>& v, long s, int t, int rows, int cols, D* dst) {
  const R* p = v.p + s * v.sb + (long)t * v.st;
  each(rows * cols, [&](int e) { const int i = e / cols, j = e - i * cols; dst[e] = (D)p[i * v.sr + j * v.sc]; });
}
template <typename R>
LQG_DEV void ld_sym(const DView<R>& v, long s, int t, int n, D* dst) {
  const R* p = v.p + s * v.sb + (long)t * v.st;
  each(n * n, [&](int e) {
    const int i = e / n, j = e - i * n;
    dst[e] = (i == j) ? (D)p[i * v.sr + i * v.sc] : 0.5 * ((D)p[i * v.sr + j * v.sc] + (D)p[j * v.sr + i * v.sc]);
  });
}
template <typename R>
LQG_DEV void ld_gram(const DView<R>& v, long s, int t, int n, int nv, D* dst) {
  const R* p = v.p + s * v.sb + (long)t * v.st;
  each(n * n, [&](int e) {
    const int i = e / n, j = e - i * n;
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    D acc = 0.0;
    for (int k = 0; k < nv; ++k) acc += (D)p[lo * v.sr + k * v.sc] * (D)p[hi * v.sr + k * v.sc];
    dst[e] = acc;
  });
}
// inverse of a small SPD matrix (n <= 4) held in LDS -> LDS (lane 0 writes; every lane computes)
LQG_DEV void small_inverse(const D* src, int n, D* dst, D eps, bool floor_) {
  coop::dispatch_small<0>(n, [&](auto nc) {
    constexpr int N = decltype(nc)::value;
    D Hi[N * N], Ht[N * N];
    coop::spd_inverse_reg<D, N>(src, eps, floor_, Hi, Ht);
    if (threadIdx.x == 0) {
      LQG_UNROLL for (int e = 0; e < N * N; ++e) dst[e] = Hi[e];
    }
  });
  wsync();
}

// The per-step kernels run one ELEMENT (system, step) on blockDim.x lanes and blockDim.y elements per 64-thread workgroup:
// (64, 1) when there are few elements (latency matters), (16, 4) when a launch holds thousands of small ones (the 4 x 4
// matrices of a tracking model keep 16 lanes busy, not 64).  A workgroup is always ONE wave, so the fences inside
// element-dependent branches are harmless.  `lds_elem`: doubles of LDS per element.
LQG_DEV int elem_index() { return (int)(blockIdx.x * blockDim.y + threadIdx.y); }
LQG_DEV D* elem_lds(double* base, int lds_elem) { return base + (int)threadIdx.y * lds_elem; }

// Riccati elements, reversed index j: j = 0 terminal (0, 0, Qf); j >= 1 <-> step t = T - j: (A_t, B R^-1 B', Q_t)
template <typename R>
LQG_DEV void build_riccati(const Args<R>& a, D* sm, int j, long s) {
  const int b = a.b, u = a.u, nn = b * b;
  D* e = a.elems + (s * (a.T + 1) + j) * 3L * nn;
  if (j == 0) {
    each(2 * nn, [&](int i) { e[i] = 0.0; });
    ld_sym(a.aQf, s, 0, b, e + 2 * nn);
    return;
  }
  const int t = a.T - j;
  D *Bm = sm, *Rm = Bm + b * u, *Ri = Rm + u * u, *BR = Ri + u * u;
  ld(a.aB, s, t, b, u, Bm);
  ld_sym(a.aR, s, t, u, Rm);
  ld(a.aA, s, t, b, b, e);
  ld_sym(a.aQ, s, t, b, e + 2 * nn);
  wsync();
  small_inverse(Rm, u, Ri, 0.0, false);
  mm(BR, u, b, u, u, Bm, u, 1, Ri, u, 1, zero_init);                 // B R^-1
  wsync();
  mm_sym(e + nn, b, u, BR, u, 1, Bm, 1, u, zero_init);               // (B R^-1) B'
}

// L_t = -Ht^-1 G from S_{t+1} (J of the suffix product at reversed index T - 1 - t)          lqr.py:22-31
template <typename R>
LQG_DEV void gains_step(const Args<R>& a, D* sm, int t, long s) {
  const int b = a.b, u = a.u, nn = b * b;
  const D* S = a.res + (s * (a.T + 1) + (a.T - 1 - t)) * 3L * nn + 2 * nn;
  D *A = sm, *Bm = A + nn, *Rm = Bm + b * u, *SA = Rm + u * u, *SB = SA + nn, *H = SB + b * u, *G = H + u * u, *Hi = G + u * b;
  ld(a.aA, s, t, b, b, A);
  ld(a.aB, s, t, b, u, Bm);
  ld_sym(a.aR, s, t, u, Rm);
  wsync();
  if (a.x + b > 24) {
    // large windows (kStepRk lanes per element): the delay augmentations' A is a shift — the non-zero ROWS of each of its columns
    // are listed (coop::RowLists on the transposed view, behind this function's working set) and S A walks them; exact, the terms
    // left out are exact zeros
    unsigned char* lp = reinterpret_cast<unsigned char*>(Hi + u * u);
    const coop::RowLists cl = coop::take_lists(lp, b, b);
    coop::build_lists<kStepRk>((int)threadIdx.x, A, 1, b, b, b, cl);
    wsync();
    each(nn, [&](int e) {
      const int i = e / b, j = e - i * b;
      SA[e] = coop::dot_list<D>(cl, j, A + j, b, S + i * b, 1, 0.0);
    });
  } else {
    mm(SA, b, b, b, b, S, b, 1, A, b, 1, zero_init);
  }
  mm(SB, u, b, u, b, S, b, 1, Bm, u, 1, zero_init);
  wsync();
  mm_sym(H, u, b, Bm, 1, u, SB, u, 1, [&](int i, int j) { return Rm[i * u + j]; });     // H = R + B'SB
  mm(G, b, u, b, b, Bm, 1, u, SA, b, 1, zero_init);                                      // G = B'SA
  wsync();
  small_inverse(H, u, Hi, a.eps, true);                                                  // (H + floor)^-1
  D* L = a.Lbuf + (s * a.T + t) * (long)(u * b);
  mm(L, b, u, b, u, Hi, u, 1, G, b, 1, zero_init);
  wsync();
  each(u * b, [&](int e) { L[e] = -L[e]; });
}

// Kalman elements, index t = 0 .. T-1 (the filtered covariance after step t)                  kf.py:10-14
template <typename R>
LQG_DEV void build_kalman(const Args<R>& a, D* sm, int t, long s) {
  const int b = a.b, y = a.y, nn = b * b;
  D* e = a.elems2 + (s * a.T + t) * 3L * nn;
  D *A = sm, *F = A + nn, *VV = F + y * b, *WW = VV + nn, *Pp = WW + y * y, *FP = Pp + nn, *Sm = FP + y * b, *Si = Sm + y * y,
    *K = Si + y * y, *T1 = K + b * y, *IKF = T1 + nn;
  ld(a.aA, s, t, b, b, A);
  ld(a.aF, s, t, y, b, F);
  ld_gram(a.aV, s, t, b, a.nva, VV);
  ld_gram(a.aW, s, t, y, a.nwa, WW);
  if (t == 0) {                                   // prior P0 -> predicted Pp = A P0 A' + V V'
    if (a.Sigma0.p) ld_sym(a.Sigma0, s, 0, b, T1);
    else ld_gram(a.aV, s, 0, b, a.nva, T1);
    wsync();
    mm(IKF, b, b, b, b, A, b, 1, T1, b, 1, zero_init);
    wsync();
    mm_sym(Pp, b, b, IKF, b, 1, A, 1, b, [&](int i, int j) { return VV[i * b + j]; });
  } else {
    wsync();
    each(nn, [&](int i) { Pp[i] = VV[i]; });      // one step from a point: "prior" covariance = process noise
  }
  wsync();
  mm(FP, b, y, b, b, F, b, 1, Pp, b, 1, zero_init);                                       // F Pp
  wsync();
  mm_sym(Sm, y, b, FP, b, 1, F, 1, b, [&](int i, int j) { return WW[i * y + j]; });       // S = F Pp F' + W W'
  wsync();
  small_inverse(Sm, y, Si, 0.0, false);
  mm(K, y, b, y, y, FP, 1, b, Si, y, 1, zero_init);                                        // K = (F Pp)' S^-1
  wsync();
  each(nn, [&](int i2) {                                                                   // I - K F
    const int i = i2 / b, j = i2 - i * b;
    IKF[i2] = coop::dot4<D>(K + i * y, 1, F + j, b, y, 0.0);
    IKF[i2] = ((i == j) ? 1.0 : 0.0) - IKF[i2];
  });
  wsync();
  const bool listed = a.x + b > 24;               // large windows: I - K F is the identity plus the few observed columns
  coop::RowLists rl{nullptr, nullptr, 0};
  if (listed) {
    unsigned char* lp = reinterpret_cast<unsigned char*>(IKF + nn);
    rl = coop::take_lists(lp, b, b);
    coop::build_lists<kStepRk>((int)threadIdx.x, IKF, b, 1, b, b, rl);
    wsync();
    each(nn, [&](int i2) {                                                                 // C = (I - K F) Pp, mirror entries averaged
      const int i = i2 / b, j = i2 - i * b;
      const D v1 = coop::dot_list<D>(rl, i, IKF + i * b, 1, Pp + j, b, 0.0);
      const D v2 = coop::dot_list<D>(rl, j, IKF + j * b, 1, Pp + i, b, 0.0);
      e[nn + i2] = 0.5 * (v1 + v2);
    });
  } else {
    mm_sym(e + nn, b, b, IKF, b, 1, Pp, b, 1, zero_init);                                  // C = (I - K F) Pp
  }
  if (t == 0) {
    each(nn, [&](int i) { e[i] = 0.0; e[2 * nn + i] = 0.0; });
  } else {
    if (listed) {
      each(nn, [&](int i2) {                                                               // A = (I - K F) A_t
        const int i = i2 / b, j = i2 - i * b;
        e[i2] = coop::dot_list<D>(rl, i, IKF + i * b, 1, A + j, b, 0.0);
      });
    } else {
      mm(e, b, b, b, b, IKF, b, 1, A, b, 1, zero_init);                                    // A = (I - K F) A_t
    }
    mm(T1, b, y, b, y, Si, y, 1, F, b, 1, zero_init);                                      // S^-1 F            [y, b]
    mm(FP, b, y, b, b, F, b, 1, A, b, 1, zero_init);                                       // F A_t             [y, b] (FP reused)
    wsync();
    mm(Pp, b, y, b, b, T1, b, 1, A, b, 1, zero_init);                                      // S^-1 F A_t        [y, b] (Pp reused)
    wsync();
    mm_sym(e + 2 * nn, b, y, FP, 1, b, Pp, b, 1, zero_init);                               // J = (F A)' S^-1 (F A)
  }
}

// K_t from the filtered covariance of the previous step                                      kf.py:10-12
template <typename R>
LQG_DEV void kgain_step(const Args<R>& a, D* sm, int t, long s) {
  const int b = a.b, y = a.y, nn = b * b;
  D *A = sm, *F = A + nn, *VV = F + y * b, *WW = VV + nn, *P0 = WW + y * y, *AP = P0 + nn, *Pp = AP + nn, *FP = Pp + nn,
    *Sm = FP + y * b, *Si = Sm + y * y;
  ld(a.aA, s, t, b, b, A);
  ld(a.aF, s, t, y, b, F);
  ld_gram(a.aV, s, t, b, a.nva, VV);
  ld_gram(a.aW, s, t, y, a.nwa, WW);
  if (t == 0) {
    if (a.Sigma0.p) ld_sym(a.Sigma0, s, 0, b, P0);
    else ld_gram(a.aV, s, 0, b, a.nva, P0);
  } else {
    const D* C = a.res2 + (s * a.T + t - 1) * 3L * nn + nn;
    each(nn, [&](int i) { P0[i] = C[i]; });
  }
  wsync();
  if (a.x + b > 24) {                             // (large windows: the rows of the shift-structured A listed, as in gains_step)
    unsigned char* lp = reinterpret_cast<unsigned char*>(Si + y * y);
    const coop::RowLists rl = coop::take_lists(lp, b, b);
    coop::build_lists<kStepRk>((int)threadIdx.x, A, b, 1, b, b, rl);
    wsync();
    each(nn, [&](int e) {
      const int i = e / b, j = e - i * b;
      AP[e] = coop::dot_list<D>(rl, i, A + i * b, 1, P0 + j, b, 0.0);
    });
    wsync();
    each(nn, [&](int e) {                         // A P A' + V V', mirror entries averaged
      const int i = e / b, j = e - i * b;
      const D v1 = coop::dot_list<D>(rl, j, A + j * b, 1, AP + i * b, 1, VV[i * b + j]);
      const D v2 = coop::dot_list<D>(rl, i, A + i * b, 1, AP + j * b, 1, VV[j * b + i]);
      Pp[e] = 0.5 * (v1 + v2);
    });
  } else {
    mm(AP, b, b, b, b, A, b, 1, P0, b, 1, zero_init);
    wsync();
    mm_sym(Pp, b, b, AP, b, 1, A, 1, b, [&](int i, int j) { return VV[i * b + j]; });
  }
  wsync();
  mm(FP, b, y, b, b, F, b, 1, Pp, b, 1, zero_init);
  wsync();
  mm_sym(Sm, y, b, FP, b, 1, F, 1, b, [&](int i, int j) { return WW[i * y + j]; });
  wsync();
  small_inverse(Sm, y, Si, 0.0, false);
  D* K = a.Kbuf + (s * a.T + t) * (long)(b * y);
  mm(K, y, b, y, y, FP, 1, b, Si, y, 1, zero_init);
}

// The Riccati and the Kalman recursion do not depend on each other: their elements are built by one launch (blocks
// 0 .. T: Riccati, T+1 .. 2T: Kalman), their scans advance in the same launches (k_scan_level's two segments), and one
// launch turns the results into the gains L_t (blocks 0 .. T-1) and K_t (T .. 2T-1).
template <typename R>
__global__ void __launch_bounds__(kStepMax) k_scan_build_rk(const Args<R> a) {
  extern __shared__ double lqg_coop_smem[];
  const int k = elem_index();
  if (k > 2 * a.T) return;
  D* sm = elem_lds(lqg_coop_smem, a.lds_elem);
  if (k <= a.T) build_riccati(a, sm, k, (long)blockIdx.y);
  else build_kalman(a, sm, k - (a.T + 1), (long)blockIdx.y);
}
template <typename R>
__global__ void __launch_bounds__(kStepMax) k_scan_gains_rk(const Args<R> a) {
  extern __shared__ double lqg_coop_smem[];
  const int k = elem_index();
  if (k >= 2 * a.T) return;
  D* sm = elem_lds(lqg_coop_smem, a.lds_elem);
  if (k < a.T) gains_step(a, sm, k, (long)blockIdx.y);
  else kgain_step(a, sm, k - a.T, (long)blockIdx.y);
}

// joint system of step t into LDS: Fj[m,m], GG[m,m]                                          system.py:167-207
template <typename R>
LQG_DEV void joint_step(const Args<R>& a, long s, int t, D* sm, D* Fj, D* GG) {
  const int x = a.x, b = a.b, u = a.u, y = a.y, m = x + b;
  const D* L = a.Lbuf + (s * a.T + t) * (long)(u * b);
  const D* K = a.Kbuf + (s * a.T + t) * (long)(b * y);
  D *Aa = sm, *Ba = Aa + b * b, *Fa = Ba + b * u, *Ad = Fa + y * b, *Bd = Ad + x * x, *Fd = Bd + x * u, *N1 = Fd + y * x,
    *WWd = N1 + x * x, *FAa = WWd + y * y, *FAd = FAa + y * b, *DB = FAd + y * x, *N2 = DB + y * u, *N3 = N2 + y * x,
    *BK = N3 + y * y, *KN2 = BK + b * u, *KN3 = KN2 + b * x;
  ld(a.aA, s, t, b, b, Aa);
  ld(a.aB, s, t, b, u, Ba);
  ld(a.aF, s, t, y, b, Fa);
  ld(a.dA, s, t, x, x, Ad);
  ld(a.dB, s, t, x, u, Bd);
  ld(a.dF, s, t, y, x, Fd);
  ld_gram(a.dV, s, t, x, a.nvd, N1);
  ld_gram(a.dW, s, t, y, a.nwd, WWd);
  wsync();
  mm(FAa, b, y, b, b, Fa, b, 1, Aa, b, 1, zero_init);
  mm(FAd, x, y, x, x, Fd, x, 1, Ad, x, 1, zero_init);
  mm(N2, x, y, x, x, Fd, x, 1, N1, x, 1, zero_init);
  each(y * u, [&](int e) {
    const int i = e / u, j = e - i * u;
    DB[e] = coop::dot4<D>(Fd + i * x, 1, Bd + j, u, x, 0.0) - coop::dot4<D>(Fa + i * b, 1, Ba + j, u, b, 0.0);
  });
  wsync();
  mm_sym(N3, y, x, N2, x, 1, Fd, 1, x, [&](int i, int j) { return WWd[i * y + j]; });
  mm(BK, u, b, u, y, K, y, 1, DB, u, 1, [&](int i, int j) { return Ba[i * u + j]; });
  mm(KN2, x, b, x, y, K, y, 1, N2, x, 1, zero_init);
  wsync();
  mm(KN3, y, b, y, y, K, y, 1, N3, y, 1, zero_init);
  wsync();
  each(m * m, [&](int e) {
    const int i = e / m, j = e - i * m;
    D f;
    if (i < x) {
      f = (j < x) ? Ad[i * x + j] : coop::dot4<D>(Bd + i * u, 1, L + (j - x), b, u, 0.0);
    } else {
      const int ib = i - x;
      if (j < x) {
        f = coop::dot4<D>(K + ib * y, 1, FAd + j, x, y, 0.0);
      } else {
        const int jb = j - x;
        D acc = Aa[ib * b + jb] - coop::dot4<D>(K + ib * y, 1, FAa + jb, b, y, 0.0);
        f = coop::dot4<D>(BK + ib * u, 1, L + jb, b, u, acc);
      }
    }
    Fj[e] = f;
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    D g;
    if (hi < x) g = N1[lo * x + hi];
    else if (lo < x) g = KN2[(hi - x) * x + lo];
    else g = 0.5 * (coop::dot4<D>(KN3 + (lo - x) * y, 1, K + (hi - x) * y, 1, y, 0.0) +
                    coop::dot4<D>(KN3 + (hi - x) * y, 1, K + (lo - x) * y, 1, y, 0.0));
    GG[e] = g;
  });
  wsync();
}
inline __host__ __device__ long joint_scratch(int x, int b, int u, int y) {
  return (long)b * b + 2L * b * u + 2L * y * b + 2L * x * x + x * u + 3L * y * x + 3L * y * y + y * u + b * x + b * y;
}

// moment-recursion elements (index k = 0 .. T-1: the conditional covariance after conditioning on x_k); block k also
// stores the joint system of step k-1.  Grid: T + 1 blocks (block T only stores the joint system of step T-1).
template <typename R>
__global__ void __launch_bounds__(kStepMax) k_scan_build_sigma(const Args<R> a) {
  extern __shared__ double lqg_coop_smem[];
  const int k = elem_index(), m = a.x + a.b, o = a.d, mm2 = m * m;
  if (k > a.T) return;
  D* sm = elem_lds(lqg_coop_smem, a.lds_elem);
  const long s = blockIdx.y;
  D *Fj = sm, *GG = Fj + mm2, *Qi = GG + mm2, *Kk = Qi + o * o, *IKH = Kk + m * o, *scratch = IKH + mm2;
  const int st = (k == 0) ? 0 : k - 1;
  joint_step(a, s, st, scratch, Fj, GG);
  if (k >= 1) {
    D* fg = a.FG + (s * a.T + st) * 2L * mm2;
    each(mm2, [&](int e) { fg[e] = Fj[e]; fg[mm2 + e] = GG[e]; });
  }
  if (k >= a.T) return;
  // The scan runs on the UNOBSERVED block only (round 3).  With the first o components observed exactly, (I - K H) has
  // zero observed rows and every conditional covariance a zero observed block, and the combine closes on the [r, r]
  // sub-blocks: M = I + C1 J2 is block triangular, M^-1 C1 = blockdiag(0, (I + C1rr J2rr)^-1 C1rr), hence
  // C = A2[r,r] (I + C1rr J2rr)^-1 C1rr A2[r,r]' + C2rr, A = A2[r,r] (..)^-1 A1[r,r], J = A1[r,r]' J2rr (..)^-1 A1[r,r] + J1rr —
  // the observed COLUMNS of A and the observed rows / columns of J only ever meet the zero block.  Elements are
  // rr x rr (rr = m - o): (m / rr)^3 fewer multiply-adds per combine (config 2: 8 -> 6, 2.4x; BoundedActor: 4 -> 2).
  const int rr = m - o, rr2 = rr * rr;
  D* e = a.elems + (s * a.T + k) * 3L * rr2;
  // Q_oo^-1, K = Q[:, :o] Q_oo^-1, I - K H (H = [I_o 0])
  each(o * o, [&](int i2) { const int i = i2 / o, j = i2 - i * o; IKH[i2] = GG[i * m + j]; });   // (IKH as temp for Q_oo)
  wsync();
  small_inverse(IKH, o, Qi, 0.0, false);
  mm(Kk, o, m, o, o, GG, m, 1, Qi, o, 1, zero_init);
  wsync();
  each(mm2, [&](int i2) {
    const int i = i2 / m, j = i2 - i * m;
    IKH[i2] = ((i == j) ? 1.0 : 0.0) - ((j < o) ? Kk[i * o + j] : 0.0);
  });
  wsync();
  // Large windows (1024 lanes per element): the unobserved rows of I - K H hold 1 + o entries each — their columns are listed
  // (coop::RowLists, in the joint system's scratch, dead by now) and the two rr x rr x m products walk the lists: exact, the
  // terms left out are exact zeros.
  const bool listed = m > 24;
  const D* Ar = IKH + o * m;                                           // rows o .. m-1 of I - K H
  coop::RowLists rl{nullptr, nullptr, 0};
  if (listed) {
    unsigned char* lp = reinterpret_cast<unsigned char*>(scratch);
    rl = coop::take_lists(lp, rr, m);
    coop::build_lists<kStepMax>((int)threadIdx.x, Ar, m, 1, rr, m, rl);
    wsync();
    each(rr2, [&](int i2) {                                            // C = ((I - K H) Q)[r, r], mirror entries averaged
      const int i = i2 / rr, j = i2 - i * rr;
      const D v1 = coop::dot_list<D>(rl, i, Ar + i * m, 1, GG + o + j, m, 0.0);
      const D v2 = coop::dot_list<D>(rl, j, Ar + j * m, 1, GG + o + i, m, 0.0);
      e[rr2 + i2] = 0.5 * (v1 + v2);
    });
  } else {
    mm_sym(e + rr2, rr, m, Ar, m, 1, GG + o, m, 1, zero_init);         // C = ((I - K H) Q)[r, r]   (k = 0: Q = Sigma_0 = GG_0)
  }
  if (k == 0) {
    each(rr2, [&](int i) { e[i] = 0.0; e[2 * rr2 + i] = 0.0; });
  } else {
    if (listed) {
      each(rr2, [&](int i2) {                                          // A = ((I - K H) F)[r, r]
        const int i = i2 / rr, j = i2 - i * rr;
        e[i2] = coop::dot_list<D>(rl, i, Ar + i * m, 1, Fj + o + j, m, 0.0);
      });
    } else {
      mm(e, rr, rr, rr, m, Ar, m, 1, Fj + o, m, 1, zero_init);         // A = ((I - K H) F)[r, r]
    }
    // J = (F[:o, r])' Q_oo^-1 F[:o, r]
    D* T1 = Kk;                                                      // [o, m] temp (Kk is dead after IKH)
    wsync();
    mm(T1, m, o, m, o, Qi, o, 1, Fj, m, 1, zero_init);
    wsync();
    mm_sym(e + 2 * rr2, rr, o, Fj + o, 1, m, T1 + o, m, 1, zero_init);
  }
}

// trial operators of step t (t = 0 .. T) from the predictive covariance Sigma_t                system.py:219-230, 244-248
template <typename R>
__global__ void __launch_bounds__(kStepMax) k_scan_ops(const Args<R> a) {
  extern __shared__ double lqg_coop_smem[];
  const int t = elem_index(), m = a.x + a.b, o = a.d, rr = m - o, mm2 = m * m;
  if (t > a.T) return;
  D* sm = elem_lds(lqg_coop_smem, a.lds_elem);
  const long s = blockIdx.y;
  D *Sg = sm, *T1 = Sg + mm2, *Lis = T1 + mm2, *hls = Lis + o * o;
  if (t == 0) {
    const D* fg = a.FG + (s * a.T) * 2L * mm2;
    each(mm2, [&](int e) { Sg[e] = fg[mm2 + e]; });                  // Sigma_0 = G_0 G_0'   system.py:212
  } else {
    const D* fg = a.FG + (s * a.T + t - 1) * 2L * mm2;
    const D* C = a.res + (s * a.T + t - 1) * 3L * (rr * rr) + rr * rr;      // conditional covariance, unobserved block
    if (m > 24) {
      // large windows (1024 lanes per element): F2 = F[:, o:] of the delay augmentations is shift-structured (9 % non-zero at
      // m = 65) — the columns of its rows are listed once (one wave per row) and both products walk the lists; the terms left
      // out are exact zeros (coop::RowLists, the lists of the sequential sweeps)
      unsigned char* lp = reinterpret_cast<unsigned char*>(hls + 1);
      const coop::RowLists rl = coop::take_lists(lp, m, rr);
      const D* F2 = fg + o;
      coop::build_lists<kStepMax>((int)threadIdx.x, F2, m, 1, m, rr, rl);
      wsync();
      each(m * rr, [&](int e) {                                      // F2 C        [m, rr]
        const int i = e / rr, j = e - i * rr;
        T1[e] = coop::dot_list<D>(rl, i, F2 + i * m, 1, C + j, rr, 0.0);
      });
      wsync();
      each(mm2, [&](int e) {                                         // F2 C F2' + GG, mirror entries averaged
        const int i = e / m, j = e - i * m;
        const D v1 = coop::dot_list<D>(rl, j, F2 + j * m, 1, T1 + i * rr, 1, fg[mm2 + i * m + j]);
        const D v2 = coop::dot_list<D>(rl, i, F2 + i * m, 1, T1 + j * rr, 1, fg[mm2 + j * m + i]);
        Sg[e] = 0.5 * (v1 + v2);
      });
    } else {
    mm(T1, rr, m, rr, rr, fg + o, m, 1, C, rr, 1, zero_init);       // F[:, o:] C        [m, rr]
    wsync();
    mm_sym(Sg, m, rr, T1, rr, 1, fg + o, 1, m, [&](int i, int j) { return fg[mm2 + i * m + j]; });   // F2 C F2' + GG
    }
    if (a.Sig.p) {
      wsync();
      R* out = const_cast<R*>(a.Sig.p) + s * a.Sig.sb + (long)(t - 1) * a.Sig.st;
      each(mm2, [&](int e) { const int i = e / m, j = e - i * m; out[i * a.Sig.sr + j * a.Sig.sc] = (R)Sg[e]; });
    }
  }
  wsync();
  const D kLogNorm = 0.5 * 1.8378770664093453 * (D)o;
  R* op = a.ops + (s * (a.T + 1) + t) * (long)a.nops;
  coop::dispatch_small<0>(o, [&](auto no_) {
    constexpr int NO = decltype(no_)::value;
    D Li[NO * NO], hl;
    coop::chol_inverse_reg<D, NO>(Sg, m, Li, hl);
    if (threadIdx.x == 0) {
      LQG_UNROLL for (int e = 0; e < NO * NO; ++e) Lis[e] = Li[e];
      int e = 0;
      LQG_UNROLL for (int i = 0; i < NO; ++i)
        LQG_UNROLL for (int j = 0; j <= i; ++j) op[m * m + rr * o + (e++)] = (R)Li[i * NO + j];
      op[m * m + rr * o + e] = (R)(hl + kLogNorm);
      hls[0] = hl;
    }
  });
  wsync();
  if (t < a.T) {
    each(rr * o, [&](int e) {                                        // U2 = S_ro Li'
      const int p = e / o, j = e - p * o;
      op[m * m + e] = (R)coop::dot4<D>(Sg + (o + p) * m, 1, Lis + j * o, 1, j + 1, 0.0);
    });
    const D* fg = a.FG + (s * a.T + t) * 2L * mm2;
    each(mm2, [&](int e) {
      const int i = e / m, j = e - i * m;
      op[e] = (R)((i == j) ? fg[e] - 1.0 : fg[e]);                  // Fj - I, rounded to the problem dtype once
    });
  }
}

}  // namespace scan
}  // namespace lqg
