// lqg_scan_inst.hip — host side of the time-parallel system sweeps (lqg_scan.hpp): workspace accounting and the launch
// sequence  elements -> log2(T) scan levels -> per-step finalisers  for the Riccati, Kalman and moment recursions.
#include <cstdlib>
#include <mutex>
#include <unordered_map>

#include "lqg_scan.hpp"
#include "lqg_coop_launch.hpp"
#include "lqg_launch.hpp"

namespace lqg {
namespace host {

namespace {
using scan::D;

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of the kernel, not of a launch: raised once per kernel and size —
// never again from inside a stream capture, where the replays of the inference loops launch these kernels
hipError_t raise_lds_once(const void* kern, size_t bytes) { return raise_dynamic_lds(kern, bytes); }   // lqg_coop_launch.hpp

struct ScanPlan {
  size_t elems_off, elems_bytes, l_off, k_off, fg_off, ops_off, total;
  long rk_reals;          // doubles of ONE element buffer of the Riccati / Kalman scans (four of them: in/out each)
  long sig_reals;         // doubles of ONE element buffer of the moment scan (two of them, reusing the same region)
};

inline size_t al(size_t v) { return (v + 255) / 256 * 256; }

ScanPlan scan_plan(const lqg_problem* p) {
  const long B = p->n_sys, T = p->T, b = p->dims.b, u = p->dims.u, y = p->dims.y, m = p->dims.x + p->dims.b;
  const size_t esz = p->dtype == LQG_F64 ? 8 : 4;
  ScanPlan s{};
  s.rk_reals = B * (T + 1) * 3 * b * b;
  s.sig_reals = B * T * 3 * m * m;
  const long reals = 4 * s.rk_reals > 2 * s.sig_reals ? 4 * s.rk_reals : 2 * s.sig_reals;
  s.elems_off = 0;
  s.elems_bytes = al((size_t)reals * sizeof(D));
  s.l_off = s.elems_off + s.elems_bytes;
  s.k_off = s.l_off + al((size_t)(B * T * u * b) * sizeof(D));
  s.fg_off = s.k_off + al((size_t)(B * T * b * y) * sizeof(D));
  s.ops_off = s.fg_off + al((size_t)(B * T * 2 * m * m) * sizeof(D));
  s.total = s.ops_off + al((size_t)B * (size_t)(T + 1) * ops_reals(p->dims) * esz) + trial_chunk_scratch(p).total;
  return s;
}

template <int N, bool PACKED>
void launch_level_v(const scan::Seg& s0, const scan::Seg& s1, long n_sys, hipStream_t st) {
  constexpr int NT = scan::scan_level_threads(N, PACKED), EPB = scan::scan_level_epb(N, PACKED);
  hipLaunchKernelGGL((scan::k_scan_level<N, NT, EPB>), dim3((unsigned)((s0.len + s1.len + EPB - 1) / EPB), (unsigned)n_sys),
                     dim3(NT * EPB), scan::scan_level_lds(N, PACKED), st, s0, s1);
}
template <int N>
void launch_level(const scan::Seg& s0, const scan::Seg& s1, long n_sys, hipStream_t st) {
  // sub-wave packing of small windows once a level holds enough elements to be throughput-bound
  if (scan::scan_level_threads(N, true) != scan::scan_level_threads(N, false) && n_sys * (long)(s0.len + s1.len) >= 4096)
    launch_level_v<N, true>(s0, s1, n_sys, st);
  else
    launch_level_v<N, false>(s0, s1, n_sys, st);
}

template <int NW>
void launch_level_rt_v(const scan::Seg& s0, const scan::Seg& s1, int n, long n_sys, hipStream_t st) {
  auto kern = scan::k_scan_level_rt<NW>;
  const size_t lds = scan::scan_level_rt_lds(n);
  (void)raise_lds_once(reinterpret_cast<const void*>(kern), lds);
  hipLaunchKernelGGL(kern, dim3((unsigned)(scan::seg_count(s0) + scan::seg_count(s1)), (unsigned)n_sys),
                     dim3(scan::scan_level_rt_threads(n, NW)), lds, st, s0, s1, n);
}
void launch_level_rt(const scan::Seg& s0, const scan::Seg& s1, int n, long n_sys, const lqg_tuning& tune, hipStream_t st) {
  // 16 waves per window (4 rows each): the elimination is bound by its two barriers and LDS round trips per column, not by
  // issue — delay12 system sweeps 7.0 / 4.5 / 4.1 ms at 4 / 8 / 16 waves (tuning.scan_rt_waves = 8 keeps the A/B)
  if (tune.scan_rt_waves == 8) launch_level_rt_v<8>(s0, s1, n, n_sys, st);
  else launch_level_rt_v<16>(s0, s1, n, n_sys, st);
}

// Hillis-Steele over one or two independent sequences of n x n triples (segment i: len[i] elements starting in in[i],
// ping-ponging with out[i]); on return res[i] is the buffer holding segment i's result.
void run_scan(int n, int nseg, D* const in[2], D* const out[2], const int len[2], const int left[2], long n_sys,
              const D* res[2], const lqg_tuning& tune, hipStream_t st) {
  D* a[2] = {in[0], nseg > 1 ? in[1] : nullptr};
  D* b[2] = {out[0], nseg > 1 ? out[1] : nullptr};
  const int l1 = nseg > 1 ? len[1] : 0;
  const int longest = len[0] > l1 ? len[0] : l1;
  // windows of 1 .. 3 whose whole sequence fits one workgroup: every level in ONE launch (k_scan_lane)
  const bool lane_off = tune.scan_lane < 0;
  if (!lane_off && n <= scan::kScanLaneMaxN && longest <= scan::scan_lane_max_len(n) && scan::scan_lane_lds(n, longest) <= 150 * 1024) {
    const scan::Seg s0{a[0], b[0], len[0], 0, left[0]};
    const scan::Seg s1{a[1], b[1], l1, 0, nseg > 1 ? left[1] : 0};
    const size_t lds = scan::scan_lane_lds(n, longest);
    const dim3 grid((unsigned)nseg, (unsigned)n_sys), blk((unsigned)((longest + 63) / 64 * 64));
    auto go = [&](auto kern) {
      (void)raise_lds_once(reinterpret_cast<const void*>(kern), lds);
      hipLaunchKernelGGL(kern, grid, blk, lds, st, s0, s1);
    };
    if (n == 1) go(scan::k_scan_lane<1, 1024>);
    else if (n == 2) go(scan::k_scan_lane<2, 1024>);
    else go(scan::k_scan_lane<3, 512>);
    res[0] = b[0];
    res[1] = b[1];
    return;
  }
  // Windows of 25 .. 64 with more combines per Hillis-Steele level than the chip runs at once (one 1024-lane workgroup per CU, a
  // level's time is its number of ROUNDS of 256 concurrent combines): a work-efficient order.  Brent-Kung in place — up-sweep
  // (blocks of 2 d ending at k = 2 d - 1 mod 2 d), down-sweep (k = 3 d - 1 mod 2 d takes the finished prefix before its block):
  // ~2 len combines in 2 log2(len) - 2 levels instead of len log2(len) in log2(len) — with its nearly empty middle levels
  // (31, 15, 7, 3, 1, 1, 3, 7, ... combines, each a full 50 .. 95 us round) replaced by a Hillis-Steele scan over the BLOCK TOTALS:
  // up-sweep to blocks of B, ping-pong scan over the len / B totals at k = B - 1 mod B (log2(len / B) levels of one round each,
  // B the smallest power of two whose totals fit one round), down-sweep from d = B / 2.  One delay-12 system (501 + 500 windows of
  // 39, then 500 of 63), rounds: Hillis-Steele 36 + 18, Brent-Kung 18 + 16, this 13 + 10.
  // tuning.scan_order: 0 = this rule, 1 = always, 2 = plain Brent-Kung, -1 = Hillis-Steele.
  const int order = tune.scan_order;
  if (n > 24 && order >= 0 && (order > 0 || n_sys * (long)(len[0] + l1) > scan::kScanRtConcurrent)) {
    const int lens[2] = {len[0], l1}, lefts[2] = {left[0], nseg > 1 ? left[1] : 0};
    // one strided level: out[k] = in[k - d] (x) in[k] for k = k0 + i ks, i < cnt (copies where k < d)
    auto level = [&](D* const from[2], D* const to[2], int d, int k0, int ks, const int cnt[2]) {
      scan::Seg sg[2];
      for (int i = 0; i < 2; ++i) {
        sg[i] = scan::Seg{from[i], to[i], lens[i], d, lefts[i]};
        sg[i].k0 = k0;
        sg[i].ks = ks;
        sg[i].cnt = cnt[i];
      }
      if (cnt[0] + cnt[1] > 0) launch_level_rt(sg[0], sg[1], n, n_sys, tune, st);
    };
    int B = 2;                                                        // block of the middle scan
    if (order == 2) B = 2 * longest;                                  // (no middle scan)
    else while (B < longest && n_sys * (long)(lens[0] / B + lens[1] / B) > scan::kScanRtConcurrent) B *= 2;
    int nb_max = lens[0] / B > lens[1] / B ? lens[0] / B : lens[1] / B, mid_levels = 0;
    for (int dj = 1; dj < nb_max; dj *= 2) ++mid_levels;
    int top = 1;                                                      // largest up-sweep distance: B / 2, or plain Brent-Kung's
    while (2 * top < B && 4 * top <= longest) top *= 2;
    for (int d = 1; d <= top; d *= 2) {                               // up-sweep, in place; the level that completes the blocks of B
      const int cnt[2] = {lens[0] / (2 * d), lens[1] / (2 * d)};      // writes the other buffer when the middle scan has an odd
      const bool flip = 2 * d == B && (mid_levels & 1);               // number of levels (its last level must land in `a`)
      level(a, flip ? b : a, d, 2 * d - 1, 2 * d, cnt);
    }
    if (mid_levels > 0) {
      D* cur[2] = {(mid_levels & 1) ? b[0] : a[0], (mid_levels & 1) ? b[1] : a[1]};
      D* oth[2] = {(mid_levels & 1) ? a[0] : b[0], (mid_levels & 1) ? a[1] : b[1]};
      const int cnt[2] = {lens[0] / B, lens[1] / B};
      for (int dj = 1; dj < nb_max; dj *= 2) {
        level(cur, oth, dj * B, B - 1, B, cnt);
        for (int i = 0; i < 2; ++i) { D* t = cur[i]; cur[i] = oth[i]; oth[i] = t; }
      }
    }
    for (int d = top; d >= 1; d /= 2) {                               // down-sweep, in place
      const int cnt[2] = {lens[0] > d ? (lens[0] - d) / (2 * d) : 0, lens[1] > d ? (lens[1] - d) / (2 * d) : 0};
      level(a, a, d, 3 * d - 1, 2 * d, cnt);
    }
    res[0] = a[0];
    res[1] = a[1];
    return;
  }
  for (int d = 1; d < longest; d *= 2) {
    const scan::Seg s0{a[0], b[0], len[0], d, left[0]};
    const scan::Seg s1{a[1], b[1], l1, d, nseg > 1 ? left[1] : 0};
    switch (n) {
#define LQG_SCAN_CASE(N_) case N_: launch_level<N_>(s0, s1, n_sys, st); break;
      LQG_SCAN_CASE(1) LQG_SCAN_CASE(2) LQG_SCAN_CASE(3) LQG_SCAN_CASE(4) LQG_SCAN_CASE(5) LQG_SCAN_CASE(6)
      LQG_SCAN_CASE(7) LQG_SCAN_CASE(8) LQG_SCAN_CASE(9) LQG_SCAN_CASE(10) LQG_SCAN_CASE(11) LQG_SCAN_CASE(12)
      LQG_SCAN_CASE(13) LQG_SCAN_CASE(14) LQG_SCAN_CASE(15) LQG_SCAN_CASE(16) LQG_SCAN_CASE(17) LQG_SCAN_CASE(18)
      LQG_SCAN_CASE(19) LQG_SCAN_CASE(20) LQG_SCAN_CASE(21) LQG_SCAN_CASE(22) LQG_SCAN_CASE(23) LQG_SCAN_CASE(24)
#undef LQG_SCAN_CASE
      default:                        // windows of 25 .. 64 (scan_supported): registers + scalar operands, run-time n
        launch_level_rt(s0, s1, n, n_sys, tune, st);
        break;
    }
    for (int i = 0; i < 2; ++i) { D* t = a[i]; a[i] = b[i]; b[i] = t; }
  }
  res[0] = a[0];
  res[1] = a[1];
}

template <typename R>
scan::Args<R> make_scan_args(const lqg_problem* p) {
  const lqg_spec& a = p->actor;
  const lqg_spec& d = p->dynamics;
  scan::Args<R> k{};
  k.aQ = dv<R>(a.Q); k.aQf = dv<R>(a.Qf); k.aR = dv<R>(a.R); k.aA = dv<R>(a.A); k.aB = dv<R>(a.B); k.aF = dv<R>(a.F);
  k.aV = dv<R>(a.V); k.aW = dv<R>(a.W);
  k.dA = dv<R>(d.A); k.dB = dv<R>(d.B); k.dF = dv<R>(d.F); k.dV = dv<R>(d.V); k.dW = dv<R>(d.W);
  k.Sigma0 = dv<R>(p->Sigma0);
  k.n_sys = (long)p->n_sys;
  k.T = p->T;
  k.x = p->dims.x; k.b = p->dims.b; k.u = p->dims.u; k.y = p->dims.y; k.d = p->dims.d;
  k.nva = p->dims.nva; k.nwa = p->dims.nwa; k.nvd = p->dims.nvd; k.nwd = p->dims.nwd;
  k.nops = (int)ops_reals(p->dims);
  k.eps = p->eps;
  return k;
}
}  // namespace

// LDS doubles per element of the four per-step kernels
struct StepLds { long build_rk, gains_rk, build_sigma, ops; };
StepLds step_lds(const lqg_dims& d) {
  const long x = d.x, b = d.b, u = d.u, y = d.y, o = d.d, m = x + b, mx = b > y ? b : y;
  (void)mx;
  // (exact sums of the working sets of build_riccati / build_kalman and of gains_step / kgain_step, lqg_scan.hpp — the round-3
  // bound of 12 max(b, y)^2 kept the delay models' Riccati / Kalman builders at one workgroup per CU: 146 KB at b = 39, now 63 KB)
  const long rkl = m > 24 ? (coop::row_lists_bytes((int)b, (int)b) + 7) / 8 : 0;          // (+ the row / column lists of A, I - K F)
  return StepLds{5L * b * b + 3 * y * b + 3 * y * y + 2 * b * u + 2 * u * u + 8 + rkl, 5L * b * b + 2 * y * b + 3 * y * y + 3 * b * u + 3 * u * u + 8 + rkl,
                 3L * m * m + o * o + m * o + scan::joint_scratch((int)x, (int)b, (int)u, (int)y) + 16,
                 2L * m * m + o * o + 8 + (m > 24 ? (coop::row_lists_bytes((int)m, (int)(m - o)) + 7) / 8 : 0)};
}
constexpr long kLdsMaxDoubles = 160 * 1024 / 8;

bool scan_supported(const lqg_problem* p) {
  const lqg_dims& d = p->dims;
  const int m = d.x + d.b;
  if (!(d.u >= 1 && d.y >= 1 && d.d >= 1 && d.u <= 4 && d.y <= 4 && d.d <= 4 && d.y <= d.b && d.d <= d.x && p->T >= 2 &&
        !affine(p)))
    return false;
  if (m <= 24) return true;
  // larger windows (the delay-augmented models): one lane per column of a window, per-step working sets within LDS
  const StepLds l = step_lds(d);
  const long mx = d.b > d.y ? d.b : d.y;
  if (12L * mx * mx > kLdsMaxDoubles) return false;              // (the range the round-3 bound admitted and the tests cover: b <= 41)
  return d.b <= scan::kScanRtMax && m - d.d <= scan::kScanRtMax && l.build_rk <= kLdsMaxDoubles && l.gains_rk <= kLdsMaxDoubles &&
         l.build_sigma <= kLdsMaxDoubles && l.ops <= kLdsMaxDoubles;
}

size_t scan_workspace_bytes(const lqg_problem* p) { return scan_plan(p).total; }

// Riccati + Kalman + moment recursion as scans; leaves the trial-operator stream at the returned offset of `workspace`
template <typename R>
hipError_t scan_system_sweeps(const lqg_problem* p, lqg_view Sig, void* workspace, void** ops_out, hipStream_t st) {
  const ScanPlan sp = scan_plan(p);
  char* base = static_cast<char*>(workspace);
  D* rk = reinterpret_cast<D*>(base + sp.elems_off);
  scan::Args<R> k = make_scan_args<R>(p);
  k.Sig = dv<R>(Sig);
  k.Lbuf = reinterpret_cast<D*>(base + sp.l_off);
  k.Kbuf = reinterpret_cast<D*>(base + sp.k_off);
  k.FG = reinterpret_cast<D*>(base + sp.fg_off);
  k.ops = reinterpret_cast<R*>(base + sp.ops_off);
  *ops_out = k.ops;
  const int T = p->T, x = p->dims.x, b = p->dims.b, u = p->dims.u, y = p->dims.y, o = p->dims.d, m = x + b;
  const unsigned B = (unsigned)p->n_sys;
  // lanes per element x elements per (single-wave) workgroup of the per-step kernels: sub-wave once a launch holds
  // thousands of small elements (36 candidates of a 4 x 4 model: the four kernels 168 -> ~60 us)
  const bool packed = (long)p->n_sys * T >= 4096 && m <= 8;
  const dim3 blk(packed ? 16 : m > 24 ? scan::kStepMax : 64, packed ? 4 : 1);      // (m > 24: sixteen waves per element)
  const StepLds sl = step_lds(p->dims);
  hipError_t attr = hipSuccess;
  auto launch = [&](auto kern, int count, long lds_doubles, unsigned threads = 0) {
    k.lds_elem = (int)lds_doubles;
    const size_t lds = (size_t)lds_doubles * blk.y * sizeof(D);
    const hipError_t e = raise_lds_once(reinterpret_cast<const void*>(kern), lds);
    if (e != hipSuccess) attr = e;
    const dim3 bl(threads ? threads : blk.x, blk.y);
    hipLaunchKernelGGL(kern, dim3((unsigned)((count + blk.y - 1) / blk.y), B), bl, lds, st, k);
  };
  // large windows: the Riccati / Kalman builders and finalisers on 512 lanes per element — at ~100 VGPRs and 63 KB of LDS two
  // workgroups share a CU, and their 2 T + 1 elements per system are four rounds of the chip on 1024 lanes
  const unsigned rk_threads = m > 24 ? scan::kStepRk : 0;
  // ---- Riccati (suffix scan over T + 1 elements, reversed storage) and Kalman (prefix scan over T elements) side by side
  {
    D* const in[2] = {rk, rk + 2 * sp.rk_reals};
    D* const out[2] = {rk + sp.rk_reals, rk + 3 * sp.rk_reals};
    const int len[2] = {T + 1, T}, left[2] = {1, 0};
    const D* res[2];
    k.elems = in[0];
    k.elems2 = in[1];
    launch(scan::k_scan_build_rk<R>, 2 * T + 1, sl.build_rk, rk_threads);
    run_scan(b, 2, in, out, len, left, p->n_sys, res, p->tuning, st);
    k.res = res[0];
    k.res2 = res[1];
    launch(scan::k_scan_gains_rk<R>, 2 * T, sl.gains_rk, rk_threads);
  }
  // ---- moment recursion: joint system per step, prefix scan over T elements of m x m, operators
  {
    D* const in[2] = {rk, nullptr};
    D* const out[2] = {rk + sp.sig_reals, nullptr};
    const int len[2] = {T, 0}, left[2] = {0, 0};
    const D* res[2];
    k.elems = in[0];
    launch(scan::k_scan_build_sigma<R>, T + 1, sl.build_sigma);
    run_scan(m - o, 1, in, out, len, left, p->n_sys, res, p->tuning, st);     // (elements on the unobserved block: lqg_scan.hpp)
    k.res = res[0];
  }
  launch(scan::k_scan_ops<R>, T + 1, sl.ops);
  return attr != hipSuccess ? attr : hipGetLastError();
}

template hipError_t scan_system_sweeps<float>(const lqg_problem*, lqg_view, void*, void**, hipStream_t);
template hipError_t scan_system_sweeps<double>(const lqg_problem*, lqg_view, void*, void**, hipStream_t);

}  // namespace host
}  // namespace lqg

#ifdef LQG_SCAN_STAMP
extern "C" int lqg_debug_scan_stamps(unsigned long long* out16, int reset) {
  (void)hipDeviceSynchronize();
  hipError_t e = hipMemcpyFromSymbol(out16, HIP_SYMBOL(lqg::scan::g_scan_stamps), 16 * sizeof(unsigned long long));
  if (reset) {
    unsigned long long z[16] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(lqg::scan::g_scan_stamps), z, sizeof(z));
  }
  return (int)e;
}
#endif
