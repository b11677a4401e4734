// lqg_scan_inst.hip — host side of the time-parallel system sweeps (lqg_scan.hpp): workspace accounting and the launch
// sequence  elements -> log2(T) scan levels -> per-step finalisers  for the Riccati, Kalman and moment recursions.
#include <cstdlib>

#include "lqg_scan.hpp"
#include "lqg_coop_launch.hpp"
#include "lqg_launch.hpp"

namespace lqg {
namespace host {

namespace {
using scan::D;

struct ScanPlan {
  size_t elems_off, elems_bytes, l_off, k_off, fg_off, ops_off, total;
  long elem_reals;        // doubles of ONE element buffer (largest of the three scans)
};

inline size_t al(size_t v) { return (v + 255) / 256 * 256; }

ScanPlan scan_plan(const lqg_problem* p) {
  const long B = p->n_sys, T = p->T, b = p->dims.b, u = p->dims.u, y = p->dims.y, m = p->dims.x + p->dims.b;
  const size_t esz = p->dtype == LQG_F64 ? 8 : 4;
  ScanPlan s{};
  const long r1 = (T + 1) * 3 * b * b, r2 = T * 3 * m * m;
  s.elem_reals = B * (r1 > r2 ? r1 : r2);
  s.elems_off = 0;
  s.elems_bytes = al(2 * (size_t)s.elem_reals * sizeof(D));
  s.l_off = s.elems_off + s.elems_bytes;
  s.k_off = s.l_off + al((size_t)(B * T * u * b) * sizeof(D));
  s.fg_off = s.k_off + al((size_t)(B * T * b * y) * sizeof(D));
  s.ops_off = s.fg_off + al((size_t)(B * T * 2 * m * m) * sizeof(D));
  s.total = s.ops_off + al((size_t)B * (size_t)(T + 1) * ops_reals(p->dims) * esz);
  return s;
}

// Hillis-Steele over `len` elements of n x n triples starting in buffer 0; returns the buffer holding the result
D* run_scan(D* buf0, D* buf1, int n, int len, long n_sys, int left, hipStream_t st) {
  const size_t lds = (size_t)(10 * n * n + n + 8) * sizeof(D);
  D *in = buf0, *out = buf1;
  for (int d = 1; d < len; d *= 2) {
    hipLaunchKernelGGL(scan::k_scan_level, dim3((unsigned)len, (unsigned)n_sys), dim3(scan::kWave), lds, st, in, out, n, len,
                       d, left);
    D* t = in;
    in = out;
    out = t;
  }
  return in;
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

bool scan_supported(const lqg_problem* p) {
  const lqg_dims& d = p->dims;
  const int m = d.x + d.b;
  return d.u >= 1 && d.y >= 1 && d.d >= 1 && d.u <= 4 && d.y <= 4 && d.d <= 4 && d.y <= d.b && d.d <= d.x && m <= 24 &&
         p->T >= 2 && !affine(p);
}

size_t scan_workspace_bytes(const lqg_problem* p) { return scan_plan(p).total; }

// Riccati + Kalman + moment recursion as scans; leaves the trial-operator stream at the returned offset of `workspace`
template <typename R>
hipError_t scan_system_sweeps(const lqg_problem* p, lqg_view Sig, void* workspace, void** ops_out, hipStream_t st) {
  const ScanPlan sp = scan_plan(p);
  char* base = static_cast<char*>(workspace);
  D* buf0 = reinterpret_cast<D*>(base + sp.elems_off);
  D* buf1 = buf0 + sp.elem_reals;
  scan::Args<R> k = make_scan_args<R>(p);
  k.Sig = dv<R>(Sig);
  k.Lbuf = reinterpret_cast<D*>(base + sp.l_off);
  k.Kbuf = reinterpret_cast<D*>(base + sp.k_off);
  k.FG = reinterpret_cast<D*>(base + sp.fg_off);
  k.ops = reinterpret_cast<R*>(base + sp.ops_off);
  *ops_out = k.ops;
  const int T = p->T, x = p->dims.x, b = p->dims.b, u = p->dims.u, y = p->dims.y, o = p->dims.d, m = x + b;
  const unsigned B = (unsigned)p->n_sys;
  const dim3 blk(scan::kWave);
  const int mx = b > y ? b : y;
  // ---- Riccati: suffix scan over T + 1 elements (reversed storage)
  k.elems = buf0;
  hipLaunchKernelGGL((scan::k_scan_build_riccati<R>), dim3(T + 1, B), blk, (size_t)(2 * b * u + 2 * u * u + 8) * sizeof(D), st, k);
  k.res = run_scan(buf0, buf1, b, T + 1, p->n_sys, 1, st);
  hipLaunchKernelGGL((scan::k_scan_gains<R>), dim3(T, B), blk,
                     (size_t)(3 * b * b + 3 * b * u + 3 * u * u + 8) * sizeof(D), st, k);
  // ---- Kalman: prefix scan over T elements
  D* kin = (k.res == buf0) ? buf1 : buf0;       // (the Riccati result is dead after k_scan_gains: stream order)
  D* kout = (kin == buf0) ? buf1 : buf0;
  k.elems = kin;
  hipLaunchKernelGGL((scan::k_scan_build_kalman<R>), dim3(T, B), blk, (size_t)(6 * mx * mx + 6 * mx * mx + 8) * sizeof(D), st, k);
  k.res = run_scan(kin, kout, b, T, p->n_sys, 0, st);
  hipLaunchKernelGGL((scan::k_scan_kgain<R>), dim3(T, B), blk, (size_t)(12 * mx * mx + 8) * sizeof(D), st, k);
  if (const char* dbg = getenv("LQG_SCAN_DEBUG_STOP")) {     // developer hook: leave the Kalman scan's buffers intact
    if (dbg[0] == '1') { *ops_out = const_cast<D*>(k.res); return hipGetLastError(); }
  }
  // ---- moment recursion: joint system per step, prefix scan over T elements of m x m, operators
  D* sin = (k.res == buf0) ? buf1 : buf0;
  D* sout = (sin == buf0) ? buf1 : buf0;
  k.elems = sin;
  const size_t lds_sig = (size_t)(3 * m * m + o * o + m * o + scan::joint_scratch(x, b, u, y) + 16) * sizeof(D);
  hipLaunchKernelGGL((scan::k_scan_build_sigma<R>), dim3(T + 1, B), blk, lds_sig, st, k);
  k.res = run_scan(sin, sout, m, T, p->n_sys, 0, st);
  hipLaunchKernelGGL((scan::k_scan_ops<R>), dim3(T + 1, B), blk, (size_t)(2 * m * m + o * o + 8) * sizeof(D), st, k);
  return hipGetLastError();
}

template hipError_t scan_system_sweeps<float>(const lqg_problem*, lqg_view, void*, void**, hipStream_t);
template hipError_t scan_system_sweeps<double>(const lqg_problem*, lqg_view, void*, void**, hipStream_t);

}  // namespace host
}  // namespace lqg
