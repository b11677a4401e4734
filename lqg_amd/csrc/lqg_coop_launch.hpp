// lqg_coop_launch.hpp — host side of the cooperative (workgroup-per-system, run-time dims) kernels of lqg_coop.hpp:
// strategy decision, workspace accounting and launch wrappers.  Defined once in lqg_coop_inst.hip (one translation
// unit for every model shape), declared here for lqg_abi.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <map>
#include <mutex>
#include <utility>

#include "../../include/lqg_hip.h"

namespace lqg {
namespace host {
// hipFuncAttributeMaxDynamicSharedMemorySize is a property of a kernel ON A DEVICE, not of a launch: raised once per
// (device, kernel) and size — never again from inside a stream capture, where the replays of the inference loops launch these
// kernels.  (Rounds 2-3 keyed the record on the kernel alone: a process driving a second GPU skipped the raise there and its
// launches with more than 64 KB of LDS failed.)  The only state the library keeps; result-neutral, behind a mutex.
inline hipError_t raise_dynamic_lds(const void* kernel, size_t bytes, size_t without_raise = 64 * 1024) {
  if (bytes <= without_raise) return hipSuccess;
  static std::mutex mu;
  static std::map<std::pair<int, const void*>, size_t> raised;
  int dev = 0;
  if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
  std::lock_guard<std::mutex> lock(mu);
  size_t& have = raised[{dev, kernel}];
  if (bytes <= have) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e == hipSuccess) have = bytes;
  return e;
}


// dims the cooperative kernels serve: u, y, d <= 4 (coop::kMaxSmall: their small factorizations run in registers); x, b unbounded
bool coop_supported(const lqg_dims& d);
// bytes of the global working set the cooperative path needs BEHIND the gain scratch + operator stream of carve():
// 0 when the per-system working set fits LDS
size_t coop_arena_bytes(const lqg_problem* p, bool kalman_only, bool riccati_only);
// true when the call can run without any global arena (stand-alone lqr.backward / kf.forward take no workspace)
bool coop_fits_lds(const lqg_problem* p, bool kalman_only, bool riccati_only);

template <typename R>
hipError_t coop_riccati(const lqg_problem* p, lqg_view L, lqg_view l, lqg_view H, void* Ls, void* arena, hipStream_t st);
template <typename R>
hipError_t coop_forward(const lqg_problem* p, const void* Ls, void* ops, lqg_view Sig, lqg_view K, void* arena,
                        hipStream_t st);
template <typename R>
hipError_t coop_trial(const lqg_problem* p, const void* ops, lqg_traj x, lqg_traj mu, void* ll, long ll_sb, long ll_sn,
                      hipStream_t st);

// run-time-dims twin of launch_simulate (any x, b, u, y whose per-thread state fits LDS)
template <typename R>
hipError_t coop_simulate(const lqg_problem* p, lqg_view L, lqg_view l, lqg_view K, lqg_traj eps, lqg_traj eta, lqg_view x0,
                         lqg_view xhat0, lqg_traj xs, lqg_traj xhat, lqg_traj ys, lqg_traj us, hipStream_t st,
                         unsigned long long seed = 0);

// ---- time-parallel system sweeps (lqg_scan.hpp / lqg_scan_inst.hip)
// dims and problem class the scan path serves (u, y, d <= 4, x + b <= 24, no affine cost terms); the caller additionally
// guarantees that the eigenvalue floor of lqr.py:27-28 is inactive
// reverse-mode gradient for shapes without adjoint lane kernels (lqg_coop_adjoint.hip): one workgroup per system, fp64,
// bars summed over the trials
int coop_adjoint_supported(int32_t dtype, const lqg_dims& d);
size_t coop_adjoint_workspace_bytes(const lqg_problem* p);
hipError_t coop_adjoint_run(const lqg_problem* p, lqg_traj x, const void* g, long g_sb, long g_sn, void* ll, long ll_sb,
                            long ll_sn, void* grad, long ld, long elements, bool time_invariant, void* ws, int phases,
                            hipStream_t st);

bool scan_supported(const lqg_problem* p);
size_t scan_workspace_bytes(const lqg_problem* p);
template <typename R>
hipError_t scan_system_sweeps(const lqg_problem* p, lqg_view Sig, void* workspace, void** ops_out, hipStream_t st);

}  // namespace host
}  // namespace lqg
