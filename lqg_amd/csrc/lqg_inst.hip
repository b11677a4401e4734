// lqg_inst.hip — explicit instantiation of ONE launcher (both dtypes) per translation unit, selected with
//   -DLQG_INST_RICCATI="b,u" | -DLQG_INST_KALMAN="b,y" | -DLQG_INST_FORWARD="x,b,u,y,d" |
//   -DLQG_INST_TRIAL="m,d" | -DLQG_INST_SIM="x,b,u,y"
// together with -DLQG_INST_F32 and/or -DLQG_INST_F64, so that lqg_amd/build.py can compile the (large, fully unrolled)
// kernels in parallel (one translation unit per family x dims x dtype).
#include "lqg_launch.hpp"

namespace lqg {
namespace host {
#ifdef LQG_INST_RICCATI
#ifdef LQG_INST_F32
template hipError_t launch_riccati<float, LQG_INST_RICCATI>(const lqg_problem*, lqg_view, lqg_view, lqg_view, void*,
                                                            long, hipStream_t);
#endif
#ifdef LQG_INST_F64
template hipError_t launch_riccati<double, LQG_INST_RICCATI>(const lqg_problem*, lqg_view, lqg_view, lqg_view, void*,
                                                             long, hipStream_t);
#endif
#endif
#ifdef LQG_INST_KALMAN
#ifdef LQG_INST_F32
template hipError_t launch_kalman<float, LQG_INST_KALMAN>(const lqg_problem*, lqg_view, hipStream_t);
#endif
#ifdef LQG_INST_F64
template hipError_t launch_kalman<double, LQG_INST_KALMAN>(const lqg_problem*, lqg_view, hipStream_t);
#endif
#endif
#ifdef LQG_INST_FORWARD
// -DLQG_INST_VARIANT=0..3 (= 2*FUSED + TI) selects which quarter of the k_forward variants this unit compiles; unit 3
// also holds the dispatching launch_forward (all quarters declared extern so that it does not instantiate them again)
#define LQG_FWD_V(R_, F_, T_)                                                                                 \
  launch_forward_v<R_, LQG_INST_FORWARD, F_, T_>(const lqg::ForwardArgs<R_>&, long, bool, hipStream_t)
#define LQG_FWD_EXTERN(R_)                                                                                    \
  extern template hipError_t LQG_FWD_V(R_, false, false);                                                     \
  extern template hipError_t LQG_FWD_V(R_, false, true);                                                      \
  extern template hipError_t LQG_FWD_V(R_, true, false);                                                      \
  extern template hipError_t LQG_FWD_V(R_, true, true);
#if LQG_INST_VARIANT == 3
#define LQG_FWD_UNIT(R_)                                                                                      \
  LQG_FWD_EXTERN(R_)                                                                                          \
  template hipError_t LQG_FWD_V(R_, true, true);                                                              \
  template hipError_t launch_forward<R_, LQG_INST_FORWARD>(const lqg_problem*, const void*, long, bool,       \
                                                           lqg_traj, void*, long, void*, lqg_view, lqg_traj,  \
                                                           lqg_view, hipStream_t);
#elif LQG_INST_VARIANT == 2
#define LQG_FWD_UNIT(R_) template hipError_t LQG_FWD_V(R_, true, false);
#elif LQG_INST_VARIANT == 1
#define LQG_FWD_UNIT(R_) template hipError_t LQG_FWD_V(R_, false, true);
#else
#define LQG_FWD_UNIT(R_) template hipError_t LQG_FWD_V(R_, false, false);
#endif
#ifdef LQG_INST_F32
LQG_FWD_UNIT(float)
#endif
#ifdef LQG_INST_F64
LQG_FWD_UNIT(double)
// the mixed-precision operator-stream variants (LQG_F32_SYS64) ride in the fp64 units of the same FUSED = false quarters
#define LQG_FWD_OPS32_V(T_) \
  launch_forward_ops32_v<LQG_INST_FORWARD, T_>(const lqg::ForwardArgs<double>&, long, hipStream_t)
#if LQG_INST_VARIANT == 3
extern template hipError_t LQG_FWD_OPS32_V(false);
extern template hipError_t LQG_FWD_OPS32_V(true);
template hipError_t launch_forward_ops32<LQG_INST_FORWARD>(const lqg_problem*, const void*, long, void*, hipStream_t);
#elif LQG_INST_VARIANT == 1
template hipError_t LQG_FWD_OPS32_V(true);
#elif LQG_INST_VARIANT == 0
template hipError_t LQG_FWD_OPS32_V(false);
#endif
#endif
#endif
#ifdef LQG_INST_TRIAL
#ifdef LQG_INST_F32
template hipError_t launch_trial<float, LQG_INST_TRIAL>(const lqg_problem*, const void*, lqg_traj, lqg_traj, void*,
                                                        long, long, hipStream_t);
#endif
#ifdef LQG_INST_F64
template hipError_t launch_trial<double, LQG_INST_TRIAL>(const lqg_problem*, const void*, lqg_traj, lqg_traj, void*,
                                                         long, long, hipStream_t);
#endif
#endif
#ifdef LQG_INST_SIM
#ifdef LQG_INST_F32
template hipError_t launch_simulate<float, LQG_INST_SIM>(const lqg_problem*, lqg_view, lqg_view, lqg_view, lqg_traj,
                                                         lqg_traj, lqg_view, lqg_view, lqg_traj, lqg_traj, lqg_traj,
                                                         lqg_traj, hipStream_t, unsigned long long);
#endif
#ifdef LQG_INST_F64
template hipError_t launch_simulate<double, LQG_INST_SIM>(const lqg_problem*, lqg_view, lqg_view, lqg_view, lqg_traj,
                                                          lqg_traj, lqg_view, lqg_view, lqg_traj, lqg_traj, lqg_traj,
                                                          lqg_traj, hipStream_t, unsigned long long);
#endif
#endif
}  // namespace host
}  // namespace lqg
