// lqg_inst.hip — explicit instantiation of ONE launcher (both dtypes) per translation unit, selected with
//   -DLQG_INST_RICCATI="b,u" | -DLQG_INST_KALMAN="b,y" | -DLQG_INST_FORWARD="x,b,u,y,d" |
//   -DLQG_INST_TRIAL="m,d" | -DLQG_INST_SIM="x,b,u,y"
// so that lqg_amd/build.py can compile the (large, fully unrolled) kernels in parallel.
#include "lqg_launch.hpp"

namespace lqg {
namespace host {
#ifdef LQG_INST_RICCATI
template hipError_t launch_riccati<float, LQG_INST_RICCATI>(const lqg_problem*, lqg_view, lqg_view, lqg_view, void*,
                                                            long, hipStream_t);
template hipError_t launch_riccati<double, LQG_INST_RICCATI>(const lqg_problem*, lqg_view, lqg_view, lqg_view, void*,
                                                             long, hipStream_t);
#endif
#ifdef LQG_INST_KALMAN
template hipError_t launch_kalman<float, LQG_INST_KALMAN>(const lqg_problem*, lqg_view, hipStream_t);
template hipError_t launch_kalman<double, LQG_INST_KALMAN>(const lqg_problem*, lqg_view, hipStream_t);
#endif
#ifdef LQG_INST_FORWARD
template hipError_t launch_forward<float, LQG_INST_FORWARD>(const lqg_problem*, const void*, long, bool, lqg_traj,
                                                            void*, long, void*, lqg_view, lqg_traj, lqg_view,
                                                            hipStream_t);
template hipError_t launch_forward<double, LQG_INST_FORWARD>(const lqg_problem*, const void*, long, bool, lqg_traj,
                                                             void*, long, void*, lqg_view, lqg_traj, lqg_view,
                                                             hipStream_t);
#endif
#ifdef LQG_INST_TRIAL
template hipError_t launch_trial<float, LQG_INST_TRIAL>(const lqg_problem*, const void*, lqg_traj, lqg_traj, void*,
                                                        long, long, hipStream_t);
template hipError_t launch_trial<double, LQG_INST_TRIAL>(const lqg_problem*, const void*, lqg_traj, lqg_traj, void*,
                                                         long, long, hipStream_t);
#endif
#ifdef LQG_INST_SIM
template hipError_t launch_simulate<float, LQG_INST_SIM>(const lqg_problem*, lqg_view, lqg_view, lqg_view, lqg_traj,
                                                         lqg_traj, lqg_view, lqg_view, lqg_traj, lqg_traj, lqg_traj,
                                                         lqg_traj, hipStream_t);
template hipError_t launch_simulate<double, LQG_INST_SIM>(const lqg_problem*, lqg_view, lqg_view, lqg_view, lqg_traj,
                                                          lqg_traj, lqg_view, lqg_view, lqg_traj, lqg_traj, lqg_traj,
                                                          lqg_traj, hipStream_t);
#endif
}  // namespace host
}  // namespace lqg
