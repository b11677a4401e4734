// lqg_adjoint_inst.hip — explicit instantiation of the adjoint launcher for ONE shape and dtype per translation
// unit:  -DLQG_INST_ADJOINT="x,b,u,y,d"  with -DLQG_INST_F32 or -DLQG_INST_F64  (lqg_amd/build.py).
#include "lqg_adjoint_launch.hpp"

namespace lqg {
namespace host {
#ifdef LQG_INST_F32
template hipError_t launch_adjoint<float, LQG_INST_ADJOINT>(const lqg_problem*, lqg_traj, const void*, long, long,
                                                            void*, long, long, void*, long, void*, int, hipStream_t);
#endif
#ifdef LQG_INST_F64
template hipError_t launch_adjoint<double, LQG_INST_ADJOINT>(const lqg_problem*, lqg_traj, const void*, long, long,
                                                             void*, long, long, void*, long, void*, int, hipStream_t);
#endif
}  // namespace host
}  // namespace lqg
