// pf_env_quad.hip -- the quadruped instances of the env-step kernel (LeggedRobot / Anymal: 17 bodies, 12 DOF, 4 feet; with and
// without leg-leg self-collision), instantiated here and declared `extern template` in pf_env.hip, which launches them.
// Compiled with -mllvm -amdgpu-sched-strategy=iterative-ilp (515 vs 535 us per 8192-env launch, anymal_c_rough); the PointFoot
// instance of pf_env.hip is compiled with the default scheduler (zero register spills at its 128 VGPRs).
#include "pf_step_entry.h"

template __global__ void pf_step_kernel<17, 12, 4, 2, PF_KIND_LEGGED_ROBOT, PF_KSPH_QUAD, 1, PF_QUAD_WPB, PF_KPAIR_QUAD, PF_NBD_QUAD, false>(const void* __restrict__, const PfConfig* __restrict__, PfStepArgs, const PfPolicyHeads);
template __global__ void pf_step_kernel<17, 12, 4, 2, PF_KIND_LEGGED_ROBOT, PF_KSPH_QUAD, 1, PF_QUAD_WPB, 0, PF_NBD_QUAD, false>(const void* __restrict__, const PfConfig* __restrict__, PfStepArgs, const PfPolicyHeads);
