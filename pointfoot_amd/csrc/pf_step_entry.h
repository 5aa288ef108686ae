// pf_step_entry.h -- the env-step kernel's entry point (the __global__ wrapper around PfKernel::run) and the kernel instances the
// library is built with.  Included by pf_env.hip (host side of the C ABI + the PointFoot instance) and pf_env_quad.hip (the
// quadruped instances, explicit instantiations: a translation unit of their own because they are compiled with another scheduler).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/pointfoot_hip.h"
#include "pf_layout.h"
#include "pf_step_kernel.h"

// dynamics on the models with their feet lumped into the shanks: 9 -> 7 bodies (PointFoot), 17 -> 13 (quadrupeds)
constexpr int PF_NBD_POINTFOOT = 7, PF_NBD_QUAD = 13;
// PointFoot instance: two wavefronts per pair of robots (PfKernel DUAL: the contact side of every substep, the observation noise
// and the height scan run on a partner wavefront beside the dynamics).  94.8 -> 89.2 us per 4096-env launch on the same box;
// -DPF_POINT_DUAL=0 builds the single-wavefront form (tools/build_variant.sh for A/B runs).
#ifndef PF_POINT_DUAL
#define PF_POINT_DUAL 1
#endif
using PfKernelPoint = PfKernel<9, 6, 2, PF_KIND_POINTFOOT, PF_KSPH, 2, 16, PF_NBD_POINTFOOT, PF_POINT_DUAL != 0>;
// quadruped instances: robots (= wavefronts) per workgroup.  A workgroup's LDS is released when its LAST wavefront ends, so the
// next workgroup of a CU waits for the slowest robot of the one before: 8 per workgroup = one workgroup per CU, 4 = two
// (measured, anymal_c_rough, 8192 envs, same box: 8 -> 538 us, 4 -> 520 us, 2 -> 549 us: every workgroup stages the read-only block).
#ifndef PF_QUAD_WPB
#define PF_QUAD_WPB 4
#endif
using PfKernelQuad = PfKernel<17, 12, 4, PF_KIND_LEGGED_ROBOT, PF_KSPH_QUAD, 1, 0, PF_NBD_QUAD>;
using PfKernelQuadSelf = PfKernel<17, 12, 4, PF_KIND_LEGGED_ROBOT, PF_KSPH_QUAD, 1, PF_KPAIR_QUAD, PF_NBD_QUAD>;
// One wavefront (= one 64-thread workgroup) per robot.  LDS per workgroup is sizeof(Shared); the
// grid is num_envs workgroups, dealt round-robin over the 8 XCDs by the dispatcher; every env only
// touches its own record, so no inter-workgroup ordering is needed.
// model / cfg are separate __restrict__ const kernel arguments so the compiler can prove they are invariant and
// read them with scalar loads (s_load) into SGPRs instead of holding hundreds of config values in VGPRs.
// WPB wavefronts per workgroup (each with RPW robots) share one read-only block (model, actuator parameters) in LDS; the
// rest of the LDS is per robot.  WAVES = waves per SIMD the register budget is sized for.  LDS comes as dynamic shared
// memory (the quadruped workgroup of 8 robots needs ~145 KB, above the static limit).
// DUAL: a workgroup is TWO wavefronts for the same RPW robots (WPB == 1): wave 0 runs the step, wave 1 the contact side of its
// substeps (PfKernel::run, role).
template <int NB, int ND, int NF, int WAVES, int KIND = PF_KIND_POINTFOOT, int KS = PF_KSPH, int RPW = 1, int WPB = 1, int KPR = (KIND == PF_KIND_POINTFOOT ? 16 : 0), int NBD = NB, bool DUAL = false>
__global__ __launch_bounds__(64 * WPB * (DUAL ? 2 : 1), (WPB == 1 || DUAL) ? WAVES : (WPB <= 4 ? 2 : 1)) void pf_step_kernel(const void* __restrict__ packed, const PfConfig* __restrict__ cfg, PfStepArgs g, const PfPolicyHeads heads) {
  using K = PfKernel<NB, ND, NF, KIND, KS, RPW, KPR, NBD, DUAL>;
  static_assert(!DUAL || WPB == 1, "a DUAL workgroup is one pair of wavefronts");
  extern __shared__ __attribute__((aligned(16))) unsigned char pf_lds[];
  typename K::SharedRO& ro = *reinterpret_cast<typename K::SharedRO*>(pf_lds);
  typename K::Shared* sh = reinterpret_cast<typename K::Shared*>(pf_lds + ((sizeof(typename K::SharedRO) + 15) & ~size_t(15)));
  g.packed = packed; g.cfg = cfg;
  // the slot the *next* launch accumulates into is cleared here, so a rollout needs no memset between steps
  if (blockIdx.x == 0 && g.stats_next && threadIdx.x < PF_STATS_FLOATS) g.stats_next[threadIdx.x] = 0.0f;
  // each group of 64 / RPW lanes runs the phase program on its own robot and its own LDS block
  const int tid = DUAL ? ((int)threadIdx.x & 63) : (int)threadIdx.x;
  const int slot = tid / (64 / RPW);                         // robot slot inside the workgroup
  const int env = (int)blockIdx.x * (RPW * WPB) + slot;
  if (env >= cfg->num_envs) return;
  K::run(sh[slot], ro, g, env, DUAL ? ((int)threadIdx.x >> 6) : 0, sh, slot, &heads);
}
template <class K, int RPW, int WPB>
static constexpr size_t pf_lds_bytes() {
  return ((sizeof(typename K::SharedRO) + 15) & ~size_t(15)) + sizeof(typename K::Shared) * RPW * WPB;
}

