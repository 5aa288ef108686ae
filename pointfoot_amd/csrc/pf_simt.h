// pf_simt.h — the small SIMT vocabulary the env-step kernel is written in.
//
// One wavefront (64 lanes) owns one robot, or each 32-lane half owns one (PF_W).  The kernel is a *phase program*:
// lane-strided loops (PF_LANES) that touch only their own items, separated by PF_SYNC(), plus "uniform" sections that
// every lane of a robot executes redundantly (branches on env state are uniform across a robot's lanes; two robots in one
// wave diverge under the EXEC mask).  Per-robot data lives in LDS; a workgroup may hold several wavefronts, which share
// only a read-only block, so PF_SYNC() is wave-scope.
//
// With -DPF_HOST_EMU (used ONLY by tests/, never by the product library) the same program compiles
// as plain serial C++: PF_LANES becomes an ordinary loop and barriers vanish.  That build checks
// the kernel's arithmetic and indexing on a machine without a GPU; it cannot detect missing barriers.
#pragma once

#if defined(PF_HOST_EMU)
#include <cmath>
#include <cstdint>
#include <cstring>
#define PF_DEV static inline
#define PF_LANES(i, n) for (int i = 0; i < (n); ++i)
#define PF_SYNC() ((void)0)
#define PF_XSYNC() ((void)0)
#define PF_HANDOFF_SET(flag, token) ((void)0)
#define PF_HANDOFF_WAIT(flag, token) ((void)0)
#define PF_ONE if (true)
#define PF_WAVE_SUM(x) (x)
#define PF_WAVE_MAX(x) (x)
#define PF_ATOMIC_ADD(p, v) (*(p) += (v))
#define PF_NOCONTRACT
#define PF_LANE_ID 0
#else
#include <hip/hip_runtime.h>
#include <stdint.h>
#define PF_DEV static __device__ __forceinline__
// PF_W = lanes per robot: a constexpr in scope at every use (PfKernel::PF_W).  64: one robot per wavefront; 32: two robots
// per wavefront, one per half -- uniform sections and loops over <= 32 items then cost one instruction for two robots.
// PF_OPAQUE_LANE (a constexpr in scope at every use, PfKernel::PF_OPAQUE_LANE): the lane id is re-materialised behind an empty
// volatile asm at every use, so nothing derived from it (LDS addresses, lane == k masks) can be hoisted out of the substep loop
// and kept alive across it.  The quadruped kernel hoisted so much of that above its loop that 41 VGPRs spilled to scratch
// (10 KB of scratch writes per robot and step: two thirds of the kernel's HBM write traffic).
template <bool OPQ>
PF_DEV int pf_thread_id() {
  int t = (int)threadIdx.x;
  if constexpr (OPQ) asm volatile("" : "+v"(t));
  return t;
}
#define PF_LANE_ID (pf_thread_id<PF_OPAQUE_LANE>() & (PF_W - 1))
#define PF_LANES(i, n) for (int i = PF_LANE_ID; i < (n); i += PF_W)
// One wavefront (or half of one) per robot, possibly several wavefronts per workgroup that share only read-only data: a
// "barrier" has to order just the wave's own LDS traffic.  The hardware executes a wave's LDS instructions in order, which
// leaves the compiler as the only thing to restrain -- a wavefront-scope fence pair around a wave barrier.  Robots in
// one workgroup may therefore diverge (a reset takes extra phases) without any workgroup barrier to mismatch.
PF_DEV void pf_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
#define PF_SYNC() pf_wave_sync()
// DUAL kernel instances (pf_step_kernel.h): the two wavefronts that share a group of robots meet at a workgroup barrier; LDS
// written before it by one is visible to the other after it
PF_DEV void pf_group_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
#define PF_XSYNC() pf_group_sync()
// One-way hand-off between the two wavefronts of a DUAL group without a barrier: the producer's LDS writes, then (LDS operations
// of a wavefront execute in order) a token in a flag word; the consumer polls the flag before it reads.  The step clears the flag before its first
// substep and every substep has a token of its own, so a stale flag never matches.
PF_DEV void pf_handoff_set(int* flag, int token, bool one) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  if (one) __hip_atomic_store(flag, token, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
PF_DEV void pf_handoff_wait(int* flag, int token) {
  while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != token) __builtin_amdgcn_s_sleep(2);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
#define PF_HANDOFF_SET(flag, token) pf_handoff_set(&(flag), (token), PF_LANE_ID == 0)
#define PF_HANDOFF_WAIT(flag, token) pf_handoff_wait(&(flag), (token))
#define PF_ONE if (PF_LANE_ID == 0)
template <int W>
PF_DEV float pf_wave_sum(float v) {
#pragma unroll
  for (int o = W / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
#define PF_WAVE_SUM(x) pf_wave_sum<PF_W>(x)
// sum over the 16 lanes of a DPP row (quad swaps, half-row mirror, row mirror): four v_add_f32_dpp, and every lane ends
// with the same bits because each lane adds the same pairs at every level
template <int CTRL>
PF_DEV float pf_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// sum over a 32-lane group (two DPP rows): row sums, then the neighbouring row's total through ds_swizzle (lane ^ 16);
// a + b == b + a bit for bit, so both rows end with the same value
PF_DEV float pf_row16_sum(float v);
PF_DEV float pf_group32_sum(float v) {
  v = pf_row16_sum(v);
  return v + __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F));
}
PF_DEV float pf_row16_sum(float v) {
  v += pf_dpp<0xB1>(v);     // quad_perm [1,0,3,2]
  v += pf_dpp<0x4E>(v);     // quad_perm [2,3,0,1]
  v += pf_dpp<0x141>(v);    // row_half_mirror
  v += pf_dpp<0x140>(v);    // row_mirror
  return v;
}
#define PF_ATOMIC_ADD(p, v) atomicAdd((p), (v))
#endif
