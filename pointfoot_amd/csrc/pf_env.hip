// pf_env.hip — launch side of the fused env-step kernel + the pf_* half of the C ABI.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/pointfoot_hip.h"
#include "../../include/pointfoot_hip_dev.h"
#include "pf_layout.h"
#include "pf_step_entry.h"

// The quadruped instances are compiled in a translation unit of their own (pf_env_quad.hip: they want the iterative-ilp scheduler,
// the PointFoot instance -- 128 VGPRs -- the default one: no register spills, no scratch traffic).
extern template __global__ void pf_step_kernel<17, 12, 4, 2, PF_KIND_LEGGED_ROBOT, PF_KSPH_QUAD, 1, PF_QUAD_WPB, PF_KPAIR_QUAD, PF_NBD_QUAD, false>(const void* __restrict__, const PfConfig* __restrict__, PfStepArgs, const PfPolicyHeads);
extern template __global__ void pf_step_kernel<17, 12, 4, 2, PF_KIND_LEGGED_ROBOT, PF_KSPH_QUAD, 1, PF_QUAD_WPB, 0, PF_NBD_QUAD, false>(const void* __restrict__, const PfConfig* __restrict__, PfStepArgs, const PfPolicyHeads);

thread_local std::string g_pf_error;
void pf_set_error(const std::string& s) { g_pf_error = s; }

#define PF_HIP_OK(expr)                                                                    \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess) {                                                                \
      pf_set_error(std::string(#expr) + ": " + hipGetErrorString(e_));                     \
      return 1;                                                                            \
    }                                                                                      \
  } while (0)

struct PfEnv {
  PfModel model;
  PfConfig cfg;
  PfLayout layout;
  PfBuffers bufs;
  void* d_packed = nullptr;       // the model the kernel integrates -- fixed-joint bodies lumped into their parents (pf_lump.h) -- packed
                                  // in the instance's LDS layout (PfKernel::pack_model): a launch stages it with one flat copy
  PfConfig* d_cfg = nullptr;
  int device = 0;
  bool bound = false;
  int instance = 0;
  unsigned* d_stamps = nullptr;   // diagnostic build (-DPF_STAMPS) only
  float* stats_next = nullptr;    // pf_set_reset_stats_next: slot the following step will use
  float* stats_cleared = nullptr; // slot already cleared on the device by the previous launch
};

// development hook: a read stream with the step kernel's record access pattern (each 32-lane group walks a 640-byte row, one
// dword per lane) over a buffer of known size -- the byte count FETCH_SIZE is calibrated against (tools/calib_fetch.py)
__global__ void pf_calib_read_kernel(const float* __restrict__ src, int rows, int row_floats, float* __restrict__ sink) {
  const int row = (int)blockIdx.x * 2 + ((int)threadIdx.x >> 5), lane = (int)threadIdx.x & 31;
  if (row >= rows) return;
  float acc = 0.0f;
  for (int i = lane; i < row_floats; i += 32) acc += src[(size_t)row * row_floats + i];
  if (acc == 1234.5678f) sink[0] = acc;      // keeps the loads alive without a store stream
}

// development hook: the height scan's gather pattern (_get_heights: 11 x 11 points, three int16 samples per point -- the cell's low
// corner, its +x and its +y neighbour -- one point per lane of a 32-lane group) on a field of known size at host-chosen base cells,
// so that the 64-byte sectors / 128-byte lines a launch touches are known exactly (tools/calib_gather.py)
__global__ void pf_calib_gather_kernel(const int16_t* __restrict__ field, int cols, const int* __restrict__ base, int robots, float* __restrict__ sink) {
  const int r = (int)blockIdx.x * 2 + ((int)threadIdx.x >> 5), lane = (int)threadIdx.x & 31;
  if (r >= robots) return;
  const int bx = base[2 * r], by = base[2 * r + 1];
  int acc = 0;
  for (int p = lane; p < 121; p += 32) {
    const int ix = p / 11, iy = p - 11 * ix;
    const size_t at = (size_t)(bx + ix) * cols + (by + iy);
    const int h0 = field[at], h1 = field[at + cols], h2 = field[at + 1];
    acc += min(h0, min(h1, h2));
  }
  if (acc == 0x7fffffff) sink[0] = (float)acc;
}

extern "C" {

int pf_debug_calib_gather(const int16_t* d_field, int cols, const int* d_base, int robots, float* d_sink, void* stream) {
  hipLaunchKernelGGL(pf_calib_gather_kernel, dim3((robots + 1) / 2), dim3(64), 0, (hipStream_t)stream, d_field, cols, d_base, robots, d_sink);
  PF_HIP_OK(hipGetLastError());
  return 0;
}

int pf_debug_calib_read(const float* d_src, int rows, int row_floats, float* d_sink, void* stream) {
  hipLaunchKernelGGL(pf_calib_read_kernel, dim3((rows + 1) / 2), dim3(64), 0, (hipStream_t)stream, d_src, rows, row_floats, d_sink);
  PF_HIP_OK(hipGetLastError());
  return 0;
}

const char* pf_last_error(void) { return g_pf_error.c_str(); }

void pf_abi_sizes(int64_t* sizes, int32_t n) {
  const int64_t v[8] = {sizeof(PfModel), sizeof(PfConfig), sizeof(PfLayout), sizeof(PfBuffers), sizeof(PfInject),
                        sizeof(PpoConfig), sizeof(PpoBuffers), sizeof(PfPolicyHeads)};
  for (int i = 0; i < n && i < 8; ++i) sizes[i] = v[i];
}

int pf_layout(const PfModel* model, const PfConfig* cfg, PfLayout* out) {
  if (!model || !cfg || !out) { pf_set_error("pf_layout: null argument"); return 1; }
  pf_compute_layout(model, cfg, out);
  return 0;
}

// kernel instances built: 0 = PointFoot PF_P441A (9 bodies / 6 dofs / 2 feet, self-collision built in); LeggedRobot quadrupeds
// with 3-DOF legs and separate feet (17 / 12 / 4: ANYmal-C, ANYmal-B, A1): 1 = built without self-collision (anymal_c_rough,
// BASELINE config 5), 2 = with it (models that come with capsule pairs: anymal_c_flat)
static int pf_instance(const PfModel* m, const PfConfig* c) {
  if (c->env_kind == PF_KIND_POINTFOOT && m->nb == 9 && m->nd == 6 && m->nfeet == 2) return 0;
  if (c->env_kind == PF_KIND_LEGGED_ROBOT && m->nb == 17 && m->nd == 12 && m->nfeet == 4) return (c->self_collisions && m->npair > 0) ? 2 : 1;
  return -1;
}
static int pf_validate(const PfModel* m, const PfConfig* c) {
  int inst = pf_instance(m, c);
  if (inst < 0) {
    pf_set_error("pf_create: no kernel instance for this model / env kind (built: PointFoot 9 bodies/6 dofs/2 feet, LeggedRobot 17/12/4)");
    return 1;
  }
  int want_obs = inst == 0 ? 9 + 3 * m->nd : 12 + 3 * m->nd + c->num_height_points;
  // PointFoot with cfg.terrain.measure_heights_actor: the height measurements follow the proprioceptive terms, and a privileged
  // buffer, if there is one, has the same layout (the reference adds the actor's noise buffer to its front)
  if (inst == 0 && c->num_obs == want_obs + c->num_height_points && c->measure_heights &&
      (c->num_priv_obs == 0 || c->num_priv_obs == c->num_obs)) want_obs = c->num_obs;
  if (m->nsph > (inst == 0 ? PF_KSPH : PF_KSPH_QUAD) || c->num_obs > PF_MAX_OBS || c->num_height_points > PF_MAX_HPTS ||
      c->num_rewards >= PF_MAX_REWARDS || c->num_actions != m->nd || c->num_obs != want_obs) {
    pf_set_error("pf_create: configuration outside the compiled limits");
    return 1;
  }
  if (m->ncap < 0 || m->npair < 0 || m->ncap > PF_MAX_CAPS || m->npair > PF_MAX_PAIRS ||
      (inst == 0 && (m->ncap > PfKernelPoint::KC || m->npair > PfKernelPoint::KP)) ||
      (inst == 1 && c->self_collisions) ||
      (inst == 2 && (m->ncap > PfKernelQuadSelf::KC || m->npair > PfKernelQuadSelf::KP))) {
    pf_set_error("pf_create: self-collision geometry outside the compiled limits (PointFoot: 8 capsules / 16 pairs; LeggedRobot: 16 / 56; "
                 "self_collisions set needs capsule pairs in the model)");
    return 1;
  }
  for (int p = 0; p < m->npair; ++p)
    if (m->pair_a[p] < 0 || m->pair_a[p] >= m->ncap || m->pair_b[p] < 0 || m->pair_b[p] >= m->ncap) { pf_set_error("pf_create: capsule pair index out of range"); return 1; }
  for (int k = 0; k < m->ncap; ++k)
    if (m->cap_s0[k] < 0 || m->cap_s0[k] >= m->nsph || m->cap_s1[k] < 0 || m->cap_s1[k] >= m->nsph) { pf_set_error("pf_create: capsule sphere index out of range"); return 1; }
  if (2 * m->nsph + m->npair > 128) { pf_set_error("pf_create: more than 128 contact candidates"); return 1; }
  if (inst == 1 && c->num_priv_obs != 0) { pf_set_error("pf_create: LeggedRobot has no privileged observations"); return 1; }
  for (int i = 1; i < m->nb; ++i)
    if (m->parent[i] < 0 || m->parent[i] >= i) { pf_set_error("pf_create: bodies must be in depth-first order"); return 1; }
  return 0;
}

PfEnv* pf_create(const PfModel* model, const PfConfig* cfg, int32_t device) {
  if (!model || !cfg) { pf_set_error("pf_create: null argument"); return nullptr; }
  if (pf_validate(model, cfg)) return nullptr;
  PfEnv* e = new PfEnv();
  e->model = *model; e->cfg = *cfg; e->device = device;
  pf_compute_layout(model, cfg, &e->layout);
  e->instance = pf_instance(model, cfg);
  bool lay_ok = e->instance == 0 ? PfKernelPoint::layout_matches(e->layout)
              : e->instance == 1 ? PfKernelQuad::layout_matches(e->layout)
                                 : PfKernelQuadSelf::layout_matches(e->layout);
  if (!lay_ok) { pf_set_error("pf_create: record layout differs from the kernel's compile-time layout"); delete e; return nullptr; }
  PfModel lumped; PfLump lump;
  pf_lump_model(model, &lumped, &lump);
  if (lumped.nb != (e->instance == 0 ? PF_NBD_POINTFOOT : PF_NBD_QUAD)) {
    pf_set_error("pf_create: the model does not reduce to the kernel instance's body count (PointFoot: 9 bodies with 2 fixed feet; LeggedRobot: 17 with 4)");
    delete e; return nullptr;
  }
  std::vector<unsigned char> img;
  auto pack = [&](auto* kernel_type) {
    using K = std::remove_pointer_t<decltype(kernel_type)>;
    img.resize(sizeof(typename K::ModelS));
    K::pack_model(lumped, lump, *reinterpret_cast<typename K::ModelS*>(img.data()));
  };
  if (e->instance == 0) pack((PfKernelPoint*)nullptr); else if (e->instance == 1) pack((PfKernelQuad*)nullptr); else pack((PfKernelQuadSelf*)nullptr);
  if (hipSetDevice(device) != hipSuccess || hipMalloc(&e->d_packed, img.size()) != hipSuccess ||
      hipMalloc(&e->d_cfg, sizeof(PfConfig)) != hipSuccess ||
      hipMemcpy(e->d_packed, img.data(), img.size(), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(e->d_cfg, cfg, sizeof(PfConfig), hipMemcpyHostToDevice) != hipSuccess) {
    pf_set_error("pf_create: HIP device allocation failed (is a GPU visible?)");
    delete e;
    return nullptr;
  }
  return e;
}

void pf_destroy(PfEnv* e) {
  if (!e) return;
  if (e->d_packed) (void)hipFree(e->d_packed);
  if (e->d_cfg) (void)hipFree(e->d_cfg);
  delete e;
}

int pf_bind_buffers(PfEnv* e, const PfBuffers* b) {
  if (!e || !b) { pf_set_error("pf_bind_buffers: null argument"); return 1; }
  if (!b->d_state || !b->d_episode_length || !b->d_reset || !b->d_time_out || !b->d_terminate || !b->d_obs || !b->d_rew) {
    pf_set_error("pf_bind_buffers: a required buffer is NULL");
    return 1;
  }
  if (e->cfg.num_priv_obs > 0 && !b->d_priv) { pf_set_error("pf_bind_buffers: d_priv required"); return 1; }
  if (e->cfg.terrain_type == PF_TERRAIN_HEIGHTFIELD && (!b->d_height_samples || e->cfg.hf_rows < 2 || e->cfg.hf_cols < 2)) {
    pf_set_error("pf_bind_buffers: height field required for this terrain type"); return 1;
  }
  if (e->cfg.env_kind == PF_KIND_LEGGED_ROBOT && e->cfg.use_actuator_net && !b->d_actuator_params) {
    pf_set_error("pf_bind_buffers: actuator network parameters required (use_actuator_net)"); return 1;
  }
  if (e->cfg.curriculum && (!b->d_terrain_level || !b->d_terrain_type || !b->d_terrain_origins)) {
    pf_set_error("pf_bind_buffers: terrain curriculum buffers required"); return 1;
  }
  e->bufs = *b; e->bound = true;
  return 0;
}

int pf_set_reset_stats(PfEnv* e, float* d_reset_stats) {
  if (!e || !e->bound) { pf_set_error("pf_set_reset_stats: buffers not bound"); return 1; }
  e->bufs.d_reset_stats = d_reset_stats;
  return 0;
}

int pf_set_reset_stats_next(PfEnv* e, float* d_next) {
  if (!e || !e->bound) { pf_set_error("pf_set_reset_stats_next: buffers not bound"); return 1; }
  e->stats_next = d_next;
  return 0;
}

/* development hook (include/pointfoot_hip_dev.h), diagnostic build only: per-env phase cycle counters (PF_NSTAMP unsigned per env) */
int pf_debug_set_stamps(PfEnv* e, unsigned* d_stamps) { if (!e) return 1; e->d_stamps = d_stamps; return 0; }

int pf_update_config(PfEnv* e, const PfConfig* cfg) {
  if (!e || !cfg) { pf_set_error("pf_update_config: null argument"); return 1; }
  if (pf_validate(&e->model, cfg)) return 1;
  if (cfg->num_envs != e->cfg.num_envs || cfg->num_rewards != e->cfg.num_rewards || cfg->num_obs != e->cfg.num_obs ||
      cfg->num_height_points != e->cfg.num_height_points) {
    pf_set_error("pf_update_config: shape-changing fields cannot be updated"); return 1;
  }
  e->cfg = *cfg;
  PF_HIP_OK(hipMemcpy(e->d_cfg, cfg, sizeof(PfConfig), hipMemcpyHostToDevice));
  return 0;
}

static int pf_launch(PfEnv* e, const float* d_actions, int64_t step_counter, const PfInject* inject, void* stream, int mode, const uint8_t* d_mask = nullptr,
                     const PfPolicyHeads* heads = nullptr) {
  if (!e || !e->bound) { pf_set_error("pf_step: buffers not bound"); return 1; }
  if (mode == 0 && !d_actions && !heads) { pf_set_error("pf_step: actions pointer is NULL"); return 1; }
  { int cur = -1; if (hipGetDevice(&cur) != hipSuccess || cur != e->device) { pf_set_error("pf_step: the calling thread's current HIP device is not the env's device (torch.cuda.set_device)"); return 1; } }
  PfStepArgs g;
  g.packed = e->d_packed; g.cfg = e->d_cfg; g.L = e->layout; g.b = e->bufs;
  if (inject) g.inj = *inject; else std::memset(&g.inj, 0, sizeof(g.inj));
  if (g.inj.d_root && !(g.inj.d_dof_pre && g.inj.d_dof_post && g.inj.d_contact && g.inj.d_feet_pos)) {
    pf_set_error("pf_step: incomplete injected sim state"); return 1;
  }
  g.actions = d_actions; g.step_counter = step_counter; g.mode = mode; g.stamps = e->d_stamps; g.reset_mask = d_mask;
  PfPolicyHeads hd;      // a kernel argument of its own (pf_step_entry.h); all zero: plain pf_step
  if (heads) hd = *heads; else std::memset(&hd, 0, sizeof(hd));
  hipStream_t s = (hipStream_t)stream;
  g.stats_next = nullptr;
  if (e->bufs.d_reset_stats && mode < 2) {
    // cleared by the previous launch when the caller announced it with pf_set_reset_stats_next; otherwise clear it now
    if (e->stats_cleared != e->bufs.d_reset_stats) PF_HIP_OK(hipMemsetAsync(e->bufs.d_reset_stats, 0, sizeof(float) * PF_STATS_FLOATS, s));
    g.stats_next = (e->stats_next && e->stats_next != e->bufs.d_reset_stats) ? e->stats_next : nullptr;
    e->stats_cleared = g.stats_next;
  }
  const int N = e->cfg.num_envs;
  // Three kernel instances (the quadruped with and without self-collision).  PointFoot: two robots per wavefront (one per 32-lane half), one wavefront per workgroup.
  // LeggedRobot: one robot per wavefront, 8 wavefronts per workgroup sharing the read-only LDS block.  (Measured and
  // dropped: one or four PointFoot robots per wave, 3 / 4 waves per SIMD register budgets -- all slower or spilling;
  // two quadrupeds per wave: 0.843 vs 0.834 ms.)
  auto launch = [&](auto kern, size_t lds, int robots_per_block, int threads) {
    static bool raised = false;          // one kernel per env handle in practice; raising the limit again is harmless
    if (lds > 64 * 1024 && !raised) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); raised = true; }
    hipLaunchKernelGGL(kern, dim3((N + robots_per_block - 1) / robots_per_block), dim3(threads), lds, s, e->d_packed, e->d_cfg, g, hd);
  };
  if (e->instance == 2) launch(pf_step_kernel<17, 12, 4, 2, PF_KIND_LEGGED_ROBOT, PF_KSPH_QUAD, 1, PF_QUAD_WPB, PF_KPAIR_QUAD, PF_NBD_QUAD>, pf_lds_bytes<PfKernelQuadSelf, 1, PF_QUAD_WPB>(), PF_QUAD_WPB, 64 * PF_QUAD_WPB);
  else if (e->instance == 1) launch(pf_step_kernel<17, 12, 4, 2, PF_KIND_LEGGED_ROBOT, PF_KSPH_QUAD, 1, PF_QUAD_WPB, 0, PF_NBD_QUAD>, pf_lds_bytes<PfKernelQuad, 1, PF_QUAD_WPB>(), PF_QUAD_WPB, 64 * PF_QUAD_WPB);
  else if constexpr (PF_POINT_DUAL != 0) launch(pf_step_kernel<9, 6, 2, 4, PF_KIND_POINTFOOT, PF_KSPH, 2, 1, 16, PF_NBD_POINTFOOT, PF_POINT_DUAL != 0>, pf_lds_bytes<PfKernelPoint, 2, 1>(), 2, 128);
  else launch(pf_step_kernel<9, 6, 2, 2, PF_KIND_POINTFOOT, PF_KSPH, 2, 1, 16, PF_NBD_POINTFOOT, false>, pf_lds_bytes<PfKernelPoint, 2, 1>(), 2, 64);
  PF_HIP_OK(hipGetLastError());
  return 0;
}

int pf_step(PfEnv* e, const float* d_actions, int64_t step_counter, const PfInject* inject, void* stream) {
  return pf_launch(e, d_actions, step_counter, inject, stream, 0);
}
int pf_step_policy(PfEnv* e, const PfPolicyHeads* h, int64_t step_counter, void* stream) {
  if (!h || !h->d_ha || !h->d_hc || !h->d_wa || !h->d_ba || !h->d_wc || !h->d_bc || !h->d_std || !h->d_obs || !h->d_critic_obs || !h->d_actions_out ||
      !h->s_obs || !h->s_critic_obs || !h->s_actions || !h->s_values || !h->s_log_prob || !h->s_mu || !h->s_sigma) {
    pf_set_error("pf_step_policy: incomplete PfPolicyHeads (fill it with ppo_act_hidden)"); return 1;
  }
  if (e && (h->num_actions != e->cfg.num_actions || h->num_actions > 16)) { pf_set_error("pf_step_policy: the policy's action count is not the env's"); return 1; }
  if (h->ka < 4 || h->kc < 4 || h->ka > 128 || h->kc > 128 || ((h->ka | h->kc) & 3) || ((((size_t)h->d_ha) | ((size_t)h->d_hc) | ((size_t)h->d_wa) | ((size_t)h->d_wc)) & 15)) {
    pf_set_error("pf_step_policy: head widths must be multiples of 4 up to 128 with 16-byte aligned rows"); return 1;
  }
  if (h->p_rew && !(h->p_dones && h->p_s_rew && h->p_s_done && (!h->p_time_outs || h->p_values))) { pf_set_error("pf_step_policy: incomplete deferred process_env_step"); return 1; }
  return pf_launch(e, nullptr, step_counter, nullptr, stream, 0, nullptr, h);
}
int pf_reset_all(PfEnv* e, int64_t step_counter, const PfInject* inject, void* stream) {
  return pf_launch(e, nullptr, step_counter, inject, stream, 1);
}
int pf_reset_idx(PfEnv* e, const uint8_t* d_mask, int64_t step_counter, const PfInject* inject, void* stream) {
  if (!d_mask) { pf_set_error("pf_reset_idx: mask pointer is NULL (use pf_reset_all)"); return 1; }
  return pf_launch(e, nullptr, step_counter, inject, stream, 1, d_mask);
}
int pf_compute_observations(PfEnv* e, int64_t step_counter, const PfInject* inject, void* stream) {
  return pf_launch(e, nullptr, step_counter, inject, stream, 2);
}
int pf_measure_heights(PfEnv* e, void* stream) {
  if (e && e->bound && !e->bufs.d_measured_heights) { pf_set_error("pf_measure_heights: d_measured_heights is not bound"); return 1; }
  return pf_launch(e, nullptr, 0, nullptr, stream, 3);
}

}  // extern "C"
