// TEST INFRASTRUCTURE — host emulation build of the HIP env-step program (pf_step_kernel.h compiled
// with -DPF_HOST_EMU: lane loops become serial loops).  Lets the CPU test-suite exercise the very
// source the GPU runs.  Never linked into the product library.
#define PF_HOST_EMU 1
#include "../../pointfoot_amd/csrc/pf_step_kernel.h"
#include "../../pointfoot_amd/csrc/pf_layout.h"

extern "C" int pfe_step(const PfModel* model, const PfConfig* cfg, const PfLayout* layout, const PfBuffers* bufs,
                        const float* actions, long long step_counter, const PfInject* inject, int mode, const unsigned char* reset_mask) {
  PfStepArgs g;
  g.packed = nullptr; g.cfg = cfg; g.L = *layout; g.b = *bufs;
  if (inject) g.inj = *inject; else std::memset(&g.inj, 0, sizeof(g.inj));
  g.actions = actions; g.step_counter = step_counter; g.mode = mode; g.stamps = nullptr; g.stats_next = nullptr; g.reset_mask = reset_mask;
  if (bufs->d_reset_stats && mode != 2) std::memset(bufs->d_reset_stats, 0, sizeof(float) * PF_STATS_FLOATS);
  // the product integrates the model with its fixed-joint bodies lumped into their parents (pf_create): so does its host build
  static thread_local PfModel lumped; static thread_local PfLump lump;
  pf_lump_model(model, &lumped, &lump);
  if (cfg->env_kind == PF_KIND_POINTFOOT && model->nb == 9 && model->nd == 6 && model->nfeet == 2) {
    using K = PfKernel<9, 6, 2, PF_KIND_POINTFOOT, PF_KSPH, 1, 16, 7>;
    static K::Shared sh; static K::SharedRO ro; static K::ModelS packed;
    K::pack_model(lumped, lump, packed); g.packed = &packed;
    for (int env = 0; env < cfg->num_envs; ++env) K::run(sh, ro, g, env);
  } else if (cfg->env_kind == PF_KIND_LEGGED_ROBOT && model->nb == 17 && model->nd == 12 && model->nfeet == 4 && model->npair > 0) {
    using K = PfKernel<17, 12, 4, PF_KIND_LEGGED_ROBOT, PF_KSPH_QUAD, 1, PF_KPAIR_QUAD, 13>;      // the instance with self-collision
    static K::Shared sh; static K::SharedRO ro; static K::ModelS packed;
    K::pack_model(lumped, lump, packed); g.packed = &packed;
    for (int env = 0; env < cfg->num_envs; ++env) K::run(sh, ro, g, env);
  } else if (cfg->env_kind == PF_KIND_LEGGED_ROBOT && model->nb == 17 && model->nd == 12 && model->nfeet == 4) {
    using K = PfKernel<17, 12, 4, PF_KIND_LEGGED_ROBOT, PF_KSPH_QUAD, 1, 0, 13>;
    static K::Shared sh; static K::SharedRO ro; static K::ModelS packed;
    K::pack_model(lumped, lump, packed); g.packed = &packed;
    for (int env = 0; env < cfg->num_envs; ++env) K::run(sh, ro, g, env);
  } else return 1;
  return 0;
}
extern "C" int pfe_layout(const PfModel* model, const PfConfig* cfg, PfLayout* out) { return pf_compute_layout(model, cfg, out); }
// the model the product integrates (fixed-joint bodies lumped into their parents) and its reporting tables, for the unit test of pf_lump.h
extern "C" void pfe_lump(const PfModel* model, PfModel* out, PfLump* map) { pf_lump_model(model, out, map); }
