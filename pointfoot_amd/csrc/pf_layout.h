// Per-env state record layout (host code).  One env = one contiguous row of floats so that the
// wavefront that owns the env moves it with a single coalesced burst in and out of LDS.
#pragma once
#include "../../include/pointfoot_hip.h"

static inline int pf_compute_layout(const PfModel* m, const PfConfig* c, PfLayout* L) {
  const int nd = m->nd, nb = m->nb, nf = m->nfeet, na = c->num_actions;
  int o = 0;
  auto take = [&](int n) { int at = o; o += n; return at; };
  // read-write part (state carried from step to step)
  L->root = take(13);
  L->dof_pos = take(nd);
  L->dof_vel = take(nd);
  L->last_actions = take(na);
  L->last_dof_vel = take(nd);
  L->last_root_vel = take(6);
  L->commands = take(4);
  L->feet_air_time = take(nf);
  L->last_feet_air_time = take(nf);
  L->cur_max_feet_height = take(nf);
  L->last_max_feet_height = take(nf);
  L->feet_height = take(nf);
  L->contact_filt = take(nf);
  L->first_contact = take(nf);
  L->last_contacts = take(nf);
  L->ext_force = take(4);
  L->env_origin = take(3);
  L->friction = take(1);
  L->added_mass = take(1);
  L->added_com = take(3);
  L->lstm_state = take(c->env_kind == PF_KIND_LEGGED_ROBOT ? 32 * nd : 0);
  L->episode_sums = take(PF_MAX_REWARDS + 1);      // (a fixed block: the offsets behind it are compile-time constants of the kernel)
  o = (o + 15) & ~15;               // whole 64-byte lines
  L->rw_floats = o;
  // write-only part: a step overwrites all of it before it reads any of it (pf_step does not load these lines)
  L->actions = take(na);
  L->torques = take(nd);
  L->base_lin_vel = take(3);
  L->base_ang_vel = take(3);
  L->projected_gravity = take(3);
  L->contact_forces = take(3 * nb);
  L->feet_pos = take(3 * nf);
  L->rec_used = o;
  L->rec_floats = (o + 15) & ~15;   // 64-byte multiple
  int r = 0;
  auto rtake = [&](int n) { int at = r; r += n; return at; };
  L->r_obs = rtake(c->num_obs);
  L->r_priv = rtake(c->num_height_points);
  L->r_dof = rtake(nd);
  L->r_xy = rtake(2);
  L->r_vel = rtake(6);
  L->r_level = rtake(1);
  L->r_push = rtake(3);
  L->r_cmd = rtake(3);
  L->r_cmd_reset = rtake(3);
  L->rand_floats = (r + 3) & ~3;
  return 0;
}
