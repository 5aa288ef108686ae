/*
 * pointfoot_hip.h — C ABI of libpointfoot_hip.so (MI355X / gfx950).
 *
 * This is the drop-in boundary for the north-star hot path of peachvegetable/pointfoot:
 * the vectorised env step (reference legged_gym/envs/pointfoot/point_foot.py:170-230, i.e.
 * PointFoot.step + post_physics_step, including the physics call it makes into Isaac Gym at
 * point_foot.py:182-185) and the rsl_rl PPO rollout / GAE / minibatch update the reference
 * reaches through legged_gym/utils/task_registry.py:147 and legged_gym/scripts/train.py:44.
 *
 * Conventions
 *   - plain C, pointers + sizes only; no torch / C++ types cross the boundary;
 *   - every pointer named d_* is a DEVICE pointer into memory the caller (PyTorch) allocated and keeps
 *     alive; kernels never allocate;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing synchronises;
 *   - int return: 0 = ok, non-zero = failure, message via pf_last_error() (thread-local);
 *   - the host side (Python `pointfoot_amd`) mirrors every struct below with ctypes and checks
 *     pf_abi_sizes() at load time.
 *
 * Reference interface each entry point replaces is cited at its declaration (paths relative to
 * /root/reference).
 */
#ifndef POINTFOOT_HIP_H
#define POINTFOOT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PF_MAX_BODIES 20
#define PF_MAX_DOF 12
#define PF_MAX_FEET 4
#define PF_MAX_SPHERES 32
#define PF_MAX_REWARDS 32
#define PF_MAX_OBS 256
#define PF_MAX_HPTS 192
#define PF_MAX_TERM 8
#define PF_MAX_PEN 20
#define PF_MAX_CAPS 16         /* collision capsules (pairs of equal-radius spheres on one body; a lone sphere is a capsule of length 0) */
#define PF_MAX_PAIRS 64        /* self-collision capsule pairs (PointFoot: 16; a quadruped with 3-link legs: 54) */
#define PF_CSET_WORDS 4        /* 64-bit words per (env, substep) of PfBuffers.d_contact_set */
#define PF_STATS_FLOATS (PF_MAX_REWARDS + 4)
#define PF_STAT_CAP_HITS (PF_MAX_REWARDS + 1)     /* d_reset_stats: (env, substep) pairs of this launch with more active contact candidates than the solver keeps */
#define PF_STAT_CAP_DROPPED (PF_MAX_REWARDS + 2)  /* d_reset_stats: contact candidates dropped by that cap in this launch */
/* actuator network `LSTMsea` (resources/actuator_nets/anydrive_v3_lstm.pt): in_scale(2), out_scale(1), LSTM(2->8) x 2 layers
 * (PyTorch gate order i,f,g,o), Linear(8->1).  Flat parameter block handed over in d_actuator_params: */
#define PF_ACT_IN_SCALE 0
#define PF_ACT_OUT_SCALE 2
#define PF_ACT_W_IH0 3      /* [32][2] */
#define PF_ACT_W_HH0 67     /* [32][8] */
#define PF_ACT_B_IH0 323    /* [32] */
#define PF_ACT_B_HH0 355
#define PF_ACT_W_IH1 387    /* [32][8] */
#define PF_ACT_W_HH1 643
#define PF_ACT_B_IH1 899
#define PF_ACT_B_HH1 931
#define PF_ACT_W_LIN 963    /* [8] */
#define PF_ACT_B_LIN 971
#define PF_ACT_PARAMS 972

/* reward term ids — one per `_reward_*` method of point_foot.py:1132-1253 */
enum PfReward {
  PF_REW_LIN_VEL_Z = 0,        /* :1132 */
  PF_REW_ANG_VEL_XY,           /* :1136 */
  PF_REW_ORIENTATION,          /* :1140 */
  PF_REW_BASE_HEIGHT,          /* :1144 */
  PF_REW_TORQUES,              /* :1149 */
  PF_REW_DOF_VEL,              /* :1153 */
  PF_REW_DOF_ACC,              /* :1157 */
  PF_REW_ACTION_RATE,          /* :1161 */
  PF_REW_COLLISION,            /* :1165 */
  PF_REW_TERMINATION,          /* :1170 */
  PF_REW_DOF_POS_LIMITS,       /* :1174 */
  PF_REW_DOF_VEL_LIMITS,       /* :1180 */
  PF_REW_TORQUE_LIMITS,        /* :1187 */
  PF_REW_TRACKING_LIN_VEL,     /* :1192 */
  PF_REW_TRACKING_ANG_VEL,     /* :1197 */
  PF_REW_FEET_AIR_TIME,        /* :1202 */
  PF_REW_NO_FLY,               /* :1215 */
  PF_REW_UNBALANCE_FEET_AIR_TIME, /* :1220 */
  PF_REW_UNBALANCE_FEET_HEIGHT,   /* :1223 */
  PF_REW_STUMBLE,              /* :1226 */
  PF_REW_STAND_STILL,          /* :1231 */
  PF_REW_FEET_CONTACT_FORCES,  /* :1237 */
  PF_REW_FEET_DISTANCE,        /* :1242 */
  PF_REW_SURVIVAL,             /* :1252 */
  PF_REW_COUNT
};

enum PfEnvKind { PF_KIND_POINTFOOT = 0 /* envs/pointfoot/point_foot.py */, PF_KIND_LEGGED_ROBOT = 1 /* envs/base/legged_robot.py (+ anymal.py) */ };
enum PfControl { PF_CTRL_P = 0, PF_CTRL_V = 1, PF_CTRL_T = 2 };      /* point_foot.py:536-546 */
enum PfTerrain { PF_TERRAIN_PLANE = 0, PF_TERRAIN_HEIGHTFIELD = 1 }; /* point_foot.py:390-400 (trimesh is served by the height field) */

/* Collapsed articulated model (what Isaac Gym's load_asset derives from the URDF with
 * collapse_fixed_joints=True; point_foot.py:883-909).  Bodies are in
 * depth-first order, parent[b] < b, a subtree is the contiguous range [b, b+subtree[b]).
 * pf_create integrates a copy in which bodies behind fixed joints (the feet) are lumped into their parents
 * (pointfoot_amd/csrc/pf_lump.h: the same rigid-body system, shorter kinematic chains); everything the env reports per body
 * (contact forces, feet positions, termination / penalised contact indices) keeps the body indices of THIS struct. */
typedef struct PfModel {
  int32_t nb, nd, nfeet, nsph;
  int32_t parent[PF_MAX_BODIES];
  int32_t subtree[PF_MAX_BODIES];
  int32_t depth[PF_MAX_BODIES];
  int32_t dof_of_body[PF_MAX_BODIES];   /* -1: floating base or fixed joint */
  int32_t body_of_dof[PF_MAX_DOF];
  float joint_pos[PF_MAX_BODIES][3];    /* joint origin in the parent link frame */
  float joint_rot[PF_MAX_BODIES][9];    /* fixed rotation parent link -> joint frame, row-major */
  float joint_axis[PF_MAX_BODIES][3];   /* unit axis in the joint frame */
  float mass[PF_MAX_BODIES];
  float com[PF_MAX_BODIES][3];          /* link frame */
  float inertia[PF_MAX_BODIES][6];      /* about the COM, link axes: xx,yy,zz,xy,xz,yz */
  float q_lo[PF_MAX_DOF], q_hi[PF_MAX_DOF];   /* hard URDF limits */
  float vel_limit[PF_MAX_DOF];
  float effort_limit[PF_MAX_DOF];
  int32_t sph_body[PF_MAX_SPHERES];     /* collision spheres (capsule ends, box corners, feet) */
  float sph_pos[PF_MAX_SPHERES][3];     /* link frame */
  float sph_radius[PF_MAX_SPHERES];
  int32_t feet_body[PF_MAX_FEET];
  int32_t n_term, n_pen;
  int32_t term_body[PF_MAX_TERM];       /* termination_contact_indices, point_foot.py:963-968 */
  int32_t pen_body[PF_MAX_PEN];         /* penalised_contact_indices,   point_foot.py:956-961 */
  /* self-collision geometry (asset.self_collisions = 0 enables it, pointfoot_rough_config.py:107, point_foot.py:942):
   * capsule c spans spheres cap_s0[c] .. cap_s1[c] (same body, same radius); pairs list the capsules that may touch
   * (bodies on different branches of the tree; a parent-child pair never collides, as in PhysX) */
  int32_t ncap, npair;
  int32_t cap_s0[PF_MAX_CAPS], cap_s1[PF_MAX_CAPS];
  int32_t pair_a[PF_MAX_PAIRS], pair_b[PF_MAX_PAIRS];
} PfModel;

/* Task constants: everything PointFoot reads from cfg (pointfoot_rough_config.py) after _parse_cfg /
 * _init_buffers / _prepare_reward_function (point_foot.py:999-1009, 698-820). */
typedef struct PfConfig {
  int32_t num_envs;
  int32_t env_index_base;        /* global index of env 0 of this shard (multi-GPU: rank*num_envs) */
  int32_t num_envs_global;
  int32_t num_obs, num_priv_obs, num_actions;
  int32_t control_type;          /* PfControl */
  int32_t decimation;
  int32_t terrain_type;          /* PfTerrain */
  int32_t measure_heights;       /* cfg.terrain.measure_heights_critic */
  int32_t num_height_points;
  int32_t heading_command;
  int32_t num_commands;
  int32_t curriculum;            /* cfg.terrain.curriculum (forced 0 on plane, :1004-1005) */
  int32_t custom_origins;        /* :974-975 */
  int32_t max_terrain_level;     /* cfg.terrain.num_rows */
  int32_t terrain_num_cols;
  int32_t hf_rows, hf_cols;      /* height_samples.shape */
  int32_t push_robots;
  int32_t push_interval;         /* ceil(push_interval_s / dt), :1009 */
  int32_t add_noise;
  int32_t only_positive_rewards;
  int32_t include_feet_height;   /* _check_if_include_feet_height_rewards, :232-237 */
  int32_t num_rewards;
  int32_t reward_id[PF_MAX_REWARDS];      /* PfReward, in the reference's summation order (:307-311) */
  float reward_scale[PF_MAX_REWARDS];     /* scale * dt (:806) */
  float termination_scale;                /* 0 when absent (:315-318) */
  float sim_dt;                  /* cfg.sim.dt */
  float dt;                      /* decimation * sim_dt (:1000) */
  float max_episode_length;      /* ceil(episode_length_s / dt) (:1007) */
  float max_episode_length_s;
  float action_scale, clip_actions, clip_obs;
  float p_gains[PF_MAX_DOF], d_gains[PF_MAX_DOF], default_dof_pos[PF_MAX_DOF];
  float torque_limits[PF_MAX_DOF];        /* :458 */
  float dof_pos_soft_lo[PF_MAX_DOF], dof_pos_soft_hi[PF_MAX_DOF];  /* :460-463 */
  float dof_vel_limits[PF_MAX_DOF];
  float obs_scale_lin_vel, obs_scale_ang_vel, obs_scale_dof_pos, obs_scale_dof_vel, obs_scale_height;
  float commands_scale[3];       /* :745 */
  float obs_noise_vec[PF_MAX_OBS];        /* _get_noise_scale_vec()[0] (:652-676) */
  float priv_height_noise;       /* _get_noise_scale_vec()[1] entries (:689-693) */
  float base_init_state[13];     /* :918-919 */
  float height_px[PF_MAX_HPTS], height_py[PF_MAX_HPTS];  /* _init_height_points (:1032-1046), x-major */
  float hscale, vscale, border;  /* cfg.terrain.horizontal_scale / vertical_scale / border_size */
  float terrain_length;          /* cfg.terrain.terrain_length (curriculum threshold, :625) */
  float base_height_target, tracking_sigma, max_contact_force;
  float soft_dof_vel_limit, soft_torque_limit;
  float min_feet_air_time, max_feet_air_time, min_feet_distance;
  float max_push_force;          /* mean(base_mass) * max_push_vel_xy / sim_dt (:592-596) */
  float gravity[3];
  /* physics (own solver; reference delegates to PhysX — SURVEY.md §8 a3, parity unpinned) */
  float terrain_friction, terrain_restitution;
  float contact_offset, bounce_threshold, max_depenetration_vel;
  int32_t solver_iterations;
  uint32_t seed_lo, seed_hi;     /* Philox key */
  /* LeggedRobot variant (legged_gym/envs/base/legged_robot.py, anymal_c/anymal.py) */
  int32_t env_kind;              /* PfEnvKind */
  int32_t resample_interval;     /* int(resampling_time / dt)        legged_robot.py:323 */
  int32_t use_actuator_net;      /* cfg.control.use_actuator_network anymal.py:71-78 */
  float cmd_lo[4], cmd_hi[4];    /* ranges: lin_vel_x, lin_vel_y, ang_vel_yaw, heading   legged_robot.py:343-349 */
  float max_push_vel;            /* cfg.domain_rand.max_push_vel_xy  legged_robot.py:417 */
  /* trimesh collision surface (legged_gym/utils/terrain.py:69-73, point_foot.py:852-868): height steps steeper than
   * slope_treshold become vertical faces.  wall_threshold = slope_treshold * horizontal_scale in metres of height per
   * grid step; 0 = plain bilinear height field (mesh_type 'heightfield', or a plane) */
  float wall_threshold;
  int32_t self_collisions;       /* 1: leg-leg capsule contacts on (asset.self_collisions == 0) */
  float joint_friction_vel;      /* regularisation speed of the Coulomb joint friction (rad/s); see d_joint_friction */
} PfConfig;

/* Per-env state record: one contiguous row of `rec_floats` floats per env (AoS: a wavefront owns
 * one env, so its record is one coalesced burst).  Offsets come from pf_layout().
 * The record is in two parts, each a whole number of 64-byte lines: [0, rw_floats) is what a step READS and writes (state that
 * carries over), [rw_floats, rec_used) what a step only WRITES (actions as clipped, torques, base-frame velocities, projected
 * gravity, net contact forces, feet positions: outputs the caller reads).  pf_step does not load the second part. */
typedef struct PfLayout {
  int32_t rec_floats;
  int32_t rw_floats, rec_used;
  int32_t root, dof_pos, dof_vel, actions, last_actions, last_dof_vel, last_root_vel, commands, torques;
  int32_t base_lin_vel, base_ang_vel, projected_gravity;
  int32_t feet_air_time, last_feet_air_time, cur_max_feet_height, last_max_feet_height, feet_height;
  int32_t contact_filt, first_contact, last_contacts;   /* 0/1 as float */
  int32_t contact_forces, feet_pos, ext_force /* xyz + pending flag */, env_origin;
  int32_t friction, added_mass, added_com;
  int32_t lstm_state;            /* LeggedRobot: actuator-net h/c, [joint][layer][h(8) c(8)] */
  int32_t episode_sums;
  int32_t rand_floats;           /* width of one row of the injected-random buffer */
  int32_t r_obs, r_priv, r_dof, r_xy, r_vel, r_level, r_push;   /* offsets inside that row */
  int32_t r_cmd, r_cmd_reset;    /* LeggedRobot command resampling: periodic / at reset (3 draws each) */
} PfLayout;

/* Device buffers owned by the caller. */
typedef struct PfBuffers {
  float* d_state;              /* (N, rec_floats) */
  int64_t* d_episode_length;   /* (N,) episode_length_buf, point_foot.py:69 */
  int32_t* d_terrain_level;    /* (N,)  :980 */
  int32_t* d_terrain_type;     /* (N,)  :981 */
  uint8_t* d_reset;            /* (N,) reset_buf as bool  :242 */
  uint8_t* d_time_out;         /* (N,) :246 */
  uint8_t* d_terminate;        /* (N,) :244 */
  float* d_obs;                /* (N, num_obs)       proprioceptive_obs_buf */
  float* d_priv;               /* (N, num_priv_obs)  privileged_obs_buf */
  float* d_rew;                /* (N,) rew_buf */
  float* d_measured_heights;   /* (N, H) or NULL */
  const int16_t* d_height_samples; /* (hf_rows, hf_cols) or NULL on a plane  :849,867 */
  const float* d_terrain_origins;  /* (max_terrain_level, terrain_num_cols, 3) or NULL */
  const float* d_actuator_params; /* LeggedRobot + actuator net: PF_ACT_PARAMS floats (see PF_ACT_* offsets) or NULL */
  float* d_reset_stats;        /* (PF_STATS_FLOATS,): [i] = sum of episode_sums[i] over the envs reset this step
                                  (point_foot.py:276-280), [PF_MAX_REWARDS] = number of envs reset, [PF_STAT_CAP_HITS] /
                                  [PF_STAT_CAP_DROPPED] = contact-cap diagnostics (HIP path only); zeroed by pf_step */
  const float* d_joint_friction; /* (N, nd) Coulomb joint friction torque per DOF (update_frictions, point_foot.py:103-108:
                                  dof_props["friction"]) or NULL = none */
  uint64_t* d_contact_set;     /* (N, decimation, PF_CSET_WORDS) or NULL.  Diagnostic output for parity tests, the discrete decisions of
                                  substep `sub`: word 0 = bit c set when contact candidate c < 64 was active, word 3 = the same for
                                  candidates 64..127 (candidates: floor contact of sphere s = s, vertical-face contact of sphere
                                  s = nsph + s, capsule pair p = 2 nsph + p); word 1 = joint-limit rows, 2 bits per DOF (1: at the
                                  lower, 2: at the upper limit); word 2 = sum over the kept terrain contacts of
                                  (candidate + 1) * (4 * grid cell + face): which cell of the (piecewise, stepped) collision
                                  surface each contact was evaluated in */
} PfBuffers;

/* Optional injected inputs (parity tests).  All NULL in production. */
typedef struct PfInject {
  const float* d_rand;         /* (N, rand_floats) uniforms in [0,1) replacing Philox; r_level holds the integer draw */
  const float* d_dof_pre;      /* (N, nd, 2) DOF state seen by the last torque computation */
  const float* d_dof_post;     /* (N, nd, 2) DOF state after the last simulate */
  const float* d_root;         /* (N, 13) */
  const float* d_contact;      /* (N, nb, 3) */
  const float* d_feet_pos;     /* (N, nfeet, 3) */
} PfInject;

/* The policy's last step handed to the env step (pf_step_policy): everything PPO.act does behind the last hidden layer --
 * the two thin heads, Normal(mu, std).sample(), log-prob, RolloutStorage.add_transitions for transition t -- and the
 * PPO.process_env_step of transition t - 1, carried out by the step kernel's own wavefronts before they read the action.
 * Filled by ppo_act_hidden (never by hand); all pointers are device pointers that stay valid until the step has run. */
typedef struct PfPolicyHeads {
  const float* d_ha;           /* (N, ka) actor, last hidden activations */
  const float* d_hc;           /* (N, kc) critic, last hidden activations */
  const float* d_wa;           /* (A, ka) action-mean head */
  const float* d_ba;           /* (A) */
  const float* d_wc;           /* (kc) value head */
  const float* d_bc;           /* (1) */
  const float* d_std;          /* (A) action std parameter */
  const float* d_inject_normal;/* (N, A) standard normal draws replacing Philox (parity tests), or NULL */
  const float* d_obs;          /* (N, O) the observations the policy acted on: copied to storage row t */
  const float* d_critic_obs;   /* (N, P) */
  float* d_actions_out;        /* (N, A) sampled actions (unclipped), as ppo_act returns them */
  float* s_obs;                /* storage rows of transition t */
  float* s_critic_obs;
  float* s_actions;
  float* s_values;
  float* s_log_prob;
  float* s_mu;
  float* s_sigma;
  /* deferred PPO.process_env_step of transition t - 1 (p_rew == NULL: none pending) */
  const float* p_rew;
  const uint8_t* p_dones;
  const uint8_t* p_time_outs;  /* NULL: no bootstrap on time-outs */
  const float* p_values;       /* storage values of transition t - 1 */
  float* p_s_rew;
  float* p_s_done;
  float* p_ep_ret;             /* runner bookkeeping (NULL: off) */
  float* p_ep_len;
  float* p_ep_stats;
  float* p_fin_ret;
  float* p_fin_len;
  float p_gamma;
  int32_t ka, kc, num_obs, num_critic_obs, num_actions;
  uint32_t seed_lo, seed_hi, counter, env_base;
} PfPolicyHeads;

typedef struct PfEnv PfEnv;

const char* pf_last_error(void);
/* sizes[0..7] = sizeof PfModel, PfConfig, PfLayout, PfBuffers, PfInject, PpoConfig, PpoBuffers, PfPolicyHeads */
void pf_abi_sizes(int64_t* sizes, int32_t n);
/* record layout for a model/config pair (pure host computation) */
int pf_layout(const PfModel* model, const PfConfig* cfg, PfLayout* out);

/* replaces: task constructor  task_class(cfg, sim_params, physics_engine, sim_device, headless)
 * (task_registry.py:97-101 -> PointFoot.__init__, point_foot.py:17-101) */
PfEnv* pf_create(const PfModel* model, const PfConfig* cfg, int32_t device);
void pf_destroy(PfEnv* env);
int pf_bind_buffers(PfEnv* env, const PfBuffers* bufs);
/* re-point d_reset_stats (host-only; lets the caller rotate per-step statistics slots without copies) */
int pf_set_reset_stats(PfEnv* env, float* d_reset_stats);
/* optional: announce the slot the FOLLOWING step will use; the step kernel clears it on the device, so a rollout that
 * rotates slots issues no memset between steps (a slot not announced this way is cleared by pf_step itself) */
int pf_set_reset_stats_next(PfEnv* env, float* d_next_reset_stats);
/* replaces: gymapi tensors + update hooks; re-uploads config (e.g. after update_cmd / cfg edits) */
int pf_update_config(PfEnv* env, const PfConfig* cfg);

/* replaces: PointFoot.step(actions) (point_foot.py:170-193): clip, decimation x (PD torque + physics
 * substep), post_physics_step, obs clip.  One launch.  `step_counter` is common_step_counter AFTER
 * the increment of :205.  With inject != NULL && inject->d_root != NULL the physics substeps are
 * skipped and the injected sim state is used instead (Python-layer parity mode). */
int pf_step(PfEnv* env, const float* d_actions, int64_t step_counter, const PfInject* inject, void* stream);

/* pf_step with the policy's heads inside the same launch: replaces the pair  actions = alg.act(obs, critic_obs);
 * env.step(actions)  of rsl_rl OnPolicyRunner.learn (call site legged_gym/scripts/train.py:44 -> point_foot.py:170) for a
 * policy whose dense layers have already run (ppo_act_hidden).  One launch fewer per policy step; storage rows, sampled
 * actions and the env step are bit-identical to ppo_act followed by pf_step.  heads->num_actions must equal the env's. */
int pf_step_policy(PfEnv* env, const PfPolicyHeads* heads, int64_t step_counter, void* stream);

/* replaces: reset_idx(arange(N)) of PointFoot.reset() (point_foot.py:137-139); the zero-action step of
 * :140-141 is a separate pf_step call made by the host. */
int pf_reset_all(PfEnv* env, int64_t step_counter, const PfInject* inject, void* stream);

/* replaces: reset_idx(env_ids) for a subset (point_foot.py:249-299; scripted single-env resets, e.g.
 * scripts/simulate_trajectory.py:72-97).  d_mask (N,) bytes: non-zero = reset this env.  Same launch as pf_reset_all. */
int pf_reset_idx(PfEnv* env, const uint8_t* d_mask, int64_t step_counter, const PfInject* inject, void* stream);

/* replaces: compute_observations() re-run by update_cmd (point_foot.py:123-129) */
int pf_compute_observations(PfEnv* env, int64_t step_counter, const PfInject* inject, void* stream);
/* The height scan alone (reference: the `measured_heights` attribute, point_foot.py:213 / :1048-1075): fills
 * PfBuffers.d_measured_heights for the current state and writes nothing else -- no observation, reward or state buffer. */
int pf_measure_heights(PfEnv* env, void* stream);

/* ------------------------------------------------------------------------------------------- PPO
 * replaces rsl_rl v1.0.2 (un-vendored dependency; setup.py:13): OnPolicyRunner.learn's inner calls
 * alg.act / alg.process_env_step / alg.compute_returns / alg.update (call site train.py:44). */
#define PPO_MAX_LAYERS 4       /* hidden layers + 1 output */
#define PPO_EP_RING 100        /* rsl_rl OnPolicyRunner: rewbuffer = deque(maxlen=100) */
#define PPO_EP_RING_INTS (4 + 4 * PPO_EP_RING)

typedef struct PpoConfig {
  int32_t num_envs, num_steps;            /* N, T (num_steps_per_env, pointfoot_rough_config.py:246) */
  int32_t num_obs, num_critic_obs, num_actions;
  int32_t n_actor_layers, n_critic_layers;    /* number of Linear layers (hidden + output) */
  int32_t actor_dims[PPO_MAX_LAYERS + 1];     /* in, h1, ..., out */
  int32_t critic_dims[PPO_MAX_LAYERS + 1];
  int32_t num_learning_epochs, num_mini_batches;
  int32_t use_clipped_value_loss, adaptive_schedule;
  float clip_param, value_loss_coef, entropy_coef;
  float gamma, lam, desired_kl, max_grad_norm;
  float learning_rate;
  float adam_beta1, adam_beta2, adam_eps;
  int32_t world_size;                     /* data-parallel ranks sharing the gradient */
  uint32_t seed_lo, seed_hi;
  int32_t gemm_mode;                      /* arithmetic of the dense layers: 1 = fp32 MFMA (v_mfma_f32_32x32x2_f32); 2 = fp32 by exact
                                             three-way bf16 operand split on the bf16 MFMA ("bf16x6": fp32-grade products, fp32
                                             accumulation; csrc/pf_gemm3.h); 0 = by size (2 when a hidden layer is >= 256 wide) */
} PpoConfig;

typedef struct PpoBuffers {
  float* d_params;     /* flat: actor (W0,b0,..), critic (W0,b0,..), std[A]; W row-major (out,in) like nn.Linear */
  float* d_grads;      /* same layout */
  float* d_adam_m;
  float* d_adam_v;
  /* rollout storage, time-major [T, N, .] (rsl_rl RolloutStorage) */
  float* d_obs;        /* [T,N,O] */
  float* d_critic_obs; /* [T,N,P] */
  float* d_actions;    /* [T,N,A] */
  float* d_rewards;    /* [T,N] */
  float* d_dones;      /* [T,N] */
  float* d_values;     /* [T,N] */
  float* d_returns;    /* [T,N] */
  float* d_advantages; /* [T,N] */
  float* d_log_prob;   /* [T,N] */
  float* d_mu;         /* [T,N,A] */
  float* d_sigma;      /* [T,N,A] */
  int32_t* d_perm;     /* [T*N] minibatch permutation (randperm), written by the host */
  float* d_workspace;  /* activations / gradients scratch */
  int64_t workspace_floats;
  float* d_ep_return;  /* [N] running undiscounted return of the current episode (runner statistics), or NULL */
  float* d_ep_length;  /* [N] running length of the current episode, or NULL */
  float* d_ep_stats;   /* [4]: sum of finished-episode returns, sum of their lengths, count (since the host last zeroed it); added to
                          by every ppo_process_env_step, or -- with d_ep_fin and d_ep_ring bound -- once per rollout by ppo_compute_returns */
  float* d_ep_fin;     /* [2,T,N] or NULL: undiscounted return and length of the episode that ended at (t, env); written where done */
  int32_t* d_ep_ring;  /* [PPO_EP_RING_INTS] or NULL: the last PPO_EP_RING finished episodes in (step, env) order -- what rsl_rl's
                          OnPolicyRunner keeps in rewbuffer / lenbuffer = deque(maxlen=100) (call site legged_gym/scripts/train.py:44).
                          Header {next slot, entries held, episodes seen, policy steps seen}; then per slot {return (float bits),
                          length (float bits), policy-step index, episode sequence number}.  Appended to once per rollout by
                          ppo_compute_returns from d_dones / d_ep_fin; the host zeroes it at start and only ever reads it. */
  float* d_scalars;    /* [16]: 0 lr, 1 mean_kl, 2 grad_norm, 3 value_loss_sum, 4 surrogate_loss_sum,
                          5 adam_step, 6 adv_mean, 7 adv_std, 8 kl_sum_local, 9 n_updates */
} PpoBuffers;

typedef struct PpoAlg PpoAlg;

PpoAlg* ppo_create(const PpoConfig* cfg, int32_t device);
void ppo_destroy(PpoAlg* alg);
int ppo_bind_buffers(PpoAlg* alg, const PpoBuffers* bufs);
int64_t ppo_num_params(const PpoConfig* cfg);
/* Offsets (in floats) of every parameter tensor inside the flat buffer, in the order actor (W, b) per layer, critic
 * (W, b) per layer, action std.  Tensors start on 16-byte boundaries; the padding floats between them stay zero
 * (zero gradient, untouched by Adam).  Returns the number of tensors (writes at most max_entries), -1 on error. */
int32_t ppo_param_offsets(const PpoConfig* cfg, int64_t* offsets, int32_t max_entries);
int64_t ppo_workspace_floats(const PpoConfig* cfg);

/* PPO.act: actor+critic forward on (obs, critic_obs), Normal(mu, std).sample(), log-prob, store
 * transition `t` (rsl_rl PPO.act + RolloutStorage.add_transitions).  d_actions_out (N,A).
 * The caller may rewrite d_params between rollouts (checkpoint load, broadcast); derived copies of the weights are refreshed
 * at t == 0, on the first dense call after ppo_bind_buffers or an optimiser step whatever its t, and by every other entry point
 * that runs a dense layer -- not between the transitions of one rollout after an external rewrite of d_params. */
int ppo_act(PpoAlg* alg, int32_t t, const float* d_obs, const float* d_critic_obs, float* d_actions_out,
            int64_t sample_counter, const float* d_inject_normal, void* stream);
/* PPO.act up to the last hidden layer (the dense launches only), for an env that finishes the step itself: fills `heads` for
 * pf_step_policy, which then computes the heads, samples, stores transition t and carries out a pending deferred
 * process_env_step.  Returns 2 (and fills nothing) when the handle cannot hand its heads over (head widths above 128 or not
 * multiples of 4, heads not fused): the caller falls back to ppo_act + pf_step. */
int ppo_act_hidden(PpoAlg* alg, int32_t t, const float* d_obs, const float* d_critic_obs, float* d_actions_out,
                   int64_t sample_counter, const float* d_inject_normal, PfPolicyHeads* heads, void* stream);
/* actor forward only (act_inference / get_inference_policy) */
int ppo_act_inference(PpoAlg* alg, const float* d_obs, int32_t n, float* d_actions_out, void* stream);
/* PPO.process_env_step: rewards += gamma * value * time_outs; store rewards / dones for step t */
int ppo_process_env_step(PpoAlg* alg, int32_t t, const float* d_rew, const uint8_t* d_dones,
                         const uint8_t* d_time_outs, void* stream);
/* The same transition processing, carried out in the tail of the NEXT ppo_act's head kernel (or by ppo_compute_returns / the next
 * ppo_process_env_step*, whichever is enqueued first) instead of a launch of its own.  The caller promises that the three input
 * buffers keep this step's values until that call has been enqueued on the same stream -- true of an env whose step outputs are
 * persistent buffers rewritten only by its next step (OnPolicyRunner: act -> env.step -> process_env_step -> act ...).  Storage
 * rows d_rewards / d_dones of step t and the episode bookkeeping are defined only after that later call. */
int ppo_process_env_step_deferred(PpoAlg* alg, int32_t t, const float* d_rew, const uint8_t* d_dones,
                                  const uint8_t* d_time_outs, void* stream);
/* PPO.compute_returns: critic(last_critic_obs), GAE reverse scan, advantage normalisation.
 * With world_size > 1 the host all-reduces d_scalars[10..12] (sum, sumsq, count) between the two halves. */
int ppo_compute_returns(PpoAlg* alg, const float* d_last_critic_obs, void* stream);
int ppo_normalize_advantages(PpoAlg* alg, void* stream);
/* Optional, once per PPO.update after d_perm has been written: brings the rollout storage into the permutation's order in
 * one pass (rsl_rl draws one permutation per update and reuses it in every epoch), so that every ppo_minibatch_backward of
 * this update reads contiguous slices instead of gathering its rows.  Without it each minibatch gathers for itself.  The
 * copy is dropped by the next ppo_act(t = 0) and ppo_normalize_advantages; rewrite d_perm or the storage any other way and
 * this has to be called again. */
int ppo_begin_update(PpoAlg* alg, void* stream);
/* PPO.update, one minibatch: gather, forward, KL + adaptive LR, losses, backward into d_grads. */
int ppo_minibatch_backward(PpoAlg* alg, int32_t minibatch_index, void* stream);
/* The same minibatch in two parts, for the multi-GPU path (no counterpart in the reference: rsl_rl is single-process, the
 * reference's only hook is the dead --horovod flag, legged_gym/utils/helpers.py:162).  part 0: forward, losses, and every
 * gradient except the two input layers' -- afterwards the EARLY range of d_grads is final; part 1: the rest -- afterwards the
 * LATE range is.  The host starts the all-reduce of a range as soon as its part has been issued, so the first (90 % of the
 * bytes) runs under the kernels of part 1, and waits for both only before ppo_minibatch_apply.  ppo_grad_groups writes
 * {early_offset, early_length, late_offset, late_length} in floats (the early range includes the 4-float KL tail behind the
 * parameters); returns 0, -1 on a bad config.  Needs the paired execution mode (actor and critic of equal depth). */
int ppo_minibatch_backward_part(PpoAlg* alg, int32_t minibatch_index, int32_t part, void* stream);
int32_t ppo_grad_groups(const PpoConfig* cfg, int64_t* ranges4);
/* 1 when THIS handle can run ppo_minibatch_backward_part (paired execution: equal depth, fused heads, not switched off by
 * PF_PPO_PAIRED=0 / PF_PPO_FUSED_HEADS=0), else 0: the host then keeps the whole-pass call and one blocking all-reduce. */
int32_t ppo_supports_parts(const PpoAlg* alg);
/* grad-norm clip (over the possibly all-reduced d_grads, divided by world_size) + Adam step */
int ppo_minibatch_apply(PpoAlg* alg, void* stream);
/* ppo_minibatch_backward + ppo_minibatch_apply for ONE process (world_size == 1: nothing happens to d_grads between the two): the
 * slab reduction leaves the gradient norm's first stage, and clip coefficient, adaptive learning rate and Adam run as one launch
 * (two launches fewer per minibatch).  Same arithmetic except the summation order of the squared norm. */
int ppo_minibatch_step(PpoAlg* alg, int32_t minibatch_index, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* POINTFOOT_HIP_H */
