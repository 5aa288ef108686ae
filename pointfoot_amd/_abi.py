"""ctypes mirror of include/pointfoot_hip.h and the loader of libpointfoot_hip.so.

The product path has NO fallback: if the HIP library is missing or its struct sizes disagree with this
mirror, loading raises.  (The CPU oracle under oracle/ is test infrastructure and is never imported here.)
"""
import ctypes as C
import os

PF_MAX_BODIES = 20
PF_MAX_DOF = 12
PF_MAX_FEET = 4
PF_MAX_SPHERES = 32
PF_MAX_REWARDS = 32
PF_MAX_OBS = 256
PF_MAX_HPTS = 192
PF_MAX_TERM = 8
PF_MAX_PEN = 20
PF_MAX_CAPS = 16
PF_MAX_PAIRS = 64
PF_CSET_WORDS = 4
PPO_MAX_LAYERS = 4
PPO_EP_RING = 100
PPO_EP_RING_INTS = 4 + 4 * PPO_EP_RING
PF_STATS_FLOATS = PF_MAX_REWARDS + 4
PF_STAT_CAP_HITS = PF_MAX_REWARDS + 1
PF_STAT_CAP_DROPPED = PF_MAX_REWARDS + 2
PF_ACT_PARAMS = 972
PF_KIND_POINTFOOT, PF_KIND_LEGGED_ROBOT = 0, 1

# order = enum PfReward
REWARD_IDS = {name: i for i, name in enumerate([
    "lin_vel_z", "ang_vel_xy", "orientation", "base_height", "torques", "dof_vel", "dof_acc", "action_rate",
    "collision", "termination", "dof_pos_limits", "dof_vel_limits", "torque_limits", "tracking_lin_vel",
    "tracking_ang_vel", "feet_air_time", "no_fly", "unbalance_feet_air_time", "unbalance_feet_height",
    "stumble", "stand_still", "feet_contact_forces", "feet_distance", "survival"])}
PF_CTRL = {"P": 0, "V": 1, "T": 2}
PF_TERRAIN_PLANE, PF_TERRAIN_HEIGHTFIELD = 0, 1

i32, u32, f32 = C.c_int32, C.c_uint32, C.c_float


class PfModel(C.Structure):
    _fields_ = [
        ("nb", i32), ("nd", i32), ("nfeet", i32), ("nsph", i32),
        ("parent", i32 * PF_MAX_BODIES), ("subtree", i32 * PF_MAX_BODIES), ("depth", i32 * PF_MAX_BODIES),
        ("dof_of_body", i32 * PF_MAX_BODIES), ("body_of_dof", i32 * PF_MAX_DOF),
        ("joint_pos", (f32 * 3) * PF_MAX_BODIES), ("joint_rot", (f32 * 9) * PF_MAX_BODIES),
        ("joint_axis", (f32 * 3) * PF_MAX_BODIES),
        ("mass", f32 * PF_MAX_BODIES), ("com", (f32 * 3) * PF_MAX_BODIES), ("inertia", (f32 * 6) * PF_MAX_BODIES),
        ("q_lo", f32 * PF_MAX_DOF), ("q_hi", f32 * PF_MAX_DOF),
        ("vel_limit", f32 * PF_MAX_DOF), ("effort_limit", f32 * PF_MAX_DOF),
        ("sph_body", i32 * PF_MAX_SPHERES), ("sph_pos", (f32 * 3) * PF_MAX_SPHERES),
        ("sph_radius", f32 * PF_MAX_SPHERES),
        ("feet_body", i32 * PF_MAX_FEET), ("n_term", i32), ("n_pen", i32),
        ("term_body", i32 * PF_MAX_TERM), ("pen_body", i32 * PF_MAX_PEN),
        ("ncap", i32), ("npair", i32), ("cap_s0", i32 * PF_MAX_CAPS), ("cap_s1", i32 * PF_MAX_CAPS),
        ("pair_a", i32 * PF_MAX_PAIRS), ("pair_b", i32 * PF_MAX_PAIRS),
    ]


class PfConfig(C.Structure):
    _fields_ = [
        ("num_envs", i32), ("env_index_base", i32), ("num_envs_global", i32),
        ("num_obs", i32), ("num_priv_obs", i32), ("num_actions", i32),
        ("control_type", i32), ("decimation", i32), ("terrain_type", i32), ("measure_heights", i32),
        ("num_height_points", i32), ("heading_command", i32), ("num_commands", i32), ("curriculum", i32),
        ("custom_origins", i32), ("max_terrain_level", i32), ("terrain_num_cols", i32),
        ("hf_rows", i32), ("hf_cols", i32), ("push_robots", i32), ("push_interval", i32), ("add_noise", i32),
        ("only_positive_rewards", i32), ("include_feet_height", i32), ("num_rewards", i32),
        ("reward_id", i32 * PF_MAX_REWARDS), ("reward_scale", f32 * PF_MAX_REWARDS), ("termination_scale", f32),
        ("sim_dt", f32), ("dt", f32), ("max_episode_length", f32), ("max_episode_length_s", f32),
        ("action_scale", f32), ("clip_actions", f32), ("clip_obs", f32),
        ("p_gains", f32 * PF_MAX_DOF), ("d_gains", f32 * PF_MAX_DOF), ("default_dof_pos", f32 * PF_MAX_DOF),
        ("torque_limits", f32 * PF_MAX_DOF), ("dof_pos_soft_lo", f32 * PF_MAX_DOF),
        ("dof_pos_soft_hi", f32 * PF_MAX_DOF), ("dof_vel_limits", f32 * PF_MAX_DOF),
        ("obs_scale_lin_vel", f32), ("obs_scale_ang_vel", f32), ("obs_scale_dof_pos", f32),
        ("obs_scale_dof_vel", f32), ("obs_scale_height", f32), ("commands_scale", f32 * 3),
        ("obs_noise_vec", f32 * PF_MAX_OBS), ("priv_height_noise", f32), ("base_init_state", f32 * 13),
        ("height_px", f32 * PF_MAX_HPTS), ("height_py", f32 * PF_MAX_HPTS),
        ("hscale", f32), ("vscale", f32), ("border", f32), ("terrain_length", f32),
        ("base_height_target", f32), ("tracking_sigma", f32), ("max_contact_force", f32),
        ("soft_dof_vel_limit", f32), ("soft_torque_limit", f32),
        ("min_feet_air_time", f32), ("max_feet_air_time", f32), ("min_feet_distance", f32),
        ("max_push_force", f32), ("gravity", f32 * 3),
        ("terrain_friction", f32), ("terrain_restitution", f32), ("contact_offset", f32),
        ("bounce_threshold", f32), ("max_depenetration_vel", f32), ("solver_iterations", i32),
        ("seed_lo", u32), ("seed_hi", u32),
        ("env_kind", i32), ("resample_interval", i32), ("use_actuator_net", i32),
        ("cmd_lo", f32 * 4), ("cmd_hi", f32 * 4), ("max_push_vel", f32),
        ("wall_threshold", f32), ("self_collisions", i32), ("joint_friction_vel", f32),
    ]


class PfLayout(C.Structure):
    _fields_ = [(n, i32) for n in (
        "rec_floats", "rw_floats", "rec_used", "root", "dof_pos", "dof_vel", "actions", "last_actions", "last_dof_vel", "last_root_vel",
        "commands", "torques", "base_lin_vel", "base_ang_vel", "projected_gravity",
        "feet_air_time", "last_feet_air_time", "cur_max_feet_height", "last_max_feet_height", "feet_height",
        "contact_filt", "first_contact", "last_contacts", "contact_forces", "feet_pos", "ext_force", "env_origin",
        "friction", "added_mass", "added_com", "lstm_state", "episode_sums",
        "rand_floats", "r_obs", "r_priv", "r_dof", "r_xy", "r_vel", "r_level", "r_push", "r_cmd", "r_cmd_reset")]


vp = C.c_void_p


class PfBuffers(C.Structure):
    _fields_ = [(n, vp) for n in (
        "d_state", "d_episode_length", "d_terrain_level", "d_terrain_type", "d_reset", "d_time_out", "d_terminate",
        "d_obs", "d_priv", "d_rew", "d_measured_heights", "d_height_samples", "d_terrain_origins", "d_actuator_params",
        "d_reset_stats", "d_joint_friction", "d_contact_set")]


class PfInject(C.Structure):
    _fields_ = [(n, vp) for n in ("d_rand", "d_dof_pre", "d_dof_post", "d_root", "d_contact", "d_feet_pos")]


class PfPolicyHeads(C.Structure):
    """include/pointfoot_hip.h PfPolicyHeads: filled by ppo_act_hidden, consumed by pf_step_policy."""
    _fields_ = [(n, vp) for n in (
        "d_ha", "d_hc", "d_wa", "d_ba", "d_wc", "d_bc", "d_std", "d_inject_normal", "d_obs", "d_critic_obs", "d_actions_out",
        "s_obs", "s_critic_obs", "s_actions", "s_values", "s_log_prob", "s_mu", "s_sigma",
        "p_rew", "p_dones", "p_time_outs", "p_values", "p_s_rew", "p_s_done", "p_ep_ret", "p_ep_len", "p_ep_stats", "p_fin_ret", "p_fin_len")] + [
        ("p_gamma", f32), ("ka", i32), ("kc", i32), ("num_obs", i32), ("num_critic_obs", i32), ("num_actions", i32),
        ("seed_lo", u32), ("seed_hi", u32), ("counter", u32), ("env_base", u32)]


class PpoConfig(C.Structure):
    _fields_ = [
        ("num_envs", i32), ("num_steps", i32), ("num_obs", i32), ("num_critic_obs", i32), ("num_actions", i32),
        ("n_actor_layers", i32), ("n_critic_layers", i32),
        ("actor_dims", i32 * (PPO_MAX_LAYERS + 1)), ("critic_dims", i32 * (PPO_MAX_LAYERS + 1)),
        ("num_learning_epochs", i32), ("num_mini_batches", i32),
        ("use_clipped_value_loss", i32), ("adaptive_schedule", i32),
        ("clip_param", f32), ("value_loss_coef", f32), ("entropy_coef", f32),
        ("gamma", f32), ("lam", f32), ("desired_kl", f32), ("max_grad_norm", f32), ("learning_rate", f32),
        ("adam_beta1", f32), ("adam_beta2", f32), ("adam_eps", f32),
        ("world_size", i32), ("seed_lo", u32), ("seed_hi", u32), ("gemm_mode", i32),
    ]


class PpoBuffers(C.Structure):
    _fields_ = [(n, vp) for n in (
        "d_params", "d_grads", "d_adam_m", "d_adam_v", "d_obs", "d_critic_obs", "d_actions", "d_rewards", "d_dones",
        "d_values", "d_returns", "d_advantages", "d_log_prob", "d_mu", "d_sigma", "d_perm", "d_workspace")] + [
        ("workspace_floats", C.c_int64), ("d_ep_return", vp), ("d_ep_length", vp), ("d_ep_stats", vp), ("d_ep_fin", vp), ("d_ep_ring", vp),
        ("d_scalars", vp)]


_LIB = None
LIB_NAME = "libpointfoot_hip.so"


def lib_path():
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), LIB_NAME)


def load():
    """Load libpointfoot_hip.so (built by __graft_entry__.build()).  Raises if absent / ABI mismatch."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} not found: the HIP extension is required (no CPU fallback). "
            "Build it with `python -c 'import __graft_entry__ as g; g.build()'` from the repo root.")
    lib = C.CDLL(path)
    lib.pf_last_error.restype = C.c_char_p
    lib.pf_abi_sizes.argtypes = [C.POINTER(C.c_int64), i32]
    sizes = (C.c_int64 * 8)()
    lib.pf_abi_sizes(sizes, 8)
    mine = [C.sizeof(t) for t in (PfModel, PfConfig, PfLayout, PfBuffers, PfInject, PpoConfig, PpoBuffers, PfPolicyHeads)]
    if list(sizes[:8]) != mine:
        raise RuntimeError(f"ABI mismatch between pointfoot_hip.h {list(sizes[:8])} and _abi.py {mine}")
    lib.pf_layout.argtypes = [C.POINTER(PfModel), C.POINTER(PfConfig), C.POINTER(PfLayout)]
    lib.pf_layout.restype = i32
    lib.pf_create.argtypes = [C.POINTER(PfModel), C.POINTER(PfConfig), i32]
    lib.pf_create.restype = vp
    lib.pf_destroy.argtypes = [vp]
    lib.pf_bind_buffers.argtypes = [vp, C.POINTER(PfBuffers)]
    lib.pf_update_config.argtypes = [vp, C.POINTER(PfConfig)]
    lib.pf_measure_heights.argtypes = [vp, vp]
    lib.pf_measure_heights.restype = i32
    lib.pf_set_reset_stats.argtypes = [vp, vp]
    lib.pf_set_reset_stats.restype = i32
    lib.pf_set_reset_stats_next.argtypes = [vp, vp]
    lib.pf_set_reset_stats_next.restype = i32
    lib.pf_step.argtypes = [vp, vp, C.c_int64, C.POINTER(PfInject), vp]
    lib.pf_step_policy.argtypes = [vp, C.POINTER(PfPolicyHeads), C.c_int64, vp]
    lib.pf_step_policy.restype = i32
    lib.pf_reset_all.argtypes = [vp, C.c_int64, C.POINTER(PfInject), vp]
    lib.pf_reset_idx.argtypes = [vp, vp, C.c_int64, C.POINTER(PfInject), vp]
    lib.pf_compute_observations.argtypes = [vp, C.c_int64, C.POINTER(PfInject), vp]
    _bind_ppo(lib)
    for fn in ("pf_bind_buffers", "pf_update_config", "pf_step", "pf_step_policy", "pf_reset_all", "pf_reset_idx", "pf_compute_observations"):
        getattr(lib, fn).restype = i32
    _LIB = lib
    return lib


def _bind_ppo(lib):
    lib.ppo_create.argtypes = [C.POINTER(PpoConfig), i32]
    lib.ppo_create.restype = vp
    lib.ppo_destroy.argtypes = [vp]
    lib.ppo_bind_buffers.argtypes = [vp, C.POINTER(PpoBuffers)]
    lib.ppo_num_params.argtypes = [C.POINTER(PpoConfig)]
    lib.ppo_num_params.restype = C.c_int64
    lib.ppo_param_offsets.argtypes = [C.POINTER(PpoConfig), C.POINTER(C.c_int64), i32]
    lib.ppo_param_offsets.restype = i32
    lib.ppo_workspace_floats.argtypes = [C.POINTER(PpoConfig)]
    lib.ppo_workspace_floats.restype = C.c_int64
    lib.ppo_act.argtypes = [vp, i32, vp, vp, vp, C.c_int64, vp, vp]
    lib.ppo_act_hidden.argtypes = [vp, i32, vp, vp, vp, C.c_int64, vp, C.POINTER(PfPolicyHeads), vp]
    lib.ppo_act_hidden.restype = i32
    lib.ppo_act_inference.argtypes = [vp, vp, i32, vp, vp]
    lib.ppo_process_env_step.argtypes = [vp, i32, vp, vp, vp, vp]
    lib.ppo_process_env_step_deferred.argtypes = [vp, i32, vp, vp, vp, vp]
    lib.ppo_compute_returns.argtypes = [vp, vp, vp]
    lib.ppo_normalize_advantages.argtypes = [vp, vp]
    lib.ppo_minibatch_backward.argtypes = [vp, i32, vp]
    lib.ppo_begin_update.argtypes = [vp, vp]
    lib.ppo_minibatch_apply.argtypes = [vp, vp]
    lib.ppo_minibatch_step.argtypes = [vp, i32, vp]
    lib.ppo_minibatch_backward_part.argtypes = [vp, i32, i32, vp]
    lib.ppo_minibatch_backward_part.restype = i32
    lib.ppo_grad_groups.argtypes = [vp, C.POINTER(C.c_int64)]
    lib.ppo_grad_groups.restype = i32
    lib.ppo_supports_parts.argtypes = [vp]
    lib.ppo_supports_parts.restype = i32
    for fn in ("ppo_bind_buffers", "ppo_act", "ppo_act_inference", "ppo_process_env_step", "ppo_process_env_step_deferred", "ppo_compute_returns",
               "ppo_normalize_advantages", "ppo_begin_update", "ppo_minibatch_backward", "ppo_minibatch_apply", "ppo_minibatch_step"):
        getattr(lib, fn).restype = i32


def check(rc, lib=None):
    if rc != 0:
        lib = lib or load()
        raise RuntimeError("pointfoot_hip: " + (lib.pf_last_error() or b"unknown error").decode())


EXPORTED_SYMBOLS = [
    "pf_last_error", "pf_abi_sizes", "pf_layout", "pf_create", "pf_destroy", "pf_bind_buffers", "pf_set_reset_stats", "pf_set_reset_stats_next", "pf_measure_heights",
    "pf_update_config", "pf_step", "pf_step_policy", "pf_reset_all", "pf_reset_idx", "pf_compute_observations",
    "ppo_create", "ppo_destroy", "ppo_bind_buffers", "ppo_num_params", "ppo_param_offsets", "ppo_workspace_floats", "ppo_act", "ppo_act_hidden",
    "ppo_act_inference", "ppo_process_env_step", "ppo_process_env_step_deferred", "ppo_compute_returns", "ppo_normalize_advantages",
    "ppo_begin_update", "ppo_minibatch_backward", "ppo_minibatch_backward_part", "ppo_grad_groups", "ppo_supports_parts", "ppo_minibatch_apply", "ppo_minibatch_step"]
