"""cfg -> C-ABI structs (PfModel / PfConfig) for the fused env-step kernel.

Restates the *setup-time* derivations of the reference env (rows a14 of SURVEY.md §8):
  _parse_cfg (point_foot.py:999-1009), _init_buffers PD gains / defaults (:777-794),
  _process_dof_props soft limits (:449-464), _get_noise_scale_vec (:652-695),
  _prepare_reward_function (:796-820), _init_height_points (:1032-1046), body-index lookup (:910-968).
"""
import ctypes as C

import numpy as np

from pointfoot_amd import LEGGED_GYM_ROOT_DIR, _abi
from pointfoot_amd.utils.helpers import class_to_dict
from pointfoot_amd.utils.urdf import load_model


def build_model(cfg):
    """Returns (PfModel, info dict with names / index lists)."""
    path = cfg.asset.file.format(LEGGED_GYM_ROOT_DIR=LEGGED_GYM_ROOT_DIR)
    t = load_model(path)
    nb, nd = len(t["body_names"]), len(t["dof_names"])
    if nb > _abi.PF_MAX_BODIES or nd > _abi.PF_MAX_DOF:
        raise ValueError(f"model too large for the kernel build: {nb} bodies / {nd} dofs")
    names = t["body_names"]
    feet = [i for i, s in enumerate(names) if cfg.asset.foot_name in s]
    pen, term = [], []
    for name in cfg.asset.penalize_contacts_on:
        pen.extend([i for i, s in enumerate(names) if name in s])
    for name in cfg.asset.terminate_after_contacts_on:
        term.extend([i for i, s in enumerate(names) if name in s])
    m = _abi.PfModel()
    m.nb, m.nd, m.nfeet = nb, nd, len(feet)
    for i in range(nb):
        m.parent[i] = t["parent"][i]
        m.subtree[i] = t["subtree"][i]
        m.depth[i] = t["depth"][i]
        m.dof_of_body[i] = t["dof_of_body"][i]
        m.mass[i] = t["mass"][i]
        for k in range(3):
            m.joint_pos[i][k] = t["joint_pos"][i][k]
            m.joint_axis[i][k] = t["joint_axis"][i][k]
            m.com[i][k] = t["com"][i][k]
        for k in range(9):
            m.joint_rot[i][k] = t["joint_rot"][i][k]
        for k in range(6):
            m.inertia[i][k] = t["inertia"][i][k]
    for k in range(nd):
        m.body_of_dof[k] = t["body_of_dof"][k]
        lo, hi = t["q_lo"][k], t["q_hi"][k]
        m.q_lo[k], m.q_hi[k] = (lo, hi) if hi > lo else (-1e9, 1e9)      # continuous joints (ANYmal): no position limit
        m.vel_limit[k], m.effort_limit[k] = t["vel_limit"][k], t["effort_limit"][k]
    # collision spheres: feet first so they survive the per-robot contact cap
    sph = sorted(t["spheres"], key=lambda s: (0 if s[0] in feet else 1))
    if len(sph) > _abi.PF_MAX_SPHERES:
        raise ValueError("too many collision spheres")
    m.nsph = len(sph)
    for s, (b, p, r) in enumerate(sph):
        m.sph_body[s] = b
        m.sph_radius[s] = r
        for k in range(3):
            m.sph_pos[s][k] = p[k]
    for f, b in enumerate(feet):
        m.feet_body[f] = b
    # self-collision geometry (asset.self_collisions == 0 enables it: pointfoot_rough_config.py:107, point_foot.py:942).
    # Capsules = the two equal-radius end spheres a cylinder / limb box became (urdf.py), or a single sphere; pairs =
    # capsules on different branches of the tree.  PhysX never collides a parent with its child; pairs further apart on ONE
    # chain (hip-shank, thigh-foot) and pairs with the base are a MODELLING LIMIT of this build, not a property of the robots:
    # PF_P441A's joint limits keep a leg's own links apart, ANYmal's continuous joints (q_lo/q_hi = -+1e9 above) do not, so a
    # leg folded through itself or into the trunk passes without a self-contact here where PhysX would stop it.  (The physics
    # is parity-unpinned either way; anymal_c_flat is the only task affected and is not a BASELINE configuration.)
    # Among the LeggedRobot tasks anymal_c_flat asks for it (anymal_c_flat_config.py:42, applied at legged_robot.py:680): 12 capsules /
    # 54 pairs on a quadruped, the kernel instance built for them; config 5 (anymal_c_rough) switches it off (anymal_c_rough_config.py:77).
    caps, pairs = [], []
    if getattr(cfg.asset, "self_collisions", 1) == 0:
        by_body = {}
        for s, (b, p, r) in enumerate(sph):
            if b != 0 and r > 0.0:
                by_body.setdefault(b, []).append(s)
        for b in sorted(by_body):
            ss = by_body[b]
            if len(ss) == 2 and sph[ss[0]][2] == sph[ss[1]][2]:
                caps.append((ss[0], ss[1]))
            else:
                caps.extend((s, s) for s in ss)

        def related(a, b):      # a is an ancestor of b (or the same body)
            return a <= b < a + t["subtree"][a]
        for i in range(len(caps)):
            for j in range(i + 1, len(caps)):
                ba, bb = sph[caps[i][0]][0], sph[caps[j][0]][0]
                if not related(ba, bb) and not related(bb, ba):
                    pairs.append((i, j))
        if len(caps) > _abi.PF_MAX_CAPS or len(pairs) > _abi.PF_MAX_PAIRS:
            raise ValueError(f"self-collision geometry too large for the kernel build: {len(caps)} capsules / {len(pairs)} pairs")
    m.ncap, m.npair = len(caps), len(pairs)
    for i, (a, b) in enumerate(caps):
        m.cap_s0[i], m.cap_s1[i] = a, b
    for i, (a, b) in enumerate(pairs):
        m.pair_a[i], m.pair_b[i] = a, b
    m.n_term, m.n_pen = len(term), len(pen)
    for i, b in enumerate(term):
        m.term_body[i] = b
    for i, b in enumerate(pen):
        m.pen_body[i] = b
    info = dict(body_names=names, dof_names=t["dof_names"], q_lo_raw=t["q_lo"], q_hi_raw=t["q_hi"], feet_indices=feet, penalised_contact_indices=pen,
                termination_contact_indices=term, base_mass=t["mass"][0])
    return m, info


def noise_scale_vec(cfg, num_dof, kind=_abi.PF_KIND_POINTFOOT):
    """_get_noise_scale_vec: PointFoot (point_foot.py:652-695, including its misalignment with the obs layout) or
    LeggedRobot (legged_robot.py:456-478, aligned, hard-coded for 12 DOF in the reference)."""
    if kind == _abi.PF_KIND_LEGGED_ROBOT:
        no = cfg.env.num_observations
        vec = np.zeros(no, dtype=np.float32)
        ns, lvl, sc = cfg.noise.noise_scales, cfg.noise.noise_level, cfg.normalization.obs_scales
        vec[:3] = ns.lin_vel * lvl * sc.lin_vel
        vec[3:6] = ns.ang_vel * lvl * sc.ang_vel
        vec[6:9] = ns.gravity * lvl
        vec[9:12] = 0.
        vec[12:12 + num_dof] = ns.dof_pos * lvl * sc.dof_pos
        vec[12 + num_dof:12 + 2 * num_dof] = ns.dof_vel * lvl * sc.dof_vel
        vec[12 + 2 * num_dof:12 + 3 * num_dof] = 0.
        if cfg.terrain.measure_heights:
            vec[12 + 3 * num_dof:no] = ns.height_measurements * lvl * sc.height_measurements
        return vec, None
    no = cfg.env.num_propriceptive_obs
    vec = np.zeros(no, dtype=np.float32)
    ns, lvl, sc = cfg.noise.noise_scales, cfg.noise.noise_level, cfg.normalization.obs_scales
    vec[:3] = ns.lin_vel * lvl * sc.lin_vel
    vec[3:6] = ns.ang_vel * lvl * sc.ang_vel
    vec[6:9] = ns.gravity * lvl
    ce = 9 + cfg.commands.num_commands
    vec[9:ce] = 0.
    pe = ce + num_dof
    vec[ce:pe] = ns.dof_pos * lvl * sc.dof_pos
    ve = pe + num_dof
    vec[pe:ve] = ns.dof_vel * lvl * sc.dof_vel
    ae = ve + cfg.env.num_actions
    vec[ve:ae] = 0.
    npts = len(cfg.terrain.measured_points_x) * len(cfg.terrain.measured_points_y)
    hnoise = ns.height_measurements * lvl * sc.height_measurements
    if cfg.terrain.measure_heights_actor:
        vec[ae:ae + npts] = hnoise
    priv = None
    if cfg.env.num_privileged_obs is not None:
        priv = np.zeros(cfg.env.num_privileged_obs - no, dtype=np.float32)
        if cfg.terrain.measure_heights_critic:
            priv[:npts] = hnoise
    return vec, priv


def reward_table(cfg, dt):
    """_prepare_reward_function (:796-820): drop zero scales, multiply by dt, keep dir() (alphabetical) order."""
    scales = class_to_dict(cfg.rewards.scales)
    out = {}
    for key in list(scales.keys()):
        if scales[key] != 0:
            out[key] = scales[key] * dt
    names = [k for k in out if k != "termination"]
    return names, out


def build_config(cfg, model, info, sim_params, num_envs, *, env_index_base=0, num_envs_global=None,
                 hf_shape=None, mean_base_mass=None, seed=1, kind=_abi.PF_KIND_POINTFOOT):
    nd = model.nd
    LR = kind == _abi.PF_KIND_LEGGED_ROBOT
    c = _abi.PfConfig()
    dt = cfg.control.decimation * sim_params.dt
    c.env_kind = kind
    c.num_envs, c.env_index_base = num_envs, env_index_base
    c.num_envs_global = num_envs_global if num_envs_global is not None else num_envs
    c.num_obs = cfg.env.num_observations if LR else cfg.env.num_propriceptive_obs
    c.num_priv_obs = 0 if LR else (cfg.env.num_privileged_obs or 0)
    if LR and cfg.env.num_privileged_obs is not None:
        raise NotImplementedError("LeggedRobot with privileged observations is not built (no reference task uses it)")
    c.num_actions = cfg.env.num_actions
    if cfg.control.control_type not in _abi.PF_CTRL:
        raise NameError(f"Unknown controller type: {cfg.control.control_type}")
    c.control_type = _abi.PF_CTRL[cfg.control.control_type]
    c.decimation = cfg.control.decimation
    mesh = cfg.terrain.mesh_type
    if mesh not in (None, 'plane', 'heightfield', 'trimesh'):
        raise ValueError("Terrain mesh type not recognised. Allowed types are [None, plane, heightfield, trimesh]")
    rough = mesh in ('heightfield', 'trimesh')
    c.terrain_type = _abi.PF_TERRAIN_HEIGHTFIELD if rough else _abi.PF_TERRAIN_PLANE
    if LR:
        c.measure_heights = int(bool(cfg.terrain.measure_heights))
    else:
        c.measure_heights = int(bool(cfg.terrain.measure_heights_critic or cfg.terrain.measure_heights_actor))
        if cfg.terrain.measure_heights_actor:
            # point_foot.py:361: the 121 measurements follow the 27 proprioceptive terms; :340-345 adds the actor's noise buffer
            # to the front of the privileged buffer, which only lines up when that buffer has the same layout or is absent
            npts = len(cfg.terrain.measured_points_x) * len(cfg.terrain.measured_points_y)
            nprop = 9 + 2 * nd + cfg.env.num_actions
            if cfg.env.num_propriceptive_obs != nprop + npts:
                raise RuntimeError(f"obs_buf size ({nprop + npts}) does not match num_obs ({cfg.env.num_propriceptive_obs})")   # :363-365
            npriv = cfg.env.num_privileged_obs
            if npriv is not None and not (cfg.terrain.measure_heights_critic and npriv == cfg.env.num_propriceptive_obs):
                raise NotImplementedError("measure_heights_actor with a privileged buffer of a different layout "
                                          "(the reference's own noise concatenation, point_foot.py:340-345, fails on it)")
    px, py = cfg.terrain.measured_points_x, cfg.terrain.measured_points_y
    H = len(px) * len(py) if c.measure_heights else 0
    if H > _abi.PF_MAX_HPTS:
        raise ValueError("too many height points")
    c.num_height_points = H
    if c.measure_heights:     # meshgrid(x, y) 'ij' flattened: x-major (:1038-1045)
        k = 0
        for xv in px:
            for yv in py:
                c.height_px[k], c.height_py[k] = xv, yv
                k += 1
    c.heading_command = int(cfg.commands.heading_command)
    c.num_commands = cfg.commands.num_commands
    c.curriculum = int(bool(cfg.terrain.curriculum) and rough)
    c.custom_origins = int(rough)
    c.max_terrain_level = cfg.terrain.num_rows
    c.terrain_num_cols = cfg.terrain.num_cols
    if hf_shape is not None:
        c.hf_rows, c.hf_cols = hf_shape
    c.push_robots = int(cfg.domain_rand.push_robots)
    c.push_interval = int(np.ceil(cfg.domain_rand.push_interval_s / dt))
    c.add_noise = int(cfg.noise.add_noise)
    c.only_positive_rewards = int(cfg.rewards.only_positive_rewards)
    c.include_feet_height = 0 if LR else int(any("feet_height" in s for s in dir(cfg.rewards.scales) if not s.startswith("__")))
    names, scales = reward_table(cfg, dt)
    if len(names) > _abi.PF_MAX_REWARDS - 1:
        raise ValueError("too many reward terms")
    c.num_rewards = len(names)
    for i, n in enumerate(names):
        key = "stumble" if n == "stumble" else n
        if key not in _abi.REWARD_IDS:
            raise AttributeError(f"'PointFoot' object has no attribute '_reward_{n}'")   # getattr failure at :815
        c.reward_id[i] = _abi.REWARD_IDS[key]
        c.reward_scale[i] = scales[n]
    c.termination_scale = scales.get("termination", 0.0)
    c.sim_dt, c.dt = sim_params.dt, dt
    c.max_episode_length_s = cfg.env.episode_length_s
    c.max_episode_length = float(np.ceil(cfg.env.episode_length_s / dt))
    c.action_scale = cfg.control.action_scale
    c.clip_actions = cfg.normalization.clip_actions
    c.clip_obs = cfg.normalization.clip_observations
    dof_names = info["dof_names"]
    for i, name in enumerate(dof_names):     # :779-793
        c.default_dof_pos[i] = cfg.init_state.default_joint_angles[name]
        for key in cfg.control.stiffness.keys():
            if key in name:
                c.p_gains[i] = cfg.control.stiffness[key]
                c.d_gains[i] = cfg.control.damping[key]
        c.torque_limits[i] = model.effort_limit[i]
        c.dof_vel_limits[i] = model.vel_limit[i]
        lo, hi = np.float32(info["q_lo_raw"][i]), np.float32(info["q_hi_raw"][i])
        mid, rng = (lo + hi) / np.float32(2), hi - lo
        c.dof_pos_soft_lo[i] = mid - np.float32(0.5) * rng * np.float32(cfg.rewards.soft_dof_pos_limit)
        c.dof_pos_soft_hi[i] = mid + np.float32(0.5) * rng * np.float32(cfg.rewards.soft_dof_pos_limit)
    sc = cfg.normalization.obs_scales
    c.obs_scale_lin_vel, c.obs_scale_ang_vel = sc.lin_vel, sc.ang_vel
    c.obs_scale_dof_pos, c.obs_scale_dof_vel, c.obs_scale_height = sc.dof_pos, sc.dof_vel, sc.height_measurements
    c.commands_scale[0], c.commands_scale[1], c.commands_scale[2] = sc.lin_vel, sc.lin_vel, sc.ang_vel
    vec, priv = noise_scale_vec(cfg, nd, kind)
    for i, v in enumerate(vec):
        c.obs_noise_vec[i] = v
    c.priv_height_noise = float(priv[0]) if (priv is not None and len(priv) and c.measure_heights) else 0.0
    bis = cfg.init_state.pos + cfg.init_state.rot + cfg.init_state.lin_vel + cfg.init_state.ang_vel
    for i, v in enumerate(bis):
        c.base_init_state[i] = v
    c.hscale, c.vscale = cfg.terrain.horizontal_scale, cfg.terrain.vertical_scale
    c.border = cfg.terrain.border_size
    c.terrain_length = cfg.terrain.terrain_length
    r = cfg.rewards
    c.base_height_target, c.tracking_sigma, c.max_contact_force = r.base_height_target, r.tracking_sigma, r.max_contact_force
    c.soft_dof_vel_limit, c.soft_torque_limit = r.soft_dof_vel_limit, r.soft_torque_limit
    c.min_feet_air_time = getattr(r, "min_feet_air_time", 0.0)
    c.max_feet_air_time = getattr(r, "max_feet_air_time", 0.0)
    c.min_feet_distance = getattr(r, "min_feet_distance", 0.0)
    mbm = mean_base_mass if mean_base_mass is not None else info["base_mass"]
    c.max_push_force = mbm * cfg.domain_rand.max_push_vel_xy / sim_params.dt     # :592-596
    for i in range(3):
        c.gravity[i] = cfg.sim.gravity[i]
    c.terrain_friction = cfg.terrain.static_friction
    c.terrain_restitution = cfg.terrain.restitution
    px_ = cfg.sim.physx
    c.contact_offset = px_.contact_offset
    c.bounce_threshold = px_.bounce_threshold_velocity
    c.max_depenetration_vel = px_.max_depenetration_velocity
    c.solver_iterations = px_.num_position_iterations
    c.seed_lo, c.seed_hi = seed & 0xFFFFFFFF, 0x9E3779B9
    # trimesh: steps steeper than slope_treshold are vertical faces (terrain.py:69-73); a 'heightfield' stays bilinear
    c.wall_threshold = float(cfg.terrain.slope_treshold) * float(cfg.terrain.horizontal_scale) if mesh == 'trimesh' else 0.0
    c.self_collisions = int(model.npair > 0)
    c.joint_friction_vel = 0.1
    if LR:      # legged_robot.py:323, 343-349, 417; anymal.py:52
        c.resample_interval = int(cfg.commands.resampling_time / dt)
        rg = cfg.commands.ranges
        for i, key in enumerate(("lin_vel_x", "lin_vel_y", "ang_vel_yaw", "heading")):
            c.cmd_lo[i], c.cmd_hi[i] = getattr(rg, key)
        c.max_push_vel = cfg.domain_rand.max_push_vel_xy
        c.use_actuator_net = int(bool(getattr(cfg.control, "use_actuator_network", False)))
    return c, names


def ptr(t):
    """device/host pointer of a torch tensor or numpy array as c_void_p (None -> NULL)."""
    if t is None:
        return C.c_void_p(None)
    if isinstance(t, np.ndarray):
        return C.c_void_p(t.ctypes.data)
    return C.c_void_p(t.data_ptr())
