#!/usr/bin/env python3
"""Generate golden vectors from the reference's own Python layer.

RUNS ONLY IN THE BUILD CONTAINER (needs /root/reference).  Nothing from the
reference is copied: the script imports `legged_gym.envs.pointfoot.point_foot`
from /root/reference, drives `PointFoot.step()` / `post_physics_step()` on the
seeded synthetic sim state of `synth.py`, and stores inputs + outputs as small
.npz fixtures next to this file.

Isaac Gym and rsl_rl are not installed (closed binary / un-vendored), so a
stub namespace is injected into sys.modules:
  * `isaacgym.torch_utils` — the seven public quaternion/rand helpers, restated
    here from their documented semantics;
  * `isaacgym.gymapi/gymtorch/gymutil/terrain_utils`, `rsl_rl.*` — inert.
The physics engine itself is replaced by a fake `gym` whose `simulate()` writes
the next synthetic DOF snapshot and whose `refresh_actor_root_state_tensor()`
writes the next synthetic root/contact/body snapshot.  Hence these fixtures
pin the *Python layer* (reference rows a1,a2,a4-a13 of SURVEY.md §8); physics
(a3) and PPO (a16-a18) have no reference artefact: parity unpinned.

Random draws (`torch.rand`, `torch.rand_like`, `torch.randint_like`) are taken
from a seeded generator and LOGGED per call so the tests can inject the very
same numbers into the oracle and the HIP kernel.
"""
import os
import sys
import types

sys.dont_write_bytecode = True
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import synth  # noqa: E402
import variants  # noqa: E402

REF = os.environ.get("POINTFOOT_REFERENCE", "/root/reference")


# ----------------------------------------------------------------------------- stubs
def _install_stubs():
    ig = types.ModuleType("isaacgym")
    for sub in ["gymtorch", "gymapi", "gymutil", "torch_utils", "terrain_utils"]:
        m = types.ModuleType("isaacgym." + sub)
        setattr(ig, sub, m)
        sys.modules["isaacgym." + sub] = m
    sys.modules["isaacgym"] = ig
    tu = sys.modules["isaacgym.torch_utils"]

    def quat_rotate(q, v):
        q_w = q[:, -1]
        q_vec = q[:, :3]
        a = v * (2.0 * q_w ** 2 - 1.0).unsqueeze(-1)
        b = torch.cross(q_vec, v, dim=-1) * q_w.unsqueeze(-1) * 2.0
        c = q_vec * torch.bmm(q_vec.view(q.shape[0], 1, 3), v.view(q.shape[0], 3, 1)).squeeze(-1) * 2.0
        return a + b + c

    def quat_rotate_inverse(q, v):
        q_w = q[:, -1]
        q_vec = q[:, :3]
        a = v * (2.0 * q_w ** 2 - 1.0).unsqueeze(-1)
        b = torch.cross(q_vec, v, dim=-1) * q_w.unsqueeze(-1) * 2.0
        c = q_vec * torch.bmm(q_vec.view(q.shape[0], 1, 3), v.view(q.shape[0], 3, 1)).squeeze(-1) * 2.0
        return a - b + c

    def quat_apply(a, b):
        shape = b.shape
        a = a.reshape(-1, 4)
        b = b.reshape(-1, 3)
        xyz = a[:, :3]
        t = xyz.cross(b, dim=-1) * 2
        return (b + a[:, 3:] * t + xyz.cross(t, dim=-1)).view(shape)

    def normalize(x, eps: float = 1e-9):
        return x / x.norm(p=2, dim=-1).clamp(min=eps, max=None).unsqueeze(-1)

    def torch_rand_float(lower, upper, shape, device):
        return (upper - lower) * torch.rand(*shape, device=device) + lower

    def to_torch(x, dtype=torch.float, device="cpu", requires_grad=False):
        return torch.tensor(x, dtype=dtype, device=device, requires_grad=requires_grad)

    def get_axis_params(value, axis_idx, x_value=0.0, dtype=float, n_dims=3):
        zs = np.zeros((n_dims,))
        zs[axis_idx] = 1.0
        params = np.where(zs == 1.0, value, zs)
        params[0] = x_value
        return list(params.astype(dtype))

    for f in (quat_rotate, quat_rotate_inverse, quat_apply, normalize, torch_rand_float, to_torch, get_axis_params):
        setattr(tu, f.__name__, f)
    tu.torch = torch
    tu.np = np
    tu.__all__ = ["quat_rotate", "quat_rotate_inverse", "quat_apply", "normalize", "torch_rand_float",
                  "to_torch", "get_axis_params", "torch", "np"]

    gt = sys.modules["isaacgym.gymtorch"]
    gt.wrap_tensor = lambda t: t
    gt.unwrap_tensor = lambda t: t
    ga = sys.modules["isaacgym.gymapi"]
    ga.ENV_SPACE = 0
    gu = sys.modules["isaacgym.gymutil"]
    gu.parse_device_str = lambda s: ("cpu", 0)

    rr = types.ModuleType("rsl_rl")
    env = types.ModuleType("rsl_rl.env")
    run = types.ModuleType("rsl_rl.runners")
    env.VecEnv = object
    run.OnPolicyRunner = object
    sys.modules.update({"rsl_rl": rr, "rsl_rl.env": env, "rsl_rl.runners": run})


# ----------------------------------------------------------------------------- rng logging
class RandLog:
    """Replaces torch.rand / rand_like / randint_like with a seeded, logged source."""

    def __init__(self, seed):
        self.gen = torch.Generator().manual_seed(seed)
        self.calls = []

    def rand(self, *shape, device=None, **kw):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)):
            shape = tuple(shape[0])
        t = _orig_rand(*shape, generator=self.gen)
        self.calls.append(("rand", tuple(shape), t.clone()))
        return t

    def rand_like(self, x, **kw):
        t = _orig_rand(*x.shape, generator=self.gen)
        self.calls.append(("rand_like", tuple(x.shape), t.clone()))
        return t

    def randint_like(self, x, high, **kw):
        t = _orig_randint(0, int(high), tuple(x.shape), generator=self.gen).to(x.dtype)
        self.calls.append(("randint_like", tuple(x.shape), t.clone()))
        return t

    def take(self):
        c, self.calls = self.calls, []
        return c


_orig_rand = torch.rand
_orig_randint = torch.randint


class FakeGym:
    """Inert stand-in for the gymapi handle; feeds synthetic snapshots."""

    def __init__(self):
        self.env = None
        self.dof_queue = []
        self.post = None

    def __getattr__(self, name):
        return lambda *a, **k: None

    def simulate(self, sim):
        s = self.dof_queue.pop(0)
        self.env.dof_state.copy_(torch.from_numpy(s.reshape(-1, 2)))

    def refresh_actor_root_state_tensor(self, sim):
        if self.post is None:
            return
        e = self.env
        e.root_states.copy_(torch.from_numpy(self.post["root"]))
        e.contact_forces.copy_(torch.from_numpy(self.post["contact"]))
        if "body" in self.post:
            e.rigid_body_states.copy_(torch.from_numpy(self.post["body"]))


BODY_NAMES = ["base_Link", "abad_L_Link", "hip_L_Link", "knee_L_Link", "foot_L_Link",
              "abad_R_Link", "hip_R_Link", "knee_R_Link", "foot_R_Link"]
DOF_NAMES = ["abad_L_Joint", "hip_L_Joint", "knee_L_Joint", "abad_R_Joint", "hip_R_Joint", "knee_R_Joint"]
DOF_LIMITS = [(-0.38397, 1.39626), (-1.012291, 1.518436), (-0.872665, 1.396263),
              (-1.39626, 0.38397), (-1.518436, 1.012291), (-1.396263, 0.872665)]
DOF_VEL = [20., 20., 20., 20., 20., 50.]
DOF_EFFORT = [80.] * 6


def build_env(task, n, rng_seed, control_type="P"):
    """Construct the reference PointFoot without running its Isaac-Gym-bound __init__."""
    from legged_gym.envs.pointfoot.point_foot import PointFoot
    from legged_gym.envs.pointfoot.mixed_terrain.pointfoot_rough_config import PointFootRoughCfg
    from legged_gym.envs.pointfoot.flat.pointfoot_flat_config import PointFootFlatCfg

    cfg = PointFootRoughCfg() if task.startswith("rough") else PointFootFlatCfg()
    cfg.env.num_envs = n
    cfg.control.control_type = control_type
    if task == "rough_hactor":      # point_foot.py:361: height measurements in the actor's observations too
        cfg.terrain.measure_heights_actor = True
        cfg.env.num_propriceptive_obs = 27 + len(cfg.terrain.measured_points_x) * len(cfg.terrain.measured_points_y)
    rough = task.startswith("rough")
    variants.apply(cfg, task)          # "..._allrew" / "..._onlypos" tasks (tests/golden/variants.py); a no-op for the others
    e = object.__new__(PointFoot)
    e.terminate = None
    e.cfg = cfg
    sp = types.SimpleNamespace(dt=cfg.sim.dt, use_gpu_pipeline=False)
    e.sim_params = sp
    e.height_samples = None
    e.debug_viz = False
    e.init_done = False
    e._parse_cfg()
    e.gym = FakeGym()
    e.gym.env = e
    e.sim = None
    e.physics_engine = None
    e.sim_device = "cpu"
    e.headless = True
    e.device = "cpu"
    e.num_envs = n
    e.num_obs = cfg.env.num_propriceptive_obs
    e.num_privileged_obs = cfg.env.num_privileged_obs
    e.num_actions = cfg.env.num_actions
    e.proprioceptive_obs_buf = torch.zeros(n, e.num_obs)
    e.rew_buf = torch.zeros(n)
    e.reset_buf = torch.ones(n, dtype=torch.long)
    e.episode_length_buf = torch.zeros(n, dtype=torch.long)
    e.time_out_buf = torch.zeros(n, dtype=torch.bool)
    e.privileged_obs_buf = torch.zeros(n, e.num_privileged_obs)
    e.extras = {}
    e.viewer = None
    e.enable_viewer_sync = True
    e.up_axis_idx = 2
    # --- what _create_envs() would have derived from the asset
    e.num_dof = e.num_dofs = 6
    e.num_bodies = 9
    e.dof_names = DOF_NAMES
    feet = [i for i, s in enumerate(BODY_NAMES) if cfg.asset.foot_name in s]
    pen = []
    for name in cfg.asset.penalize_contacts_on:
        pen.extend([i for i, s in enumerate(BODY_NAMES) if name in s])
    term = []
    for name in cfg.asset.terminate_after_contacts_on:
        term.extend([i for i, s in enumerate(BODY_NAMES) if name in s])
    e.feet_indices = torch.tensor(feet, dtype=torch.long)
    e.penalised_contact_indices = torch.tensor(pen, dtype=torch.long)
    e.termination_contact_indices = torch.tensor(term, dtype=torch.long)
    bis = cfg.init_state.pos + cfg.init_state.rot + cfg.init_state.lin_vel + cfg.init_state.ang_vel
    e.base_init_state = torch.tensor(bis, dtype=torch.float)
    nrs = np.random.RandomState(rng_seed + 100)
    e.base_mass = torch.tensor(8.567 + nrs.uniform(-1, 2, n), dtype=torch.float)
    # _process_dof_props for env 0
    e.dof_pos_limits = torch.zeros(6, 2)
    e.dof_vel_limits = torch.zeros(6)
    e.torque_limits = torch.zeros(6)
    for i in range(6):
        lo, hi = DOF_LIMITS[i]
        e.dof_pos_limits[i, 0] = lo
        e.dof_pos_limits[i, 1] = hi
        e.dof_vel_limits[i] = DOF_VEL[i]
        e.torque_limits[i] = DOF_EFFORT[i]
        m = (e.dof_pos_limits[i, 0] + e.dof_pos_limits[i, 1]) / 2
        r = e.dof_pos_limits[i, 1] - e.dof_pos_limits[i, 0]
        e.dof_pos_limits[i, 0] = m - 0.5 * r * cfg.rewards.soft_dof_pos_limit
        e.dof_pos_limits[i, 1] = m + 0.5 * r * cfg.rewards.soft_dof_pos_limit
    # origins / terrain
    if rough:
        hf = synth.synth_heightfield()
        to = synth.synth_terrain_origins()
        e.terrain = types.SimpleNamespace(cfg=cfg.terrain, env_length=cfg.terrain.terrain_length,
                                          env_width=cfg.terrain.terrain_width, env_origins=to)
        e.height_samples = torch.tensor(hf).view(synth.TOT_ROWS, synth.TOT_COLS)
        e.custom_origins = True
        e.terrain_levels = torch.from_numpy(nrs.randint(0, cfg.terrain.max_init_terrain_level + 1, n)).long()
        # a few at the top level so the ">= max -> random level" branch is hit
        e.terrain_levels[:4] = cfg.terrain.num_rows - 1
        e.terrain_types = torch.div(torch.arange(n), (n / cfg.terrain.num_cols), rounding_mode="floor").to(torch.long)
        e.max_terrain_level = cfg.terrain.num_rows
        e.terrain_origins = torch.from_numpy(to).to(torch.float)
        e.env_origins = torch.zeros(n, 3)
        e.env_origins[:] = e.terrain_origins[e.terrain_levels, e.terrain_types]
    else:
        e.custom_origins = False
        e.env_origins = torch.zeros(n, 3)
        num_cols = np.floor(np.sqrt(n))
        num_rows = np.ceil(n / num_cols)
        xx, yy = torch.meshgrid(torch.arange(num_rows), torch.arange(num_cols), indexing="ij")
        e.env_origins[:, 0] = cfg.env.env_spacing * xx.flatten()[:n]
        e.env_origins[:, 1] = cfg.env.env_spacing * yy.flatten()[:n]
    e._include_feet_height_rewards = e._check_if_include_feet_height_rewards()
    # sim tensors
    e._t_root = torch.zeros(n, 13)
    e._t_root[:, 6] = 1.0
    e._t_dof = torch.zeros(n * 6, 2)
    e._t_contact = torch.zeros(n * 9, 3)
    e._t_body = torch.zeros(n * 9, 13)
    g = e.gym
    g.acquire_actor_root_state_tensor = lambda sim: e._t_root
    g.acquire_dof_state_tensor = lambda sim: e._t_dof
    g.acquire_net_contact_force_tensor = lambda sim: e._t_contact
    g.acquire_rigid_body_state_tensor = lambda sim: e._t_body
    e._init_buffers()
    e._prepare_reward_function()
    e.init_done = True
    return e, cfg


def record_terms(e):
    """wrap the env's bound reward functions so that what each returned inside step() is kept (harness only: calling them a
    second time would advance LeggedRobot's feet-air-time state machine, legged_robot.py:882-893)"""
    rec = {}

    def wrap(i, fn):
        def w():
            v = fn()
            rec[i] = v.detach().clone().to(torch.float32).numpy()
            return v
        return w
    e.reward_functions = [wrap(i, f) for i, f in enumerate(e.reward_functions)]
    return rec


def scatter_rows(n, ids, t, width):
    out = np.zeros((n, width), dtype=np.float32)
    if len(ids):
        out[ids] = t.numpy().reshape(len(ids), width)
    return out


def run_sequence(task, n, steps, seed, control_type="P", add_noise=True, push_at=2):
    e, cfg = build_env(task, n, seed, control_type)
    rough = task.startswith("rough")
    term_rec = record_terms(e) if "allrew" in task else None
    rl = RandLog(seed)
    torch.rand = rl.rand
    torch.rand_like = rl.rand_like
    torch.randint_like = rl.randint_like
    torch.save = lambda *a, **k: None          # reference point_foot.py:334 side effect
    e.add_noise = add_noise
    rs = np.random.RandomState(seed)
    origins = e.env_origins.numpy().astype(np.float64)
    feet = e.feet_indices.tolist()
    # initial sim state + runner-style random episode lengths
    s0 = synth.synth_sim_state(rs, n, 9, 6, origins, feet, rough)
    e.root_states.copy_(torch.from_numpy(s0["root"]))
    e.dof_state.copy_(torch.from_numpy(s0["dof"].reshape(-1, 2)))
    e.contact_forces.copy_(torch.from_numpy(s0["contact"]))
    e.rigid_body_states.copy_(torch.from_numpy(s0["body"]))
    e.episode_length_buf = torch.from_numpy(rs.randint(0, 1000, n)).long()
    e.episode_length_buf[n // 2: n // 2 + 5] = 1000      # -> time-out on the first step (> 1000)
    e.common_step_counter = int(cfg.domain_rand.push_interval) - push_at
    e.commands[:, 0] = torch.from_numpy(rs.uniform(-1, 1, n)).float()
    e.commands[:, 1] = torch.from_numpy(rs.uniform(-0.3, 0.3, n)).float()
    if cfg.commands.heading_command:
        e.commands[:, 3] = torch.from_numpy(rs.uniform(-3.14, 3.14, n)).float()
    else:
        e.commands[:, 2] = torch.from_numpy(rs.uniform(-1, 1, n)).float()
    # non-trivial carried buffers
    e.last_actions[:] = torch.from_numpy(rs.normal(0, 1, (n, 6))).float()
    e.last_dof_vel[:] = torch.from_numpy(rs.normal(0, 3, (n, 6))).float()
    e.feet_air_time[:] = torch.from_numpy(rs.uniform(0, 0.9, (n, 2))).float() * (torch.from_numpy(rs.uniform(size=(n, 2))) > 0.3)
    e.last_feet_air_time[:] = torch.from_numpy(rs.uniform(0, 0.9, (n, 2))).float()
    e.contact_filt[:] = torch.from_numpy(rs.uniform(size=(n, 2)) > 0.5)
    e.first_contact[:] = torch.from_numpy(rs.uniform(size=(n, 2)) > 0.7)
    e.last_contacts[:] = torch.from_numpy(rs.uniform(size=(n, 2)) > 0.5)
    e.current_max_feet_height[:] = torch.from_numpy(rs.uniform(0, 0.2, (n, 2))).float()
    e.last_max_feet_height[:] = torch.from_numpy(rs.uniform(0, 0.2, (n, 2))).float()
    for k in e.episode_sums:
        e.episode_sums[k][:] = torch.from_numpy(rs.normal(0, 1, n)).float()

    out = {}
    names = list(e.reward_names)
    # the termination term is not a reward *function* (point_foot.py:808-809) but has an episode sum (:818-820): kept last
    sum_names = names + (["termination"] if "termination" in e.episode_sums else [])
    out["reward_names"] = np.array(names)
    if len(sum_names) != len(names):
        out["sum_names"] = np.array(sum_names)
        out["termination_scale"] = np.array(e.reward_scales["termination"], np.float64)
        out["only_positive_rewards"] = np.array(bool(cfg.rewards.only_positive_rewards))
    out["reward_scales"] = np.array([e.reward_scales[k] for k in names], dtype=np.float64)
    out["obs_noise_vec"] = e.noise_scale_vec[0].numpy()
    pv = e.noise_scale_vec[1]
    out["priv_noise_vec"] = pv.numpy() if pv is not None else np.zeros(0, np.float32)
    out["p_gains"] = e.p_gains.numpy()
    out["d_gains"] = e.d_gains.numpy()
    out["torque_limits"] = e.torque_limits.numpy()
    out["dof_pos_limits"] = e.dof_pos_limits.numpy()
    out["base_mass"] = e.base_mass.numpy()
    out["feet_indices"] = e.feet_indices.numpy()
    out["penalised_contact_indices"] = e.penalised_contact_indices.numpy()
    out["termination_contact_indices"] = e.termination_contact_indices.numpy()
    out["max_episode_length"] = np.array(e.max_episode_length)
    out["push_interval"] = np.array(cfg.domain_rand.push_interval)
    out["dt"] = np.array(e.dt)
    if rough:
        out["terrain_types"] = e.terrain_types.numpy()
        out["terrain_origins"] = e.terrain_origins.numpy()

    def snap(prefix):
        d = {}
        d["root"] = e.root_states.numpy().copy()
        d["dof"] = e.dof_state.numpy().reshape(n, 6, 2).copy()
        d["episode_length"] = e.episode_length_buf.numpy().copy()
        d["commands"] = e.commands.numpy().copy()
        d["last_actions"] = e.last_actions.numpy().copy()
        d["last_dof_vel"] = e.last_dof_vel.numpy().copy()
        d["feet_air_time"] = e.feet_air_time.numpy().copy()
        d["last_feet_air_time"] = e.last_feet_air_time.numpy().copy()
        d["contact_filt"] = e.contact_filt.numpy().copy()
        d["first_contact"] = e.first_contact.numpy().copy()
        d["last_contacts"] = e.last_contacts.numpy().copy()
        d["feet_height"] = e.feet_height.numpy().copy()
        d["current_max_feet_height"] = e.current_max_feet_height.numpy().copy()
        d["last_max_feet_height"] = e.last_max_feet_height.numpy().copy()
        d["episode_sums"] = np.stack([e.episode_sums[k].numpy() for k in sum_names]).copy()
        d["env_origins"] = e.env_origins.numpy().copy()
        if rough:
            d["terrain_levels"] = e.terrain_levels.numpy().copy()
        d["common_step_counter"] = np.array(e.common_step_counter)
        return {f"{prefix}_{k}": v for k, v in d.items()}

    out.update(snap("init"))
    for t in range(steps):
        acts = rs.normal(0, 1.5, (n, 6)).astype(np.float32)
        acts[0, 0] = 250.0                       # exercises clip_actions (+-100)
        subs = [synth.synth_sim_state(rs, n, 9, 6, origins, feet, rough) for _ in range(4)]
        if "allrew" in task:
            for s_ in subs:      # joint speeds past the soft velocity limit (18 / 45 rad/s) on every third env: _reward_dof_vel_limits
                s_["dof"][::3, :, 1] *= 6.0
        e.gym.dof_queue = [s["dof"] for s in subs]
        e.gym.post = subs[3]
        if t == 1:
            # force termination contacts on a handful of envs
            subs[3]["contact"][5:9, 0, :] = 7.0
        rl.take()
        obs, priv, rew, reset, extras = e.step(torch.from_numpy(acts))
        calls = rl.take()
        ids = reset.nonzero(as_tuple=False).flatten().numpy()
        # decode the logged draws in the reference's call order (point_foot.py:249-338, 590-603)
        ci = 0
        r_push = np.zeros((n, 3), np.float32)
        r_level = np.full((n,), -1, np.int64)
        r_dof = np.zeros((n, 6), np.float32)
        r_xy = np.zeros((n, 2), np.float32)
        r_vel = np.zeros((n, 6), np.float32)
        r_obs = np.zeros((n, e.num_obs), np.float32)
        r_priv = np.zeros((n, e.num_privileged_obs - e.num_obs), np.float32)
        pushed = e.cfg.domain_rand.push_robots and (e.common_step_counter % e.cfg.domain_rand.push_interval == 0)
        if pushed:
            kind, shape, tt = calls[ci]; ci += 1
            assert kind == "rand" and shape == (n, 3)
            r_push = tt.numpy().copy()
        if len(ids):
            if rough:
                kind, shape, tt = calls[ci]; ci += 1
                assert kind == "randint_like" and shape == (len(ids),)
                r_level[ids] = tt.numpy()
            kind, shape, tt = calls[ci]; ci += 1
            assert kind == "rand" and shape == (len(ids), 6)
            r_dof = scatter_rows(n, ids, tt, 6)
            if rough:
                kind, shape, tt = calls[ci]; ci += 1
                assert kind == "rand" and shape == (len(ids), 2)
                r_xy = scatter_rows(n, ids, tt, 2)
            kind, shape, tt = calls[ci]; ci += 1
            assert kind == "rand" and shape == (len(ids), 6)
            r_vel = scatter_rows(n, ids, tt, 6)
        if add_noise:
            kind, shape, tt = calls[ci]; ci += 1
            assert kind == "rand_like" and shape == (n, e.num_obs)
            r_obs = tt.numpy().copy()
            kind, shape, tt = calls[ci]; ci += 1
            assert kind == "rand_like" and shape == (n, e.num_privileged_obs - e.num_obs)
            r_priv = tt.numpy().copy()
        assert ci == len(calls), (ci, len(calls))
        p = f"s{t}"
        out[f"{p}_actions"] = acts
        out[f"{p}_dof_pre"] = subs[2]["dof"]
        out[f"{p}_dof_post"] = subs[3]["dof"]
        out[f"{p}_sim_root"] = subs[3]["root"]
        out[f"{p}_sim_contact"] = subs[3]["contact"]
        out[f"{p}_sim_feet_pos"] = subs[3]["body"][:, feet, 0:3]
        out[f"{p}_r_push"] = r_push
        out[f"{p}_r_level"] = r_level
        out[f"{p}_r_dof"] = r_dof
        out[f"{p}_r_xy"] = r_xy
        out[f"{p}_r_vel"] = r_vel
        out[f"{p}_r_obs"] = r_obs
        out[f"{p}_r_priv"] = r_priv
        out[f"{p}_pushed"] = np.array(bool(pushed))
        out[f"{p}_obs"] = obs.numpy().copy()
        out[f"{p}_priv"] = priv.numpy().copy()
        out[f"{p}_rew"] = rew.numpy().copy()
        out[f"{p}_reset"] = reset.numpy().copy()
        out[f"{p}_time_out"] = e.time_out_buf.numpy().copy()
        out[f"{p}_terminate"] = e.terminate.numpy().copy()
        out[f"{p}_torques"] = e.torques.numpy().copy()
        out[f"{p}_base_lin_vel"] = e.base_lin_vel.numpy().copy()
        out[f"{p}_base_ang_vel"] = e.base_ang_vel.numpy().copy()
        out[f"{p}_projected_gravity"] = e.projected_gravity.numpy().copy()
        if rough:
            out[f"{p}_measured_heights"] = e.measured_heights.numpy().copy()
        out[f"{p}_ext_force"] = e.rigid_body_external_forces.numpy()[:, 0, :].copy()
        if "allrew" in task:      # every term's value inside this step (the tests assert that every term is exercised)
            out[f"{p}_terms"] = np.stack([term_rec[i] for i in range(len(names))])
        if len(ids):
            ep = extras["episode"]
            out[f"{p}_extras_rew"] = np.array([float(ep["rew_" + k]) for k in sum_names], np.float64)
            if "terrain_level" in ep:
                out[f"{p}_extras_terrain_level"] = np.array(float(ep["terrain_level"]))
        out.update(snap(p))
    torch.rand = _orig_rand
    return out


# ----------------------------------------------------------------------------- LeggedRobot (anymal_c_rough) golden
def build_env_lr(n, rng_seed, model, variant=None):
    """Construct the reference LeggedRobot (legged_gym/envs/base/legged_robot.py) with the anymal_c_rough config, bypassing
    the Isaac-Gym-bound constructors.  `model` = body/dof names of the collapsed ANYmal-C tree (pointfoot_amd URDF parser)."""
    from legged_gym.envs.base.legged_robot import LeggedRobot
    from legged_gym.envs.anymal_c.mixed_terrains.anymal_c_rough_config import AnymalCRoughCfg
    cfg = AnymalCRoughCfg()
    cfg.env.num_envs = n
    variants.apply(cfg, variant)
    e = object.__new__(LeggedRobot)
    e.cfg = cfg
    e.sim_params = types.SimpleNamespace(dt=cfg.sim.dt, use_gpu_pipeline=False)
    e.height_samples = None
    e.debug_viz = False
    e.init_done = False
    e._parse_cfg(cfg)
    e.gym = FakeGym(); e.gym.env = e
    e.sim = None; e.physics_engine = None; e.sim_device = "cpu"; e.headless = True; e.device = "cpu"
    e.num_envs, e.num_obs, e.num_privileged_obs, e.num_actions = n, cfg.env.num_observations, cfg.env.num_privileged_obs, cfg.env.num_actions
    e.obs_buf = torch.zeros(n, e.num_obs)
    e.rew_buf = torch.zeros(n)
    e.reset_buf = torch.ones(n, dtype=torch.long)
    e.episode_length_buf = torch.zeros(n, dtype=torch.long)
    e.time_out_buf = torch.zeros(n, dtype=torch.bool)
    e.privileged_obs_buf = None
    e.extras = {}
    e.viewer = None; e.enable_viewer_sync = True; e.up_axis_idx = 2
    bn, dn = model["body_names"], model["dof_names"]
    e.num_dof = e.num_dofs = len(dn); e.num_bodies = len(bn); e.dof_names = dn
    feet = [i for i, s in enumerate(bn) if cfg.asset.foot_name in s]
    pen, term = [], []
    for name in cfg.asset.penalize_contacts_on:
        pen.extend([i for i, s in enumerate(bn) if name in s])
    for name in cfg.asset.terminate_after_contacts_on:
        term.extend([i for i, s in enumerate(bn) if name in s])
    e.feet_indices = torch.tensor(feet, dtype=torch.long)
    e.penalised_contact_indices = torch.tensor(pen, dtype=torch.long)
    e.termination_contact_indices = torch.tensor(term, dtype=torch.long)
    bis = cfg.init_state.pos + cfg.init_state.rot + cfg.init_state.lin_vel + cfg.init_state.ang_vel
    e.base_init_state = torch.tensor(bis, dtype=torch.float)
    nd = len(dn)
    e.dof_pos_limits = torch.zeros(nd, 2); e.dof_vel_limits = torch.zeros(nd); e.torque_limits = torch.zeros(nd)
    for i in range(nd):      # _process_dof_props for env 0 (legged_robot.py:286-305)
        e.dof_pos_limits[i, 0] = model["q_lo"][i]; e.dof_pos_limits[i, 1] = model["q_hi"][i]
        if variant and "allrew" in variant:      # ANYmal's joints carry no position limits: synthetic ones, so that
            e.dof_pos_limits[i, 0], e.dof_pos_limits[i, 1] = -0.9 + 0.05 * i, 0.7 + 0.04 * i      # _reward_dof_pos_limits is exercised
        e.dof_vel_limits[i] = model["vel_limit"][i]; e.torque_limits[i] = model["effort_limit"][i]
        m = (e.dof_pos_limits[i, 0] + e.dof_pos_limits[i, 1]) / 2
        r = e.dof_pos_limits[i, 1] - e.dof_pos_limits[i, 0]
        e.dof_pos_limits[i, 0] = m - 0.5 * r * cfg.rewards.soft_dof_pos_limit
        e.dof_pos_limits[i, 1] = m + 0.5 * r * cfg.rewards.soft_dof_pos_limit
    nrs = np.random.RandomState(rng_seed + 100)
    hf, to = synth.synth_heightfield(), synth.synth_terrain_origins()
    e.terrain = types.SimpleNamespace(cfg=cfg.terrain, env_length=cfg.terrain.terrain_length, env_width=cfg.terrain.terrain_width, env_origins=to)
    e.height_samples = torch.tensor(hf).view(synth.TOT_ROWS, synth.TOT_COLS)
    e.custom_origins = True
    e.terrain_levels = torch.from_numpy(nrs.randint(0, cfg.terrain.max_init_terrain_level + 1, n)).long()
    e.terrain_levels[:3] = cfg.terrain.num_rows - 1
    e.terrain_types = torch.div(torch.arange(n), (n / cfg.terrain.num_cols), rounding_mode="floor").to(torch.long)
    e.max_terrain_level = cfg.terrain.num_rows
    e.terrain_origins = torch.from_numpy(to).to(torch.float)
    e.env_origins = torch.zeros(n, 3)
    e.env_origins[:] = e.terrain_origins[e.terrain_levels, e.terrain_types]
    e._t_root = torch.zeros(n, 13); e._t_root[:, 6] = 1.0
    e._t_dof = torch.zeros(n * nd, 2)
    e._t_contact = torch.zeros(n * len(bn), 3)
    g = e.gym
    g.acquire_actor_root_state_tensor = lambda sim: e._t_root
    g.acquire_dof_state_tensor = lambda sim: e._t_dof
    g.acquire_net_contact_force_tensor = lambda sim: e._t_contact
    e._init_buffers()
    e._prepare_reward_function()
    e.init_done = True
    return e, cfg, feet


def run_sequence_lr(n, steps, seed, model, variant=None):
    e, cfg, feet = build_env_lr(n, seed, model, variant)
    allrew = bool(variant) and "allrew" in variant
    term_rec = record_terms(e) if allrew else None
    nb, nd = len(model["body_names"]), len(model["dof_names"])
    rl = RandLog(seed)
    torch.rand = rl.rand; torch.rand_like = rl.rand_like; torch.randint_like = rl.randint_like
    rs = np.random.RandomState(seed)
    origins = e.env_origins.numpy().astype(np.float64)
    s0 = synth.synth_sim_state(rs, n, nb, nd, origins, feet, True)
    e.root_states.copy_(torch.from_numpy(s0["root"]))
    e.dof_state.copy_(torch.from_numpy(s0["dof"].reshape(-1, 2)))
    e.contact_forces.copy_(torch.from_numpy(s0["contact"]))
    e.episode_length_buf = torch.from_numpy(rs.randint(0, 1000, n)).long()
    e.episode_length_buf[n // 2: n // 2 + 3] = 1000           # time-outs on the first step
    e.episode_length_buf[4:8] = 499                            # periodic command resampling at 500
    e.common_step_counter = int(cfg.domain_rand.push_interval) - 2
    e.commands[:, 0] = torch.from_numpy(rs.uniform(-1, 1, n)).float()
    e.commands[:, 1] = torch.from_numpy(rs.uniform(-1, 1, n)).float()
    e.commands[:, 3] = torch.from_numpy(rs.uniform(-3.14, 3.14, n)).float()
    if allrew:
        e.commands[::5, :2] = 0.0          # standing commands: _reward_stand_still (legged_robot.py:900-902)
    e.last_actions[:] = torch.from_numpy(rs.normal(0, 1, (n, nd))).float()
    e.last_dof_vel[:] = torch.from_numpy(rs.normal(0, 3, (n, nd))).float()
    e.feet_air_time[:] = torch.from_numpy(rs.uniform(0, 0.9, (n, 4))).float() * (torch.from_numpy(rs.uniform(size=(n, 4))) > 0.3)
    e.last_contacts[:] = torch.from_numpy(rs.uniform(size=(n, 4)) > 0.5)
    for k in e.episode_sums:
        e.episode_sums[k][:] = torch.from_numpy(rs.normal(0, 1, n)).float()
    names = list(e.reward_names)
    sum_names = names + (["termination"] if "termination" in e.episode_sums else [])
    out = dict(reward_names=np.array(names), reward_scales=np.array([e.reward_scales[k] for k in names], np.float64),
               noise_scale_vec=e.noise_scale_vec.numpy(), p_gains=e.p_gains.numpy(), d_gains=e.d_gains.numpy(),
               default_dof_pos=e.default_dof_pos.numpy().reshape(-1), terrain_types=e.terrain_types.numpy(),
               terrain_origins=e.terrain_origins.numpy(), feet_indices=np.array(feet),
               penalised_contact_indices=e.penalised_contact_indices.numpy(), termination_contact_indices=e.termination_contact_indices.numpy(),
               push_interval=np.array(cfg.domain_rand.push_interval), max_episode_length=np.array(e.max_episode_length))
    if allrew:
        out.update(sum_names=np.array(sum_names), termination_scale=np.array(e.reward_scales["termination"], np.float64),
                   dof_pos_limits=e.dof_pos_limits.numpy().copy(), dof_vel_limits=e.dof_vel_limits.numpy().copy(),
                   torque_limits=e.torque_limits.numpy().copy())

    def snap(prefix):
        d = dict(root=e.root_states.numpy().copy(), dof=e.dof_state.numpy().reshape(n, nd, 2).copy(),
                 episode_length=e.episode_length_buf.numpy().copy(), commands=e.commands.numpy().copy(),
                 last_actions=e.last_actions.numpy().copy(), last_dof_vel=e.last_dof_vel.numpy().copy(),
                 feet_air_time=e.feet_air_time.numpy().copy(), last_contacts=e.last_contacts.numpy().copy(),
                 episode_sums=np.stack([e.episode_sums[k].numpy() for k in sum_names]).copy(), env_origins=e.env_origins.numpy().copy(),
                 terrain_levels=e.terrain_levels.numpy().copy(), common_step_counter=np.array(e.common_step_counter))
        return {f"{prefix}_{k}": v for k, v in d.items()}
    out.update(snap("init"))
    for t in range(steps):
        acts = rs.normal(0, 1.0, (n, nd)).astype(np.float32)
        subs = [synth.synth_sim_state(rs, n, nb, nd, origins, feet, True) for _ in range(4)]
        if allrew:
            for s_ in subs:
                s_["dof"][::3, :, 1] *= 6.0
        if t == 1:
            subs[3]["contact"][9:12, 0, :] = 7.0
        e.gym.dof_queue = [s["dof"] for s in subs]
        e.gym.post = dict(root=subs[3]["root"], contact=subs[3]["contact"])
        ep_before = e.episode_length_buf.numpy().copy()
        rl.take()
        obs, priv, rew, reset, extras = e.step(torch.from_numpy(acts))
        calls = rl.take()
        ids = reset.nonzero(as_tuple=False).flatten().numpy()
        rsm = np.nonzero((ep_before + 1) % int(cfg.commands.resampling_time / e.dt) == 0)[0]
        ci = 0
        r_cmd = np.zeros((n, 3), np.float32); r_cmd_reset = np.zeros((n, 3), np.float32)
        for k in range(3):      # periodic resample: 3 draws of shape (len(rsm), 1), even when empty
            kind, shape, tt = calls[ci]; ci += 1
            assert kind == "rand" and shape == (len(rsm), 1), (kind, shape, len(rsm))
            if len(rsm): r_cmd[rsm, k] = tt.numpy().reshape(-1)
        pushed = cfg.domain_rand.push_robots and (e.common_step_counter % cfg.domain_rand.push_interval == 0)
        r_push = np.zeros((n, 3), np.float32)
        if pushed:
            kind, shape, tt = calls[ci]; ci += 1
            assert kind == "rand" and shape == (n, 2)
            r_push[:, :2] = tt.numpy()
        r_level = np.full((n,), -1, np.int64); r_dof = np.zeros((n, nd), np.float32)
        r_xy = np.zeros((n, 2), np.float32); r_vel = np.zeros((n, 6), np.float32)
        if len(ids):
            kind, shape, tt = calls[ci]; ci += 1
            assert kind == "randint_like" and shape == (len(ids),)
            r_level[ids] = tt.numpy()
            kind, shape, tt = calls[ci]; ci += 1
            assert kind == "rand" and shape == (len(ids), nd)
            r_dof = scatter_rows(n, ids, tt, nd)
            kind, shape, tt = calls[ci]; ci += 1
            assert kind == "rand" and shape == (len(ids), 2)
            r_xy = scatter_rows(n, ids, tt, 2)
            kind, shape, tt = calls[ci]; ci += 1
            assert kind == "rand" and shape == (len(ids), 6)
            r_vel = scatter_rows(n, ids, tt, 6)
            for k in range(3):
                kind, shape, tt = calls[ci]; ci += 1
                assert kind == "rand" and shape == (len(ids), 1)
                r_cmd_reset[ids, k] = tt.numpy().reshape(-1)
        kind, shape, tt = calls[ci]; ci += 1
        assert kind == "rand_like" and shape == (n, e.num_obs)
        r_obs = tt.numpy().copy()
        assert ci == len(calls), (ci, len(calls))
        p = f"s{t}"
        out.update({f"{p}_actions": acts, f"{p}_dof_pre": subs[2]["dof"], f"{p}_dof_post": subs[3]["dof"], f"{p}_sim_root": subs[3]["root"],
                    f"{p}_sim_contact": subs[3]["contact"], f"{p}_r_cmd": r_cmd, f"{p}_r_cmd_reset": r_cmd_reset, f"{p}_r_push": r_push,
                    f"{p}_r_level": r_level, f"{p}_r_dof": r_dof, f"{p}_r_xy": r_xy, f"{p}_r_vel": r_vel, f"{p}_r_obs": r_obs,
                    f"{p}_pushed": np.array(bool(pushed)), f"{p}_obs": obs.numpy().copy(), f"{p}_rew": rew.numpy().copy(),
                    f"{p}_reset": reset.numpy().copy(), f"{p}_time_out": e.time_out_buf.numpy().copy(), f"{p}_torques": e.torques.numpy().copy(),
                    f"{p}_base_lin_vel": e.base_lin_vel.numpy().copy(), f"{p}_measured_heights": e.measured_heights.numpy().copy()})
        if len(ids):
            out[f"{p}_extras_rew"] = np.array([float(extras["episode"]["rew_" + k]) for k in sum_names], np.float64)
        if allrew:
            out[f"{p}_terms"] = np.stack([term_rec[i] for i in range(len(names))])
        out.update(snap(p))
    torch.rand = _orig_rand
    return out


def feet_stumble_error():
    """A non-zero `feet_stumble` scale: the reference looks up `_reward_feet_stumble`, which does not exist (the method is
    `_reward_stumble`, point_foot.py:1226) -> AttributeError out of _prepare_reward_function (:814-815)."""
    import types as _t
    from legged_gym.envs.pointfoot.point_foot import PointFoot
    e = object.__new__(PointFoot)
    e.dt, e.num_envs, e.device = 0.02, 4, "cpu"
    e.reward_scales = {"feet_stumble": -1.0, "torques": -1e-5}
    try:
        e._prepare_reward_function()
    except AttributeError as ex:
        return {"error_type": np.array("AttributeError"), "message": np.array(str(ex))}
    return {"error_type": np.array("none"), "message": np.array("")}


def anymal_model():
    """ANYmal-C body / DOF tables from the reference URDF via this repo's parser (names and limits only are used here)."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from pointfoot_amd.utils.urdf import parse_urdf
    return parse_urdf(os.path.join(REF, "resources", "robots", "anymal_c", "urdf", "anymal_c.urdf"))


def misc_vectors():
    from legged_gym.utils.math import wrap_to_pi, quat_apply_yaw
    rs = np.random.RandomState(5)
    ang = torch.from_numpy(rs.uniform(-12, 12, 64)).float()
    out = {"wrap_in": ang.numpy().copy(), "wrap_out": wrap_to_pi(ang.clone()).numpy()}
    q = torch.from_numpy(synth.rand_unit_quat(rs, 64)).float()
    v = torch.from_numpy(rs.normal(0, 1, (64, 3))).float()
    out["qay_q"] = q.numpy()
    out["qay_v"] = v.numpy()
    out["qay_out"] = quat_apply_yaw(q, v).numpy()
    return out


def terrain_grid_vectors(seed=1):
    """Pins the reference's OWN curriculum-grid logic (legged_gym/utils/terrain.py:38-164: curiculum(), make_terrain(),
    add_terrain_to_map(), origin rule, borders) for the pointfoot_rough terrain config.  The sub-terrain generators
    belong to the closed isaacgym.terrain_utils package; THIS REPO's restatements (pointfoot_amd/utils/terrain.py) are
    plugged into the stub module, wrapped so that every call the reference makes is logged (tile, generator, arguments).
    The fixture therefore fixes which generator the reference calls for which tile with which parameters, in which
    order (= numpy RNG consumption order), where it pastes the tile and how it derives env_origins -- not the
    generators' own arithmetic, which stays parity-unpinned."""
    import hashlib
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from pointfoot_amd.utils import terrain as mine
    from pointfoot_amd.envs.pointfoot.mixed_terrain.pointfoot_rough_config import PointFootRoughCfg
    tu = sys.modules["isaacgym.terrain_utils"]
    calls = []

    def wrap(name):
        fn = getattr(mine, name)

        def logged(terrain, *a, **kw):
            calls.append((name, tuple(float(v) for v in a), tuple(sorted((k, float(v)) for k, v in kw.items()))))
            return fn(terrain, *a, **kw)
        return logged
    tu.SubTerrain = mine.SubTerrain
    for name in ("pyramid_sloped_terrain", "random_uniform_terrain", "pyramid_stairs_terrain", "discrete_obstacles_terrain",
                 "stepping_stones_terrain"):
        setattr(tu, name, wrap(name))
    tu.convert_heightfield_to_trimesh = lambda hf, hs, vs, thr: (np.zeros((0, 3), np.float32), np.zeros((0, 3), np.uint32))
    from legged_gym.envs.pointfoot.point_foot import PointFoot  # noqa: F401  (resolves the reference's envs <-> utils import cycle)
    from legged_gym.utils.terrain import Terrain as RefTerrain
    cfg = PointFootRoughCfg.terrain()          # same values as the reference's pointfoot_rough_config.py:14-44
    np.random.seed(seed)
    t = RefTerrain(cfg, 4096)
    hf = np.ascontiguousarray(t.height_field_raw)
    names = sorted({c[0] for c in calls})
    out = {"seed": np.int64(seed), "shape": np.array(hf.shape), "sha256": np.frombuffer(hashlib.sha256(hf.tobytes()).digest(), np.uint8),
           "env_origins": t.env_origins.copy(), "coarse": hf[::13, ::17].copy(),
           "tile_max": np.array([[hf[t.border + i * t.length_per_env_pixels:t.border + (i + 1) * t.length_per_env_pixels,
                                     t.border + j * t.width_per_env_pixels:t.border + (j + 1) * t.width_per_env_pixels].max()
                                  for j in range(cfg.num_cols)] for i in range(cfg.num_rows)]),
           "call_names": np.array(names), "call_gen": np.array([names.index(c[0]) for c in calls]),
           "call_args": np.array([list(c[1]) + [v for _, v in c[2]] + [np.nan] * (8 - len(c[1]) - len(c[2])) for c in calls]),
           "border_px": np.int64(t.border), "tot_rows": np.int64(t.tot_rows), "tot_cols": np.int64(t.tot_cols)}
    return out


def main():
    _install_stubs()
    sys.path.insert(0, REF)
    import legged_gym  # noqa: F401
    if len(sys.argv) > 1 and sys.argv[1] == "terrain":        # only the terrain-grid fixture
        np.savez_compressed(os.path.join(HERE, "terrain_grid.npz"), **terrain_grid_vectors())
        print("wrote terrain_grid.npz")
        return
    jobs = [("flat", 64, 6, 1, "P", True), ("rough", 64, 6, 2, "P", True),
            ("flat", 16, 2, 3, "V", False), ("flat", 16, 2, 4, "T", False), ("rough_hactor", 16, 3, 5, "P", True)]
    # round 4: every reward term live (24 PointFoot methods / 19 LeggedRobot methods), and only_positive_rewards on PointFoot
    allrew_jobs = [("rough_allrew", 64, 4, 7, "P", True), ("rough_allrew_onlypos", 32, 3, 8, "P", True), ("flat_allrew", 32, 3, 9, "P", False)]
    if len(sys.argv) > 1 and sys.argv[1] == "hactor":        # only the heights-in-actor-observations fixture
        jobs = jobs[-1:]
    elif len(sys.argv) > 1 and sys.argv[1] == "allrew":      # only the round-4 fixtures
        jobs = allrew_jobs
    else:
        jobs = jobs + allrew_jobs
    for task, n, steps, seed, ct, noise in jobs:
        out = run_sequence(task, n, steps, seed, ct, noise)
        fn = os.path.join(HERE, f"pf_{task}_{ct}_n{n}.npz")
        np.savez_compressed(fn, **out)
        print("wrote", fn, os.path.getsize(fn), "bytes; rewards:", list(out["reward_names"]))
    if len(sys.argv) > 1 and sys.argv[1] == "hactor":
        return
    out = run_sequence_lr(32, 4, 10, anymal_model(), variant="allrew")
    fn = os.path.join(HERE, "lr_anymal_c_rough_allrew_n32.npz")
    np.savez_compressed(fn, **out)
    print("wrote", fn, os.path.getsize(fn), "bytes; rewards:", list(out["reward_names"]))
    np.savez_compressed(os.path.join(HERE, "feet_stumble_error.npz"), **feet_stumble_error())
    if len(sys.argv) > 1 and sys.argv[1] == "allrew":
        return
    np.savez_compressed(os.path.join(HERE, "misc.npz"), **misc_vectors())
    out = run_sequence_lr(32, 5, 6, anymal_model())
    fn = os.path.join(HERE, "lr_anymal_c_rough_n32.npz")
    np.savez_compressed(fn, **out)
    print("wrote", fn, os.path.getsize(fn), "bytes; rewards:", list(out["reward_names"]))
    np.savez_compressed(os.path.join(HERE, "terrain_grid.npz"), **terrain_grid_vectors())
    print("wrote terrain_grid.npz")


if __name__ == "__main__":
    main()
