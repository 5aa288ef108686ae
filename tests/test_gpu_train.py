"""End-to-end GPU tests through the reference-compatible Python surface: task_registry.make_env / PointFoot.step /
OnPolicyRunner.learn / save / load / get_inference_policy / export (BASELINE configs[0] analogue on the GPU)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

import common

pytestmark = pytest.mark.gpu


def _args(task, n, extra=()):
    from pointfoot_amd.utils import get_args
    return get_args(["--task", task, "--headless", "--num_envs", str(n), "--sim_device", "cuda:0", "--rl_device", "cuda:0", *extra])


@pytest.mark.parametrize("task", ["pointfoot_flat", "pointfoot_rough"])
def test_env_class_matches_oracle_and_protocol(task):
    from pointfoot_amd.envs import task_registry
    n = 64
    env, cfg = task_registry.make_env(task, args=_args(task, n))
    rough = task.endswith("rough")
    assert env.num_envs == n and env.num_obs == 27 and env.num_privileged_obs == (148 if rough else 27) and env.num_actions == 6
    assert env.max_episode_length == 1000.0 and abs(env.dt - 0.02) < 1e-9
    obs, priv = env.reset()
    assert obs.shape == (n, 27) and priv.shape == (n, env.num_privileged_obs) and obs.is_cuda
    assert env.get_observations() is obs and env.get_privileged_observations() is priv
    # the runner ASSIGNS episode_length_buf (random episode start): the kernel must see the new values
    env.episode_length_buf = torch.randint_like(env.episode_length_buf, high=1000)
    env.episode_length_buf[:3] = 1000
    # mirror the env's state into a HostSim with the very same structs and step both
    sim = common.HostSim(cfg, n, hf=None if not rough else env.height_samples.cpu().numpy(),
                         terrain_origins=None if not rough else env.terrain_origins.cpu().numpy())
    sim.c = env._cfg_c
    assert bytes(sim.L) == bytes(env._L)
    rs = np.random.RandomState(0)
    cset = env.bind_contact_set()
    mismatched = 0
    for t in range(5):
        torch.cuda.synchronize()
        sim.state[:] = env._state.cpu().numpy()
        sim.episode_length[:] = env.episode_length_buf.cpu().numpy()
        sim.terrain_level[:] = env.terrain_levels.cpu().numpy(); sim.terrain_type[:] = env.terrain_types.cpu().numpy()
        act = torch.from_numpy(rs.normal(0, 0.5, (n, 6)).astype(np.float32))
        o, p, r, d, extras = env.step(act)            # CPU tensor in: moved to the device like the reference does (:177)
        sim.step("oracle", act.numpy(), env.common_step_counter)
        torch.cuda.synchronize()
        # strict 1e-4 wherever the kernel took the oracle's discrete decisions (reset, contact / joint-limit sets, cells)
        same = (d.cpu().numpy() == sim.reset.astype(bool)) & (cset.cpu().numpy().view(np.uint64) == sim.contact_set).all(axis=(1, 2))
        mismatched += int((~same).sum())
        assert same.mean() >= 62 / 64                   # measured: 0 mismatched env-steps of 320 on both tasks
        assert np.abs(o.cpu().numpy()[same] - sim.obs[same]).max() <= 1e-4
        assert np.abs(r.cpu().numpy()[same] - sim.rew[same]).max() <= 1e-4
        assert d.dtype == torch.bool and extras["time_outs"].dtype == torch.bool
        if t == 0:
            assert bool(extras["time_outs"][:3].all())     # episode length 1001 > 1000
        ep = extras["episode"]
        assert set(k for k in ep if k.startswith("rew_")) == {"rew_" + k for k in env.reward_names}
        if rough and t == 2:
            # `measured_heights` is materialised on first access (the kernel keeps the scan in LDS otherwise) ...
            assert env._measured_heights is None
            o_before, p_before, r_before = o.clone(), p.clone(), r.clone()
            mh = env.measured_heights
            torch.cuda.synchronize()
            # ... by the height scan alone: the attribute read leaves observations / rewards as the step wrote them
            assert torch.equal(o, o_before) and torch.equal(p, p_before) and torch.equal(r, r_before)
            keep = ~sim.reset.astype(bool) & same
            np.testing.assert_allclose(mh.cpu().numpy()[keep], sim.heights[keep], atol=1e-6)
        if rough and t == 3:      # ... and is then written by every step
            keep = ~sim.reset.astype(bool) & same
            assert int(keep.sum()) > 32
            np.testing.assert_allclose(env.measured_heights.cpu().numpy()[keep], sim.heights[keep], atol=1e-6)
        assert ("terrain_level" in ep) == rough
    common.record_parity(f"test_env_class_matches_oracle_and_protocol[{task}]", env_steps=5 * n, flipped=mismatched, flipped_frac=mismatched / (5 * n))
    # live views: scripts write commands / read dof state like on the reference's wrapped tensors
    env.commands[0, 0] = 0.7
    assert abs(env._state[0, env._L.commands].item() - 0.7) < 1e-7
    assert env.dof_pos.shape == (n, 6) and env.contact_forces.shape == (n, 9, 3) and env.base_lin_vel.shape == (n, 3)
    env.update_cmd([0.1, 0.2, 0.3])
    torch.cuda.synchronize()
    assert abs(env.get_observations()[0, 24].item() - 0.1 * 2.0) < (0.5 if cfg.noise.add_noise else 1e-6)


def test_train_save_load_play_export(tmp_path):
    from pointfoot_amd.envs import task_registry
    from pointfoot_amd.utils import export_policy_as_jit
    from pointfoot_amd.utils.helpers import get_load_path
    args = _args("pointfoot_flat", 64, ["--max_iterations", "3"])
    env, _ = task_registry.make_env("pointfoot_flat", args=args)
    runner, train_cfg = task_registry.make_alg_runner(env, "pointfoot_flat", args=args, log_root=str(tmp_path))
    p0 = runner.alg.params.clone()
    runner.learn(num_learning_iterations=3, init_at_random_ep_len=True)
    torch.cuda.synchronize()
    assert torch.isfinite(runner.alg.params).all() and not torch.equal(p0, runner.alg.params)
    assert runner.current_learning_iteration == 3
    path = get_load_path(str(tmp_path))
    assert path.endswith("model_3.pt")
    ck = torch.load(path, weights_only=True)
    assert set(ck) == {"model_state_dict", "optimizer_state_dict", "iter", "infos"}
    assert list(ck["model_state_dict"])[:3] == ["std", "actor.0.weight", "actor.0.bias"]
    # resume into a fresh runner: parameters and Adam moments come back bit-identical
    env2, _ = task_registry.make_env("pointfoot_flat", args=args)
    r2, _ = task_registry.make_alg_runner(env2, "pointfoot_flat", args=args, log_root=None)
    r2.load(path)
    assert torch.equal(r2.alg.params, runner.alg.params) and torch.equal(r2.alg.adam_m, runner.alg.adam_m)
    assert r2.alg.scalars[5].item() == runner.alg.scalars[5].item() == 60.0       # 3 iterations x 20 Adam steps
    policy = r2.get_inference_policy(device="cuda:0")
    obs = env2.get_observations()
    a = policy(obs.detach())
    out = torch.zeros_like(a)
    common._abi.check(r2.alg._lib.ppo_act_inference(r2.alg._h, obs.contiguous().data_ptr(), 64, out.data_ptr(), r2.alg._stream()))
    torch.testing.assert_close(a, out, atol=2e-5, rtol=1e-5)                     # nn.Module view == HIP actor
    export_policy_as_jit(r2.alg.actor_critic, str(tmp_path / "exported"))
    jit = torch.jit.load(str(tmp_path / "exported" / "policy_1.pt"))
    torch.testing.assert_close(jit(obs.cpu()), a.cpu(), atol=2e-5, rtol=1e-5)


def test_reference_test_env_and_play(tmp_path, monkeypatch):
    """the reference's own smoke test (tests/test_env.py) and play.py flow on the GPU"""
    import pointfoot_amd
    from pointfoot_amd.tests.test_env import test_env
    from pointfoot_amd.scripts import play as play_mod
    from pointfoot_amd.envs import task_registry
    env = test_env(_args("pointfoot_flat", 10), steps=300)
    assert env.num_envs == 10
    # train 2 iterations into <root>/logs/pointfoot_flat/..., then play from the checkpoint
    monkeypatch.setattr(pointfoot_amd, "LEGGED_GYM_ROOT_DIR", str(tmp_path))
    import sys
    tr = sys.modules["pointfoot_amd.utils.task_registry"]
    monkeypatch.setattr(tr, "LEGGED_GYM_ROOT_DIR", str(tmp_path))
    monkeypatch.setattr(play_mod, "LEGGED_GYM_ROOT_DIR", str(tmp_path))
    args = _args("pointfoot_flat", 64, ["--max_iterations", "2"])
    env_cfg, train_cfg = task_registry.get_cfgs("pointfoot_flat")
    saved = (env_cfg.env.num_envs, train_cfg.runner.resume, env_cfg.noise.add_noise, env_cfg.domain_rand.push_robots,
             env_cfg.domain_rand.randomize_friction)
    try:
        e, _ = task_registry.make_env("pointfoot_flat", args=args)
        r, _ = task_registry.make_alg_runner(e, "pointfoot_flat", args=args)
        r.learn(2, init_at_random_ep_len=True)
        from pointfoot_amd.utils import get_args
        played = play_mod.play(get_args(["--task", "pointfoot_flat", "--headless"]), num_steps=50)   # no --num_envs: play caps at 50
        assert played.num_envs == 50 and not played.cfg.noise.add_noise
        assert (tmp_path / "logs" / "pointfoot_flat" / "exported" / "policies" / "policy_1.pt").exists()
    finally:
        env_cfg.env.num_envs, train_cfg.runner.resume, env_cfg.noise.add_noise, env_cfg.domain_rand.push_robots, \
            env_cfg.domain_rand.randomize_friction = saved


def test_anymal_c_flat_trains_through_registry():
    """config-5 family on the GPU: Anymal (LeggedRobot + actuator net), 12 actions, 48-dim obs, critic = actor obs."""
    from pointfoot_amd.envs import task_registry, Anymal
    args = _args("anymal_c_flat", 128, ["--max_iterations", "3"])
    env, cfg = task_registry.make_env("anymal_c_flat", args=args)
    assert isinstance(env, Anymal) and env.num_obs == 48 and env.num_privileged_obs is None and env.num_actions == 12
    assert env.privileged_obs_buf is None and env.obs_buf.shape == (128, 48)
    assert env.sea_hidden_state_per_env.shape == (128, 12, 2, 8)
    runner, _ = task_registry.make_alg_runner(env, "anymal_c_flat", args=args, log_root=None)
    p0 = runner.alg.params.clone()
    runner.learn(3, init_at_random_ep_len=True)
    torch.cuda.synchronize()
    assert torch.isfinite(runner.alg.params).all() and not torch.equal(p0, runner.alg.params)
    obs, priv, rew, done, extras = env.step(torch.zeros(128, 12, device="cuda:0"))
    assert priv is None and torch.isfinite(obs).all() and (rew >= 0).all()       # only_positive_rewards
    assert "rew_feet_air_time" in extras["episode"] and "terrain_level" not in extras["episode"]


@pytest.mark.parametrize("task,obs", [("anymal_c_rough", 235), ("anymal_b", 235), ("a1", 235)])
def test_quadruped_tasks_step_and_stand(task, obs):
    """every registered LeggedRobot task builds through the registry and holds its default pose under zero actions"""
    from pointfoot_amd.envs import task_registry
    env, cfg = task_registry.make_env(task, args=_args(task, 200))
    assert env.num_obs == obs and env.num_actions == 12 and env.num_bodies == 17
    o, _ = env.reset()
    for _ in range(50):
        o, p, r, d, e = env.step(torch.zeros(200, 12, device="cuda:0"))
    torch.cuda.synchronize()
    assert p is None and torch.isfinite(o).all() and torch.isfinite(env._state).all()
    assert float(d.float().mean()) < 0.2                       # not collapsing every step
    z = (env.root_states[:, 2] - env.env_origins[:, 2])
    assert float(z.median()) > 0.5 * cfg.init_state.pos[2]      # still standing on its legs


def test_partial_reset_joint_friction_and_stats_ring():
    """Host entry points of SURVEY row f4 / a12: reset_idx(subset) (point_foot.py:249-299), update_frictions as DOF
    friction (:103-108), and the per-step statistics ring sized from num_steps_per_env."""
    from pointfoot_amd.envs import task_registry
    n = 64
    env, cfg = task_registry.make_env("pointfoot_flat", args=_args("pointfoot_flat", n))
    env.reset()
    g = torch.Generator(device="cuda:0").manual_seed(3)
    for _ in range(20):
        env.step(0.3 * torch.randn(n, 6, device="cuda:0", generator=g))
    torch.cuda.synchronize()
    before = env._state.clone()
    ep = env.episode_length_buf.clone()
    assert (ep > 0).any()
    ids = torch.tensor([3, 17, 40], device="cuda:0")
    env.reset_idx(ids)
    torch.cuda.synchronize()
    keep = torch.ones(n, dtype=torch.bool, device="cuda:0"); keep[ids] = False
    assert torch.equal(env._state[keep], before[keep]) and torch.equal(env.episode_length_buf[keep], ep[keep])
    assert (env.episode_length_buf[ids] == 0).all() and env.reset_buf[ids].all()
    torch.testing.assert_close(env.root_states[ids, 2], env.env_origins[ids, 2] + cfg.init_state.pos[2])
    assert (env.dof_vel[ids] == 0).all() and (env.last_actions[ids] == 0).all() and (env.feet_air_time[ids] == 0).all()
    for k in env.episode_sums:
        assert (env.episode_sums[k][ids] == 0).all()
    assert int(env.extras["episode"]["rew_survival"].isnan()) == 0          # three envs were reset: a fresh mean exists
    # joint friction: the same swing torque moves a joint with friction less
    env.reset()
    env.update_frictions([0.0] * 6, 0, 0)
    env.update_frictions([6.0] * 6, 1, 0)
    assert env.joint_friction.shape == (n, 6) and float(env.joint_friction[1].sum()) == 36.0
    env.root_states[:2, 2] += 3.0            # airborne: only the joint torques act on the legs
    env._state[1] = env._state[0]
    env._state[1, env._L.env_origin:env._L.env_origin + 3] = env._state[0, env._L.env_origin:env._L.env_origin + 3]
    a = torch.zeros(n, 6, device="cuda:0"); a[:, 1] = 0.4; a[:, 4] = -0.4
    travel = torch.zeros(2, device="cuda:0")
    q0 = env.dof_pos[:2].clone()
    for _ in range(6):
        env.step(a)
    torch.cuda.synchronize()
    travel = (env.dof_pos[:2] - q0).abs().sum(dim=1)
    assert float(travel[1]) < 0.8 * float(travel[0]), travel
    # statistics ring follows the runner's rollout length
    runner, _ = task_registry.make_alg_runner(env, "pointfoot_flat", args=_args("pointfoot_flat", n), log_root=None)
    assert env._stats_ring.shape[0] >= runner.num_steps_per_env + 2
    env.ensure_stats_ring(100)
    assert env._stats_ring.shape[0] == 102
    # a re-bind after the ring grew must not resurrect the old ring's pointer (the buffer table is handed over whole
    # by update_frictions / measured_heights / bind_contact_set); a full reset then lands in the slot the extras read
    ring_ptrs = (env._stats_ring.data_ptr(), env._stats_ring.data_ptr() + env._stats_ring.numel() * 4)
    env.update_frictions([1.0] * 6, 2, 0)
    _ = env.bind_contact_set()
    assert ring_ptrs[0] <= int(env._bufs.d_reset_stats or 0) < ring_ptrs[1]
    prev = env._reset_stats
    env.reset()
    torch.cuda.synchronize()
    assert ring_ptrs[0] <= env._reset_stats.data_ptr() < ring_ptrs[1] and env._reset_stats.data_ptr() == int(env._bufs.d_reset_stats)
    assert env._reset_stats.data_ptr() != prev.data_ptr()          # the reset and the step after it took slots of their own
    from pointfoot_amd import _abi as abi
    slots = env._stats_ring[:, abi.PF_MAX_REWARDS]
    assert float(slots.max()) == n                                  # reset_idx(all): every env counted, in a slot of this ring
    assert float(env._stats_ring.sum(dim=1).ne(0).sum()) <= 2       # nothing leaked into other slots
    env.step(a)
    torch.cuda.synchronize()


def test_command_curriculum_widens_ranges():
    """update_command_curriculum (point_foot.py:638-650 / legged_robot.py:303-315): on a step where
    common_step_counter % max_episode_length == 0 and the reset envs' mean tracking reward is above 80 % of its maximum,
    lin_vel_x grows by 0.5 up to max_curriculum -- and the LeggedRobot kernel resamples from the new range."""
    from pointfoot_amd.envs import task_registry
    env_cfg, _ = task_registry.get_cfgs("anymal_c_flat")
    saved = (env_cfg.commands.curriculum, list(env_cfg.commands.ranges.lin_vel_x), env_cfg.commands.max_curriculum)
    try:
        env_cfg.commands.curriculum = True
        env_cfg.commands.max_curriculum = 2.0
        env, cfg = task_registry.make_env("anymal_c_flat", args=_args("anymal_c_flat", 64))
        env.reset()
        lo0, hi0 = env.command_ranges["lin_vel_x"]
        env.common_step_counter = int(env.max_episode_length) - 1          # the next step lands on the curriculum check
        # every env times out on that step with a perfect tracking history
        env.episode_length_buf = torch.full((64,), int(env.max_episode_length), device="cuda:0")
        i = env._sum_names.index("tracking_lin_vel")
        env._state[:, env._L.episode_sums + i] = env.reward_scales["tracking_lin_vel"] * env.max_episode_length
        env.step(torch.zeros(64, 12, device="cuda:0"))
        torch.cuda.synchronize()
        lo1, hi1 = env.command_ranges["lin_vel_x"]
        assert (lo1, hi1) == (max(lo0 - 0.5, -2.0), min(hi0 + 0.5, 2.0))
        assert abs(env._cfg_c.cmd_hi[0] - hi1) < 1e-6 and abs(env._cfg_c.cmd_lo[0] - lo1) < 1e-6
        assert float(env.extras["episode"]["max_command_x"]) == hi1
        # a poor tracking history leaves the range alone
        env.common_step_counter = 2 * int(env.max_episode_length) - 1
        env.episode_length_buf = torch.full((64,), int(env.max_episode_length), device="cuda:0")
        env._state[:, env._L.episode_sums + i] = 0.0
        env.step(torch.zeros(64, 12, device="cuda:0"))
        torch.cuda.synchronize()
        assert tuple(env.command_ranges["lin_vel_x"]) == (lo1, hi1)
    finally:
        env_cfg.commands.curriculum, env_cfg.commands.ranges.lin_vel_x, env_cfg.commands.max_curriculum = saved[0], saved[1], saved[2]


def test_measure_heights_actor_env_and_training_iteration():
    """cfg.terrain.measure_heights_actor (point_foot.py:361-373, :683-687): the 121 height measurements follow the 27
    proprioceptive terms in the ACTOR's observations (num_obs 148); the privileged buffer carries the same entries with the
    same noise draws (:340-345).  The step is pinned by the reference-generated fixture pf_rough_hactor_P_n16; here the class
    surface and one PPO iteration on the 148-wide actor."""
    from pointfoot_amd.envs import task_registry
    env_cfg, train_cfg = task_registry.get_cfgs("pointfoot_rough")
    saved = (env_cfg.terrain.measure_heights_actor, env_cfg.env.num_propriceptive_obs)
    try:
        env_cfg.terrain.measure_heights_actor = True
        env_cfg.env.num_propriceptive_obs = 27 + 121
        n = 64
        env, cfg = task_registry.make_env("pointfoot_rough", args=_args("pointfoot_rough", n), env_cfg=env_cfg)
        assert env.num_obs == 148 and env.num_privileged_obs == 148
        obs, priv = env.reset()
        assert obs.shape == (n, 148) and priv.shape == (n, 148)
        a = torch.zeros(n, 6, device=obs.device)
        _ = env.measured_heights                             # first read binds the output; the next step fills it
        for _ in range(3):
            obs, priv, rew, done, extras = env.step(a)
        torch.cuda.synchronize()
        assert torch.isfinite(obs).all() and torch.equal(obs, priv)
        h = torch.clip(env.root_states[:, 2:3] - 0.5 - env.measured_heights, -1.0, 1.0) * cfg.normalization.obs_scales.height_measurements
        noise = cfg.noise.noise_scales.height_measurements * cfg.noise.noise_level * cfg.normalization.obs_scales.height_measurements
        keep = ~done.bool()                                  # reset envs keep the pre-reset terms (stale-obs quirk, App. D.6)
        assert float((obs[keep, 27:] - h[keep]).abs().max()) <= noise + 1e-5
        runner, _ = task_registry.make_alg_runner(env, "pointfoot_rough", args=_args("pointfoot_rough", n), log_root=None)
        assert runner.alg.actor_critic.actor[0].in_features == 148
        runner.learn(num_learning_iterations=1, init_at_random_ep_len=True)
        torch.cuda.synchronize()
        assert torch.isfinite(runner.alg.params).all()
        # the reference refuses a buffer width that does not match (:363-365)
        env_cfg.env.num_propriceptive_obs = 27
        with pytest.raises(RuntimeError, match="does not match num_obs"):
            task_registry.make_env("pointfoot_rough", args=_args("pointfoot_rough", n), env_cfg=env_cfg)
    finally:
        env_cfg.terrain.measure_heights_actor, env_cfg.env.num_propriceptive_obs = saved


@pytest.mark.parametrize("task,n", [("pointfoot_rough", 96), ("pointfoot_flat", 33), ("anymal_c_rough", 40)])
def test_policy_heads_inside_the_step_launch_give_the_same_bits(task, n, monkeypatch):
    """pf_step_policy (PF_FUSED_POLICY_STEP=1): the policy's heads, the sampling, the transition store and the deferred
    process_env_step run inside the env-step launch instead of k_act_heads (rsl_rl OnPolicyRunner.learn: actions = alg.act(...);
    env.step(actions); alg.process_env_step(...), call site legged_gym/scripts/train.py:44).  Same sums in the same order, same Philox
    keys: two rollouts + one update from the same seeds must leave the SAME BITS in the storage, the env and the parameters."""
    from pointfoot_amd.envs import task_registry

    first = {}

    def run(fused):
        monkeypatch.setenv("PF_FUSED_POLICY_STEP", "1" if fused else "0")
        args = _args(task, n)
        env, _ = task_registry.make_env(task, args=args)
        runner, _ = task_registry.make_alg_runner(env, task, args=args, log_root=None)
        if "params" in first:      # the same initial policy for both runs (torch's initialisers do not restart with the env's seed)
            runner.alg.params.copy_(first["params"])
        else:
            first["params"] = runner.alg.params.clone()
        obs = env.get_observations()
        cobs = env.get_privileged_observations()
        cobs = cobs if cobs is not None else obs
        calls = {"policy": 0}
        orig = env.step_policy

        def counted(h):
            calls["policy"] += 1
            return orig(h)
        env.step_policy = counted
        out = {}
        for it in range(2):
            obs, cobs, _ = runner.rollout(obs, cobs)
            st = runner.alg.storage
            for k in ("observations", "privileged_observations", "actions", "values", "actions_log_prob", "mu", "sigma", "rewards", "dones", "returns"):
                out[f"{k}{it}"] = getattr(st, k).clone()
            out[f"ep{it}"] = runner._ep_stats.clone()
            if it == 0:
                torch.manual_seed(5)
                runner.alg.update()
        torch.cuda.synchronize()
        out["params"] = runner.alg.params.clone()
        out["obs"], out["rew"], out["reset"] = env.obs_buf.clone(), env.rew_buf.clone(), env.reset_buf.clone()
        assert calls["policy"] == (2 * runner.num_steps_per_env if fused else 0)
        return out

    a, b = run(False), run(True)
    for k in a:
        assert torch.equal(a[k], b[k]), f"{k} differs between the two-launch and the fused policy step"
