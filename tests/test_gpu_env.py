"""GPU parity tests of the fused HIP env-step kernel, called through the C ABI (libpointfoot_hip.so)."""
import ctypes as C

import numpy as np
import pytest

import common
from test_golden_python_layer import CASES, check_step

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fname,task,ctrl,noise,steps", CASES)
def test_hip_python_layer_matches_reference_golden(fname, task, ctrl, noise, steps):
    g = common.load_golden(fname)
    sim = common.sim_from_golden(g, task, ctrl, noise)
    hip = common.HipMirror(sim)
    names = list(g["reward_names"])
    counter = int(g["init_common_step_counter"])
    keep = []
    for t in range(steps):
        counter += 1
        _, arrs = common.inject_from_golden(g, t, sim, keep)
        hip.step(g[f"s{t}_actions"], counter, arrs)
        check_step(sim, g, t, task == "rough", names)
    hip.close()


def _physics_sim(task, n, seed=0, noise=True):
    import synth
    cfg = common.make_cfg(task, n, add_noise=noise)
    rough = task == "rough"
    hf = synth.synth_heightfield() if rough else None
    to = synth.synth_terrain_origins().astype(np.float32) if rough else None
    sim = common.HostSim(cfg, n, hf=hf, terrain_origins=to, seed=seed + 1)
    rs = np.random.RandomState(seed)
    if rough:
        sim.terrain_level[:] = rs.randint(0, 6, n)
        sim.terrain_type[:] = (np.arange(n) // max(n // 20, 1)) % 20
        sim.f("env_origin", 3)[:] = to[sim.terrain_level, sim.terrain_type]
    root = sim.f("root", 13)
    root[:, :3] = sim.f("env_origin", 3)
    root[:, 0:2] += rs.uniform(-1, 1, (n, 2))
    # start above the local terrain
    if rough:
        ix = np.clip(((root[:, 0] + 25) / 0.1).astype(int), 0, hf.shape[0] - 1)
        iy = np.clip(((root[:, 1] + 25) / 0.1).astype(int), 0, hf.shape[1] - 1)
        root[:, 2] = hf[ix, iy] * 0.005
    root[:, 2] += 0.70
    root[:, 3:7] = synth.rand_unit_quat(rs, n, 0.05)
    sim.f("dof_pos", 6)[:] = rs.uniform(-0.1, 0.1, (n, 6))
    sim.f("friction")[:] = rs.uniform(0.2, 1.4, n)
    sim.f("added_mass")[:] = rs.uniform(-1, 2, n)
    sim.f("added_com", 3)[:] = rs.uniform(-0.02, 0.02, (n, 3))
    sim.f("commands", 4)[:, 3] = rs.uniform(-3, 3, n)
    return sim, rs


@pytest.mark.parametrize("task", ["flat", "rough"])
def test_hip_full_step_matches_oracle(task):
    """Physics + python layer, Philox noise on both sides (bit-identical uniforms), one step from the same state,
    repeated from fresh oracle states along a 30-step rollout: obs/reward within 1e-4 (north_star tolerance)."""
    n = 256
    sim, rs = _physics_sim(task, n)
    ref = sim.clone()
    hip = common.HipMirror(sim)
    bad = 0
    for t in range(30):
        act = rs.normal(0, 0.7, (n, 6)).astype(np.float32)
        # both sides start this step from the oracle's state
        for k in common.HipMirror.FIELDS:
            getattr(sim, k)[...] = getattr(ref, k)
        hip.upload()
        ref.step("oracle", act, t + 1)
        hip.step(act, t + 1)
        same = ref.reset == sim.reset
        bad += int((~same).sum())
        m = same
        L = sim.L
        cf = slice(L.contact_forces, L.contact_forces + 27)
        keep_cols = np.ones(L.rec_floats, bool); keep_cols[cf] = False
        if task == "flat":
            np.testing.assert_allclose(sim.obs[m], ref.obs[m], atol=1e-4, rtol=0)
            np.testing.assert_allclose(sim.priv[m], ref.priv[m], atol=1e-4, rtol=0)
            np.testing.assert_allclose(sim.rew[m], ref.rew[m], atol=1e-4, rtol=1e-4)
            np.testing.assert_allclose(sim.state[m][:, keep_cols], ref.state[m][:, keep_cols], atol=5e-4, rtol=1e-3)
            # contact force = impulse / 5 ms: float32 reassociation in the solver is amplified 200x
            np.testing.assert_allclose(sim.state[m][:, cf], ref.state[m][:, cf], atol=0.5, rtol=2e-2)
        else:
            # On the (deliberately jagged) synthetic height field float32 contact dynamics are chaotic at the
            # 1e-4 level: the oracle's own float32 and float64 builds disagree on ~1 % of envs per step
            # (measured: 965 obs elements > 1e-4 over this rollout).  So: per-env agreement rate.
            e_obs = np.abs(sim.obs[m] - ref.obs[m]).max(axis=1)
            e_priv = np.abs(sim.priv[m] - ref.priv[m]).max(axis=1)
            e_rew = np.abs(sim.rew[m] - ref.rew[m])
            ok = (e_obs <= 1e-4) & (e_priv <= 1e-4) & (e_rew <= 1e-4 + 1e-4 * np.abs(ref.rew[m]))
            assert ok.mean() >= 0.97, f"step {t}: only {ok.mean():.3f} of envs within 1e-4"
            assert np.median(e_obs) <= 2e-6
    assert bad <= (2 if task == "flat" else 12), f"{bad} reset decisions differ (contact threshold flips)"
    hip.close()


@pytest.mark.parametrize("n", [1, 7, 33])
def test_hip_ragged_env_counts_match_oracle(n):
    """Odd / tiny env counts: the last wavefront holds a single robot (the other half-wave exits at once); every env
    must still be stepped exactly once and agree with the oracle (flat terrain, strict tolerance)."""
    sim, rs = _physics_sim("flat", n, seed=3)
    ref = sim.clone()
    hip = common.HipMirror(sim)
    for t in range(8):
        act = rs.normal(0, 0.7, (n, 6)).astype(np.float32)
        for k in common.HipMirror.FIELDS:
            getattr(sim, k)[...] = getattr(ref, k)
        hip.upload()
        ref.step("oracle", act, t + 1)
        hip.step(act, t + 1)
        m = ref.reset == sim.reset
        assert m.sum() >= n - 1
        np.testing.assert_allclose(sim.obs[m], ref.obs[m], atol=1e-4, rtol=0)
        np.testing.assert_allclose(sim.rew[m], ref.rew[m], atol=1e-4, rtol=1e-4)
        assert (sim.episode_length[m] == ref.episode_length[m]).all()
    hip.close()


def test_hip_rollout_tracks_oracle_statistically():
    """Free-running 100-step rollouts (chaotic contact dynamics): compare reset rate and mean reward."""
    n = 512
    sim, rs = _physics_sim("flat", n, seed=5)
    ref = sim.clone()
    hip = common.HipMirror(sim)
    r_h, r_o, k_h, k_o = [], [], 0, 0
    for t in range(100):
        act = rs.normal(0, 0.5, (n, 6)).astype(np.float32)
        ref.step("oracle", act, t + 1)
        hip.step(act, t + 1)
        r_h.append(sim.rew.mean()); r_o.append(ref.rew.mean())
        k_h += int(sim.reset.sum()); k_o += int(ref.reset.sum())
    assert abs(np.mean(r_h) - np.mean(r_o)) < 0.05 * max(1.0, abs(np.mean(r_o)))
    assert abs(k_h - k_o) <= 0.15 * max(k_o, 20)
    assert np.isfinite(sim.state).all()
    hip.close()


def test_hip_full_size_properties():
    """BASELINE size (4096 envs, rough): determinism, bounded obs, reset bookkeeping, finite state over 200 steps."""
    n = 4096
    sim, rs = _physics_sim("rough", n, seed=9)
    sim2 = sim.clone()
    hip, hip2 = common.HipMirror(sim), common.HipMirror(sim2)
    for t in range(200):
        act = rs.normal(0, 1.0, (n, 6)).astype(np.float32)
        hip.step(act, t + 1)
        if t < 20:
            hip2.step(act, t + 1)
            np.testing.assert_array_equal(sim.obs, sim2.obs)      # same seed, same inputs -> bit-identical
            np.testing.assert_array_equal(sim.state, sim2.state)
        assert np.isfinite(sim.obs).all() and np.isfinite(sim.priv).all() and np.isfinite(sim.rew).all()
        assert np.abs(sim.obs).max() <= 100.0 and np.abs(sim.priv).max() <= 100.0
        assert (sim.episode_length[sim.reset != 0] == 0).all()
        assert (sim.episode_length[sim.reset == 0] > 0).all()
        assert int(sim.stats[common._abi.PF_MAX_REWARDS]) == int(sim.reset.sum())
        assert (sim.terrain_level >= 0).all() and (sim.terrain_level < 10).all()
    assert np.isfinite(sim.state).all()
    hip.close(); hip2.close()


# ------------------------------------------------------------------------------- LeggedRobot / Anymal variant (config 5 path)
def test_hip_legged_robot_python_layer_matches_reference_golden():
    from test_golden_legged_robot import lr_sim_from_golden, lr_inject, lr_check_step
    g = common.load_golden("lr_anymal_c_rough_n32")
    sim = lr_sim_from_golden(g)
    hip = common.HipMirror(sim)
    names = list(g["reward_names"])
    counter = int(g["init_common_step_counter"])
    keep = []
    for t in range(5):
        counter += 1
        _, arrs = lr_inject(g, t, sim, keep)
        hip.step(g[f"s{t}_actions"], counter, arrs)
        lr_check_step(sim, g, t, names)
    hip.close()


def test_hip_anymal_full_step_matches_oracle():
    """17-body/12-DOF instance, actuator-network torques, LeggedRobot python layer, Philox noise: HIP vs oracle."""
    from test_golden_legged_robot import act_params
    n = 128
    cfg = common.make_cfg("anymal_c_flat", n, add_noise=True)
    sim = common.HostSim(cfg, n, actuator=act_params())
    sim.c.use_actuator_net = 1
    rs = np.random.RandomState(2)
    sim.f("root", 13)[:, 2] = 0.62
    sim.f("root", 13)[:, 0:2] = rs.uniform(-2, 2, (n, 2))
    sim.f("friction")[:] = rs.uniform(0.3, 1.2, n)
    sim.f("added_mass")[:] = rs.uniform(-5, 5, n)
    sim.f("dof_pos", 12)[:] = np.array([sim.c.default_dof_pos[i] for i in range(12)]) + rs.uniform(-0.05, 0.05, (n, 12))
    sim.episode_length[:] = rs.randint(0, 990, n)
    ref = sim.clone()
    hip = common.HipMirror(sim)
    bad = 0
    for t in range(25):
        act = rs.normal(0, 0.4, (n, 12)).astype(np.float32)
        for k in common.HipMirror.FIELDS:
            getattr(sim, k)[...] = getattr(ref, k)
        hip.upload()
        ref.step("oracle", act, t + 1)
        hip.step(act, t + 1)
        same = ref.reset == sim.reset
        bad += int((~same).sum())
        e_obs = np.abs(sim.obs[same] - ref.obs[same]).max(axis=1)
        e_rew = np.abs(sim.rew[same] - ref.rew[same])
        ok = (e_obs <= 1e-4) & (e_rew <= 1e-4 + 1e-4 * np.abs(ref.rew[same]))
        assert ok.mean() >= 0.97, f"step {t}: {ok.mean():.3f} of envs within 1e-4 (max {e_obs.max():.2e})"
        L = sim.L
        np.testing.assert_allclose(sim.state[same][:, L.lstm_state:L.lstm_state + 384], ref.state[same][:, L.lstm_state:L.lstm_state + 384], atol=1e-4)
    assert bad <= 3
    hip.close()
