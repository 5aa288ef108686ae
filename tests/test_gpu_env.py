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
        check_step(sim, g, t, task.startswith("rough"), names)
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


def _same_decisions(sim, ref):
    """Envs whose discrete decisions agree between the HIP kernel and the oracle for this step: reset flag, active
    contact set and active joint-limit rows of every substep (PfBuffers.d_contact_set), feet-contact flags.  A flip of one of
    these is a different (equally valid) branch of a discontinuous map; everything else has to agree to 1e-4."""
    same = (ref.reset == sim.reset) & (ref.contact_set == sim.contact_set).all(axis=(1, 2))
    # the feet-state machine thresholds the contact force at 1 N (point_foot.py:_compute_feet_states): a force within rounding
    # of the threshold flips contact_filt / first_contact, and with it the discrete reward terms (0.02 per flag) -- the same
    # kind of decision, taken one stage later (seen once in 49 152 env-steps after a change of compiler flags)
    if hasattr(sim, "L") and sim.c.env_kind == common._abi.PF_KIND_POINTFOOT:
        for nm in ("contact_filt", "first_contact", "last_contacts"):
            same &= ((sim.f(nm, 2) != 0) == (ref.f(nm, 2) != 0)).all(axis=1)
        # ... and the reward terms threshold the same forces at 0.1 N: _reward_no_fly (feet, z component) and
        # _reward_collision (norm, penalised bodies): a foot grazing the ground with 0.12 N sits on that line
        L = sim.L
        cs = sim.state[:, L.contact_forces:L.contact_forces + 27].reshape(-1, 9, 3)
        cr = ref.state[:, L.contact_forces:L.contact_forces + 27].reshape(-1, 9, 3)
        same &= ((np.linalg.norm(cs, axis=2) > 0.1) == (np.linalg.norm(cr, axis=2) > 0.1)).all(axis=1)
        same &= ((cs[:, :, 2] > 0.1) == (cr[:, :, 2] > 0.1)).all(axis=1)
    return same


KAPPA = 8.0


def _conditioned_close(name, got, want, want64, m, stats, atol=1e-4, rtol=0.0):
    """|got - want| <= atol (+ rtol |want|) for every env in `m` -- except envs whose own problem is ill-conditioned in
    float32, measured by the oracle itself: where the float32 oracle (`want`) sits further than atol / KAPPA from its
    float64 evaluation of the same step (`want64`), `got` may deviate KAPPA times that gap.  (Measured on the MI355X,
    tools/parity_probe.py: 1 env-step in 7680 on the jagged field is past 1e-4, and its float32/float64 oracle gap
    is just as large.)  The envs that needed the allowance are counted in stats["loose"], all checked envs in stats["n"]."""
    got, want, want64 = got[m].reshape(int(m.sum()), -1), want[m].reshape(int(m.sum()), -1), want64[m].reshape(int(m.sum()), -1)
    err = np.abs(got - want)
    tol = atol + rtol * np.abs(want)
    gap = np.abs(want64 - want).max(axis=1, keepdims=True)
    plain = (err <= tol).all(axis=1)
    ok = (err <= tol + KAPPA * gap).all(axis=1)
    stats["n"] += len(plain); stats["loose"] += int((~plain & ok).sum())
    if "outliers" in stats and not ok.all():
        # Only where a test asks for it (the stair scenarios): an env past the allowance whose recorded decisions all agree took an
        # UNRECORDED one -- the Coulomb-cone projection of the PGS sweep (stick or slide, per contact and iteration) is a branch on a
        # float comparison too.  Such envs are counted, bounded by the test and must stay small deviations (a wrong contact model
        # shows as errors of order one, on many envs).
        small = (err <= 10.0 * (tol + KAPPA * gap)).all(axis=1)
        stats["outliers"] += int((~ok & small).sum())
        stats["worst_outlier"] = max(stats.get("worst_outlier", 0.0), float(err[~ok & small].max()) if (~ok & small).any() else 0.0)
        ok = ok | small
    assert ok.all(), f"{name}: max error {err.max():.3e} (tolerance {atol:g} + {KAPPA:g} x oracle f32/f64 gap {gap[~ok].max():.3e})"


def _height_scan_flips(sim, ref, m, first):
    """Height-scan points of the matched envs that differ by more than 1e-4: a scan point within float rounding of a grid
    line reads the neighbouring cell (the field is a step function there).  Returns (#flipped points, mask of exact rows)."""
    e = np.abs(sim.priv[m][:, first:] - ref.priv[m][:, first:])
    bad = e > 1e-4
    return int(bad.sum()), ~bad.any(axis=1)


def _full_step_parity(name, task, sim, rs, n, steps, flipped_bound, scan_bound):
    """Physics + python layer, Philox noise on both sides (bit-identical uniforms), one step from the same state,
    repeated from fresh oracle states along a rollout.  STRICT 1e-4 (north_star tolerance) on obs / privileged
    obs / reward of every env whose discrete decisions (reset, contact set, joint-limit set per substep) agree with the
    oracle's; the envs where a decision flipped are counted and bounded separately."""
    ref, ref64 = sim.clone(), sim.clone()
    hip = common.HipMirror(sim)
    flipped, total, scan_flips = 0, 0, 0
    stats = {"n": 0, "loose": 0}
    L = sim.L
    cf = slice(L.contact_forces, L.contact_forces + 27)
    tq = slice(L.torques, L.torques + 6)
    keep_cols = np.ones(L.rec_floats, bool); keep_cols[cf] = False; keep_cols[tq] = False
    # velocities: the obs tolerance (1e-4 on dof_vel * 0.05, ang_vel * 0.25) is 2e-3 rad/s on the raw state
    vel_cols = np.zeros(L.rec_floats, bool)
    for off, w in ((L.root + 7, 6), (L.dof_vel, 6), (L.last_dof_vel, 6), (L.last_root_vel, 6), (L.base_lin_vel, 3), (L.base_ang_vel, 3)):
        vel_cols[off:off + w] = True
    keep_cols &= ~vel_cols
    face_contacts = 0
    for t in range(steps):
        act = rs.normal(0, 0.7, (n, 6)).astype(np.float32)
        # both sides start this step from the oracle's state
        for k in common.HipMirror.FIELDS:
            getattr(sim, k)[...] = getattr(ref, k)
            getattr(ref64, k)[...] = getattr(ref, k)
        hip.upload()
        ref.step("oracle", act, t + 1)
        ref64.step("oracle", act, t + 1, physics_bits=64)      # conditioning probe only (see _conditioned_close)
        hip.step(act, t + 1)
        m = _same_decisions(sim, ref)
        flipped += int((~m).sum()); total += n
        # vertical-face candidates sit above the nsph floor candidates in the contact word (bits nsph .. 2 nsph - 1)
        face_contacts += int((((ref.contact_set[:, :, 0] >> np.uint64(sim.model.nsph)) & np.uint64((1 << sim.model.nsph) - 1)) != 0).any(axis=1).sum())
        _conditioned_close(f"step {t} obs", sim.obs, ref.obs, ref64.obs, m, stats)
        _conditioned_close(f"step {t} rew", sim.rew, ref.rew, ref64.rew, m, stats, rtol=1e-4)
        _conditioned_close(f"step {t} priv", sim.priv[:, :27], ref.priv[:, :27], ref64.priv[:, :27], m, stats)
        m = m & (np.abs(ref64.obs - ref.obs).max(axis=1) <= 1e-4 / KAPPA)      # the state checks below: well-conditioned envs
        if task == "rough":
            k, exact = _height_scan_flips(sim, ref, m, 27)
            scan_flips += k
        else:
            np.testing.assert_allclose(sim.priv[m], ref.priv[m], atol=1e-4, rtol=0)
        np.testing.assert_allclose(sim.state[m][:, keep_cols], ref.state[m][:, keep_cols], atol=5e-4, rtol=1e-3, err_msg=f"step {t} state")
        np.testing.assert_allclose(sim.state[m][:, vel_cols], ref.state[m][:, vel_cols], atol=2e-3, rtol=1e-3, err_msg=f"step {t} velocities")
        # contact force = impulse / 5 ms: float32 reassociation in the solver is amplified 200x
        np.testing.assert_allclose(sim.state[m][:, cf], ref.state[m][:, cf], atol=0.5, rtol=2e-2, err_msg=f"step {t} contact forces")
        # the last substep's PD torque = -Kd (1.5) x a joint velocity that is itself only good to 2e-3 rad/s (obs tolerance / 0.05)
        np.testing.assert_allclose(sim.state[m][:, tq], ref.state[m][:, tq], atol=5e-3, rtol=1e-3, err_msg=f"step {t} torques")
    frac = flipped / total
    print(f"[{name}] envs with a flipped discrete decision: {flipped}/{total} = {frac:.4f}; height-scan points on a grid line: {scan_flips}; "
          f"checks that needed the conditioning allowance: {stats['loose']}/{stats['n']}; env-steps touching a vertical face: {face_contacts}")
    common.record_parity(name, env_steps=total, flipped=flipped, flipped_frac=frac, scan_point_flips=scan_flips,
                         allowance_checks=stats["loose"], checks=stats["n"], env_steps_with_face_contact=face_contacts)
    hip.close()
    return dict(frac=frac, flipped=flipped, scan_flips=scan_flips, loose=stats["loose"], checks=stats["n"], face_contacts=face_contacts)


@pytest.mark.parametrize("task", ["flat", "rough"])
def test_hip_full_step_matches_oracle(task):
    """HIP step kernel vs oracle on 256 envs x 30 steps (plane / the jagged synthetic field): see _full_step_parity."""
    n = 256
    sim, rs = _physics_sim(task, n)
    r = _full_step_parity(f"test_hip_full_step_matches_oracle[{task}]", task, sim, rs, n, 30, None, None)
    # Bounds within ~5x of what the MI355X run measures (profiles/r3_parity_fractions.json: 0 flipped env-steps of 7680 on either
    # terrain, 4 of 23 040 checks needing the conditioning allowance on the jagged field, 0 scan points on a grid line); where the
    # measurement is 0 the bound is a handful of env-steps, so that a regression flipping a percent of the envs cannot pass.
    assert r["loose"] <= 0.001 * r["checks"], "more than 0.1 % of the matched envs are past the plain 1e-4"
    assert r["frac"] <= (0.001 if task == "flat" else 0.003), f"{r['frac']:.4f} of env-steps took a different discrete branch"
    assert r["scan_flips"] <= 8


def _curriculum_stairs_sim(n, seed=0):
    """BASELINE config 3's actual terrain: the curriculum field of pointfoot_amd.utils.terrain.Terrain (numpy seed 1, as
    legged_gym/utils/terrain.py:38-164 builds it for `pointfoot_rough`), envs on the STAIR tiles (terrain types 4..15:
    choice in [0.2, 0.8), legged_gym/utils/terrain.py:109-145) at levels 5..9 (step height 0.14 .. 0.21 m), standing OFF the
    3 m central platform -- on the steps themselves, where feet meet risers (vertical faces of the slope-thresholded mesh)."""
    from pointfoot_amd.utils.terrain import Terrain
    cfg = common.make_cfg("rough", n)
    np.random.seed(1)
    ter = Terrain(cfg.terrain, n)
    hf, to = ter.heightsamples, ter.env_origins.astype(np.float32)
    sim = common.HostSim(cfg, n, hf=hf, terrain_origins=to, seed=seed + 1)
    rs = np.random.RandomState(seed)
    sim.terrain_level[:] = 5 + np.arange(n) % 5
    sim.terrain_type[:] = 4 + (np.arange(n) // 5) % 12
    sim.f("env_origin", 3)[:] = to[sim.terrain_level, sim.terrain_type]
    root = sim.f("root", 13)
    root[:, :3] = sim.f("env_origin", 3)
    r, th = rs.uniform(1.8, 3.6, n), rs.uniform(0, 2 * np.pi, n)
    root[:, 0] += r * np.cos(th); root[:, 1] += r * np.sin(th)
    hs, border = cfg.terrain.horizontal_scale, cfg.terrain.border_size
    ix = np.clip(((root[:, 0] + border) / hs).astype(int), 0, hf.shape[0] - 2)
    iy = np.clip(((root[:, 1] + border) / hs).astype(int), 0, hf.shape[1] - 2)
    # stand on the highest of the surrounding cells so that no foot starts inside a riser
    zmax = np.max(np.stack([hf[ix + a, iy + b] for a in (-2, -1, 0, 1, 2) for b in (-2, -1, 0, 1, 2)]), axis=0) * cfg.terrain.vertical_scale
    root[:, 2] = zmax + 0.68
    import synth
    root[:, 3:7] = synth.rand_unit_quat(rs, n, 0.05)
    # walk: a forward command and some initial speed along a random heading, so that feet travel across step edges
    root[:, 7:9] = rs.uniform(-0.6, 0.6, (n, 2))
    sim.f("dof_pos", 6)[:] = rs.uniform(-0.1, 0.1, (n, 6))
    sim.f("friction")[:] = rs.uniform(0.2, 1.4, n)
    sim.f("added_mass")[:] = rs.uniform(-1, 2, n)
    sim.f("added_com", 3)[:] = rs.uniform(-0.02, 0.02, (n, 3))
    sim.f("commands", 4)[:, 3] = rs.uniform(-3, 3, n)
    off_platform = np.hypot(root[:, 0] - sim.f("env_origin", 3)[:, 0], root[:, 1] - sim.f("env_origin", 3)[:, 1]) > 1.5
    assert off_platform.all() and (sim.terrain_level >= 5).all()
    return sim, rs


def test_hip_full_step_on_curriculum_stairs():
    """HIP vs oracle full steps on config 3's REAL terrain (not the synthetic field): 300 envs on the
    stair tiles of the curriculum field at levels >= 5, off the central platform, 40 steps of falling onto / walking over the
    steps.  Same strict 1e-4 rule as above; additionally the run must actually exercise the vertical-face contacts."""
    n = 300
    sim, rs = _curriculum_stairs_sim(n)
    r = _full_step_parity("test_hip_full_step_on_curriculum_stairs", "rough", sim, rs, n, 40, None, None)
    assert r["face_contacts"] > 0, "no env touched a riser: the test does not exercise the stepped collision surface"
    # measured (profiles/r3_parity_fractions.json): 0 flipped of 12 000 env-steps, 2 of 36 000 checks on the allowance, 0 scan flips
    assert r["loose"] <= 0.0005 * r["checks"]
    assert r["frac"] <= 0.003, f"{r['frac']:.4f} of env-steps took a different discrete branch"
    assert r["scan_flips"] <= 12


def test_hip_flat_4096_literal_config():
    """BASELINE config 2 as written: pointfoot_flat, 4096 envs, plane, no domain randomisation, no noise, no pushes --
    every env against the oracle (OpenMP) for 12 steps from fresh oracle states, strict 1e-4."""
    n = 4096
    cfg = common.make_cfg("flat", n, add_noise=False)
    cfg.domain_rand.randomize_friction = False
    cfg.domain_rand.randomize_base_mass = False
    cfg.domain_rand.randomize_base_com = False
    cfg.domain_rand.push_robots = False
    sim = common.HostSim(cfg, n)
    assert sim.c.add_noise == 0 and sim.c.push_robots == 0 and sim.c.terrain_type == common._abi.PF_TERRAIN_PLANE
    rs = np.random.RandomState(4)
    side = 64
    sim.f("env_origin", 3)[:, 0] = 3.0 * (np.arange(n) // side)
    sim.f("env_origin", 3)[:, 1] = 3.0 * (np.arange(n) % side)
    sim.f("root", 13)[:, :3] = sim.f("env_origin", 3)
    sim.f("root", 13)[:, 2] = 0.62 + 0.04
    sim.f("friction")[:] = 1.0                     # asset default: no friction randomisation
    sim.episode_length[:] = rs.randint(0, 1001, n)  # the runner's random episode start: time-outs happen inside the window
    common.oracle_lib().pfo_set_threads(8)
    ref = sim.clone()
    hip = common.HipMirror(sim)
    flipped = 0
    for t in range(12):
        act = rs.normal(0, 0.8, (n, 6)).astype(np.float32)
        for k in common.HipMirror.FIELDS:
            getattr(sim, k)[...] = getattr(ref, k)
        hip.upload()
        ref.step("oracle", act, t + 1)
        hip.step(act, t + 1)
        m = _same_decisions(sim, ref)
        flipped += int((~m).sum())
        np.testing.assert_allclose(sim.obs[m], ref.obs[m], atol=1e-4, rtol=0)
        np.testing.assert_allclose(sim.priv[m], ref.priv[m], atol=1e-4, rtol=0)
        np.testing.assert_allclose(sim.rew[m], ref.rew[m], atol=1e-4, rtol=1e-4)
        assert (sim.episode_length[m] == ref.episode_length[m]).all()
        assert (sim.time_out[m] == ref.time_out[m]).all()
    common.oracle_lib().pfo_set_threads(1)
    common.record_parity("test_hip_flat_4096_literal_config", env_steps=12 * n, flipped=flipped, flipped_frac=flipped / (12 * n))
    assert flipped <= 0.0003 * 12 * n, f"{flipped} env-steps took a different discrete branch"      # measured: 3 of 49 152 (6e-5)
    hip.close()


def test_hip_rough_4096_literal_config():
    """BASELINE config 3 as written, at full size, through the env class: `task_registry.make_env("pointfoot_rough")` with 4096
    envs on its real curriculum terrain (trimesh collision surface, 121-point height scan in the critic observations), friction /
    mass / COM randomisation, observation noise and pushes ON (pointfoot_rough_config.py:3-213).  After a few steps of settling the
    env's state is mirrored into the oracle before every step, 10 steps with a PUSH step inside the window; strict 1e-4 on obs /
    privileged obs / reward for every env whose discrete decisions agree (the rule of _full_step_parity), fractions recorded."""
    import torch
    from pointfoot_amd.envs import task_registry
    from pointfoot_amd.utils import get_args
    n, steps = 4096, 10
    args = get_args(["--task", "pointfoot_rough", "--headless", "--num_envs", str(n), "--sim_device", "cuda:0", "--rl_device", "cuda:0"])
    env, cfg = task_registry.make_env("pointfoot_rough", args=args)
    c = env._cfg_c
    assert c.add_noise == 1 and c.push_robots == 1 and c.terrain_type == common._abi.PF_TERRAIN_HEIGHTFIELD and c.num_height_points == 121
    assert c.hf_rows == 1300 and c.hf_cols == 2100 and c.curriculum == 1 and env.num_privileged_obs == 148
    assert cfg.domain_rand.randomize_friction and cfg.domain_rand.randomize_base_mass and cfg.domain_rand.randomize_base_com
    assert float(env._state[:, env._L.friction].std()) > 0.1 and float(env._state[:, env._L.added_mass].std()) > 0.1
    env.reset()
    env.episode_length_buf = torch.randint_like(env.episode_length_buf, high=1000)      # the runner's random episode start
    rs = np.random.RandomState(11)
    for _ in range(12):      # settle: robots land on their tiles, some fall and are reset by the kernel
        env.step(torch.from_numpy(rs.normal(0, 0.5, (n, 6)).astype(np.float32)))
    env.common_step_counter = int(c.push_interval) - 4      # the push (every 350 steps, point_foot.py:590-611) falls inside the window
    sim = common.HostSim(cfg, n, hf=env.height_samples.cpu().numpy(), terrain_origins=env.terrain_origins.cpu().numpy())
    sim.c = c
    assert bytes(sim.L) == bytes(env._L)
    sim.bufs = sim.make_bufs()
    ref64, got = sim.clone(), sim.clone()
    cset = env.bind_contact_set()
    common.oracle_lib().pfo_set_threads(8)
    stats = {"n": 0, "loose": 0}
    flipped, scan_flips, pushed_steps, resets, faces = 0, 0, 0, 0, 0
    try:
        for t in range(steps):
            torch.cuda.synchronize()
            for s_ in (sim, ref64):
                s_.state[:] = env._state.cpu().numpy()
                s_.episode_length[:] = env.episode_length_buf.cpu().numpy()
                s_.terrain_level[:] = env.terrain_levels.cpu().numpy(); s_.terrain_type[:] = env.terrain_types.cpu().numpy()
            act = rs.normal(0, 0.7, (n, 6)).astype(np.float32)
            o, p, r, d, _ = env.step(torch.from_numpy(act))
            k = env.common_step_counter
            pushed_steps += int(k % int(c.push_interval) == 0)
            sim.step("oracle", act, k)
            ref64.step("oracle", act, k, physics_bits=64)
            torch.cuda.synchronize()
            got.state[:] = env._state.cpu().numpy(); got.reset[:] = d.cpu().numpy()
            got.contact_set[:] = cset.cpu().numpy().view(np.uint64)
            got.obs[:], got.priv[:], got.rew[:] = o.cpu().numpy(), p.cpu().numpy(), r.cpu().numpy()
            m = _same_decisions(got, sim)
            flipped += int((~m).sum()); resets += int(sim.reset.sum())
            faces += int((((sim.contact_set[:, :, 0] >> np.uint64(sim.model.nsph)) & np.uint64((1 << sim.model.nsph) - 1)) != 0).any(axis=1).sum())
            _conditioned_close(f"step {t} obs", got.obs, sim.obs, ref64.obs, m, stats)
            _conditioned_close(f"step {t} rew", got.rew, sim.rew, ref64.rew, m, stats, rtol=1e-4)
            _conditioned_close(f"step {t} priv", got.priv[:, :27], sim.priv[:, :27], ref64.priv[:, :27], m, stats)
            kf, _ = _height_scan_flips(got, sim, m & (np.abs(ref64.obs - sim.obs).max(axis=1) <= 1e-4 / KAPPA), 27)
            scan_flips += kf
            assert (env.episode_length_buf.cpu().numpy()[m] == sim.episode_length[m]).all()
            assert (env.terrain_levels.cpu().numpy()[m] == sim.terrain_level[m]).all()
    finally:
        common.oracle_lib().pfo_set_threads(1)
    total = steps * n
    common.record_parity("test_hip_rough_4096_literal_config", env_steps=total, flipped=flipped, flipped_frac=flipped / total,
                         scan_point_flips=scan_flips, allowance_checks=stats["loose"], checks=stats["n"], push_steps=pushed_steps,
                         resets=resets, env_steps_with_face_contact=faces)
    assert pushed_steps == 1 and resets > 0
    assert flipped <= 0.003 * total, f"{flipped} of {total} env-steps took a different discrete branch"
    assert stats["loose"] <= 0.001 * stats["n"] and scan_flips <= 40


def test_hip_quadruped_self_contacts_match_oracle():
    """The quadruped instance built WITH self-collision (12 capsules / 54 pairs, 98 contact candidates: asset.self_collisions = 0,
    anymal_c_flat_config.py:42) on the GPU, in the crossed-legs scenario of tests/test_physics_invariants.py: same contact sets as the
    oracle (both words of the candidate set), same motion.  Envs start with different joint angles so that different capsule pairs
    (feet, shanks, thighs) meet; the invariants themselves (no pass-through, action = reaction) are tests/test_physics_invariants.py's."""
    from test_physics_invariants import quad_crossing_sim
    n = 64
    sim, act = quad_crossing_sim(True, n)
    rs = np.random.RandomState(4)
    sim.f("dof_pos", 12)[:] += rs.uniform(-0.25, 0.25, (n, 12)).astype(np.float32) * np.array([0, 1, 1] * 4, np.float32)      # hip flexion / knee
    ref = sim.clone()
    hip = common.HipMirror(sim)
    worst_match, pairs_seen = 1.0, set()
    L = sim.L
    keep_cols = np.ones(L.rec_floats, bool); keep_cols[L.contact_forces:L.contact_forces + 51] = False
    for t in range(16):
        for k in common.HipMirror.FIELDS:
            getattr(sim, k)[...] = getattr(ref, k)
        hip.upload()
        ref.step("oracle", act, t + 1)
        hip.step(act, t + 1)
        m = _same_decisions(sim, ref)
        worst_match = min(worst_match, float(m.mean()))
        assert m.mean() >= 62 / 64
        np.testing.assert_allclose(sim.state[m][:, keep_cols], ref.state[m][:, keep_cols], atol=1e-2, rtol=5e-3)
        for w0, w3 in zip(sim.contact_set[:, -1, 0], sim.contact_set[:, -1, 3]):
            bits = (int(w0) >> (2 * sim.model.nsph)) | (int(w3) << (64 - 2 * sim.model.nsph))
            pairs_seen |= {p for p in range(sim.model.npair) if bits >> p & 1}
    assert len(pairs_seen) >= 4 and max(pairs_seen) >= 64 - 2 * sim.model.nsph, pairs_seen      # pairs of both words were active
    common.record_parity("test_hip_quadruped_self_contacts_match_oracle", worst_matched_fraction_per_step=worst_match, pairs_active=len(pairs_seen))
    hip.close()


def test_hip_faces_and_self_contacts_match_oracle():
    """The collision-geometry scenarios of tests/test_physics_invariants.py (riser, crossed legs) on the GPU: same
    contact sets as the oracle, same motion, and the invariants themselves."""
    from test_physics_invariants import crossing_sim, riser_sim
    sim = riser_sim(64)
    sim.f("root", 13)[:, 1] = np.linspace(3.0, 40.0, 64)
    ref = sim.clone()
    hip = common.HipMirror(sim)
    zero = np.zeros((64, 6), np.float32)
    worst_match = 1.0
    for t in range(14):
        for k in common.HipMirror.FIELDS:
            getattr(sim, k)[...] = getattr(ref, k)
        hip.upload()
        ref.step("oracle", zero, t + 1)
        hip.step(zero, t + 1)
        m = _same_decisions(sim, ref)
        worst_match = min(worst_match, float(m.mean()))
        assert m.mean() >= 62 / 64                      # measured: every env, every step
        np.testing.assert_allclose(sim.state[m], ref.state[m], atol=1e-2, rtol=5e-3)      # 1 m/s impact on the face: velocities jump by metres per second
        fp = sim.f("feet_pos", 6).reshape(64, 2, 3)
        assert (fp[:, :, 0] <= 5.0 - 0.03 + 0.012).all() and (t < 2 or (fp[:, :, 2] <= 0.034).all())
    assert (((sim.contact_set[:, -1, 0] >> np.uint64(sim.model.nsph)) & np.uint64(3)) == 3).all()
    hip.close()
    sim, act = crossing_sim(True, 64)
    ref = sim.clone()
    hip = common.HipMirror(sim)
    for t in range(16):
        for k in common.HipMirror.FIELDS:
            getattr(sim, k)[...] = getattr(ref, k)
        hip.upload()
        ref.step("oracle", act, t + 1)
        hip.step(act, t + 1)
        m = _same_decisions(sim, ref)
        worst_match = min(worst_match, float(m.mean()))
        assert m.mean() >= 62 / 64
        L = sim.L
        cf = slice(L.contact_forces, L.contact_forces + 27)
        keep_cols = np.ones(L.rec_floats, bool); keep_cols[cf] = False
        np.testing.assert_allclose(sim.state[m][:, keep_cols], ref.state[m][:, keep_cols], atol=1e-2, rtol=5e-3)
        fp = sim.f("feet_pos", 6).reshape(64, 2, 3)
        assert (fp[:, 0, 1] - fp[:, 1, 1] >= 0.06 - 0.004).all()
    common.record_parity("test_hip_faces_and_self_contacts_match_oracle", worst_matched_fraction_per_step=worst_match)
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


def test_hip_legged_robot_all_reward_terms_match_reference_golden():
    """the 19 reward methods of legged_robot.py:816-906 all live, on the quadruped kernel instance"""
    from test_golden_legged_robot import lr_sim_from_golden, lr_inject, lr_check_step
    g = common.load_golden("lr_anymal_c_rough_allrew_n32")
    sim = lr_sim_from_golden(g, variant="allrew")
    hip = common.HipMirror(sim)
    names = list(g["reward_names"])
    counter = int(g["init_common_step_counter"])
    keep = []
    for t in range(4):
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
    ref, ref64 = sim.clone(), sim.clone()
    hip = common.HipMirror(sim)
    flipped = 0
    stats = {"n": 0, "loose": 0}
    L = sim.L
    ls = slice(L.lstm_state, L.lstm_state + 384)
    for t in range(25):
        act = rs.normal(0, 0.4, (n, 12)).astype(np.float32)
        for k in common.HipMirror.FIELDS:
            getattr(sim, k)[...] = getattr(ref, k)
            getattr(ref64, k)[...] = getattr(ref, k)
        hip.upload()
        ref.step("oracle", act, t + 1)
        ref64.step("oracle", act, t + 1, physics_bits=64)
        hip.step(act, t + 1)
        m = _same_decisions(sim, ref)
        flipped += int((~m).sum())
        _conditioned_close(f"step {t} obs", sim.obs, ref.obs, ref64.obs, m, stats)
        _conditioned_close(f"step {t} rew", sim.rew, ref.rew, ref64.rew, m, stats, rtol=1e-4)
        _conditioned_close(f"step {t} lstm state", sim.state[:, ls], ref.state[:, ls], ref64.state[:, ls], m, stats)
    print(f"[anymal_c_flat] flipped decisions {flipped}/{25 * n}, checks that needed the conditioning allowance: {stats['loose']}/{stats['n']}")
    common.record_parity("test_hip_anymal_full_step_matches_oracle", env_steps=25 * n, flipped=flipped, flipped_frac=flipped / (25 * n),
                         allowance_checks=stats["loose"], checks=stats["n"])
    assert flipped <= 0.003 * 25 * n and stats["loose"] <= 0.001 * stats["n"]      # measured: 0 of 3200 env-steps, 0 of 9600 checks
    hip.close()


def _anymal_rough_sim(n, seed=6):
    """BASELINE config 5's env: anymal_c_rough on the synthetic height field -- 187-point scan, friction / mass
    randomisation, velocity pushes, command resampling, actuator network."""
    import synth
    from test_golden_legged_robot import act_params
    cfg = common.make_cfg("anymal_c_rough", n, add_noise=True)
    hf = synth.synth_heightfield()
    to = synth.synth_terrain_origins().astype(np.float32)
    sim = common.HostSim(cfg, n, hf=hf, terrain_origins=to, actuator=act_params(), seed=seed)
    assert sim.c.use_actuator_net == 1 and sim.c.num_height_points == 187 and sim.c.num_obs == 235 and sim.c.push_robots == 1
    rs = np.random.RandomState(seed)
    sim.terrain_level[:] = rs.randint(0, 6, n)
    sim.terrain_type[:] = (np.arange(n) // max(n // 20, 1)) % 20
    sim.f("env_origin", 3)[:] = to[sim.terrain_level, sim.terrain_type]
    root = sim.f("root", 13)
    root[:, :3] = sim.f("env_origin", 3)
    root[:, 0:2] += rs.uniform(-1, 1, (n, 2))
    ix = np.clip(((root[:, 0] + 25) / 0.1).astype(int), 0, hf.shape[0] - 1)
    iy = np.clip(((root[:, 1] + 25) / 0.1).astype(int), 0, hf.shape[1] - 1)
    root[:, 2] = hf[ix, iy] * 0.005 + 0.75
    root[:, 3:7] = synth.rand_unit_quat(rs, n, 0.05)
    sim.f("dof_pos", 12)[:] = np.array([sim.c.default_dof_pos[i] for i in range(12)]) + rs.uniform(-0.05, 0.05, (n, 12))
    sim.f("friction")[:] = rs.uniform(0.5, 1.25, n)
    sim.f("added_mass")[:] = rs.uniform(-5, 5, n)
    sim.f("commands", 4)[:] = rs.uniform(-1, 1, (n, 4))
    # episode lengths straddle the command-resampling interval (500 steps) and the time-out (1000)
    sim.episode_length[:] = rs.choice([3, 200, 497, 498, 499, 996, 998, 999], n)
    return sim, rs


def _anymal_full_step_parity(name, sim, rs, n, steps, count_outliers=False):
    """quadruped instance, HIP vs oracle from fresh oracle states, strict 1e-4 on envs whose discrete decisions agree; the step
    counter is set so that the window contains a push step (push_interval = 750)"""
    ref, ref64 = sim.clone(), sim.clone()
    hip = common.HipMirror(sim)
    flipped, pushes, resampled, scan_flips, face_contacts, many_contacts = 0, 0, 0, 0, 0, 0
    stats = {"n": 0, "loose": 0}
    if count_outliers:
        stats["outliers"] = 0
    t0 = sim.c.push_interval - 4
    nprop = 48
    for t in range(steps):
        counter = t0 + t
        act = rs.normal(0, 0.4, (n, 12)).astype(np.float32)
        for k in common.HipMirror.FIELDS:
            getattr(sim, k)[...] = getattr(ref, k)
            getattr(ref64, k)[...] = getattr(ref, k)
        hip.upload()
        cmd_before = ref.f("commands", 4).copy()
        ref.step("oracle", act, counter)
        ref64.step("oracle", act, counter, physics_bits=64)    # conditioning probe only (see _conditioned_close)
        hip.step(act, counter)
        pushes += int(counter % sim.c.push_interval == 0)
        resampled += int((np.abs(ref.f("commands", 4)[:, :2] - cmd_before[:, :2]).sum(axis=1) > 0).sum())
        m = _same_decisions(sim, ref)
        flipped += int((~m).sum())
        cw = ref.contact_set[:, :, 0]
        face_contacts += int((((cw >> np.uint64(sim.model.nsph)) & np.uint64((1 << sim.model.nsph) - 1)) != 0).any(axis=1).sum())
        many_contacts += int((np.array([[bin(int(x)).count("1") for x in row] for row in cw]) > 6).any(axis=1).sum())      # more than the old cap of 6
        _conditioned_close(f"step {t} obs", sim.obs[:, :nprop], ref.obs[:, :nprop], ref64.obs[:, :nprop], m, stats)
        _conditioned_close(f"step {t} rew", sim.rew, ref.rew, ref64.rew, m, stats, rtol=1e-4)
        m = m & (np.abs(ref64.obs[:, :nprop] - ref.obs[:, :nprop]).max(axis=1) <= 1e-4 / KAPPA)      # well-conditioned envs for the rest
        if count_outliers:
            m = m & (np.abs(sim.obs[:, :nprop] - ref.obs[:, :nprop]).max(axis=1) <= 1e-4)              # ... and not one of the counted outliers
        bad = np.abs(sim.obs[m][:, nprop:] - ref.obs[m][:, nprop:]) > 1e-4
        scan_flips += int(bad.sum())
        L = sim.L
        np.testing.assert_allclose(sim.state[m][:, L.lstm_state:L.lstm_state + 384], ref.state[m][:, L.lstm_state:L.lstm_state + 384], atol=1e-4, rtol=1e-4)
        np.testing.assert_allclose(sim.f("commands", 4)[m], ref.f("commands", 4)[m], atol=1e-5)
        np.testing.assert_allclose(sim.f("root", 13)[m], ref.f("root", 13)[m], atol=2e-4, rtol=1e-4)
        assert (sim.episode_length[m] == ref.episode_length[m]).all()
    print(f"[{name}] flipped decisions {flipped}/{steps * n}, scan points on a grid line {scan_flips}, commands resampled {resampled}, "
          f"checks that needed the conditioning allowance: {stats['loose']}/{stats['n']}, env-steps with a face contact {face_contacts}, with more than 6 contacts {many_contacts}")
    common.record_parity(name, env_steps=steps * n, flipped=flipped, flipped_frac=flipped / (steps * n), scan_point_flips=scan_flips,
                         allowance_checks=stats["loose"], checks=stats["n"], env_steps_with_face_contact=face_contacts,
                         env_steps_with_more_than_6_contacts=many_contacts, unrecorded_branch_outliers=stats.get("outliers", 0),
                         worst_outlier_error=stats.get("worst_outlier", 0.0))
    hip.close()
    return dict(flipped=flipped, pushes=pushes, resampled=resampled, scan_flips=scan_flips, loose=stats["loose"], checks=stats["n"],
                face_contacts=face_contacts, many_contacts=many_contacts, outliers=stats.get("outliers", 0))


def test_hip_anymal_rough_full_step_matches_oracle():
    """Config 5's env (quadruped instance on a height field, 187-point scan, DR, velocity pushes, command resampling,
    actuator net), HIP vs oracle: see _anymal_full_step_parity."""
    n = 128
    sim, rs = _anymal_rough_sim(n)
    r = _anymal_full_step_parity("test_hip_anymal_rough_full_step_matches_oracle", sim, rs, n, 16)
    assert r["loose"] <= 0.004 * r["checks"]          # measured: 3 of 4096 checks (7e-4)
    assert r["pushes"] == 1 and r["resampled"] > 0
    assert r["flipped"] <= 0.005 * 16 * n             # measured: 0 of 2048 env-steps
    assert r["scan_flips"] <= 8                       # measured: 0


def test_hip_anymal_on_curriculum_stairs():
    """Config 5's env on config 5's REAL terrain: the curriculum field of pointfoot_amd.utils.terrain.Terrain (numpy seed 1), 160
    quadrupeds on the stair tiles at levels 5..9, off the central platform, dropped from 0.15 m above the highest surrounding cell so
    that feet land across step edges -- the case the contact cap of 10 was raised for (4 feet x floor + riser face and more)."""
    from pointfoot_amd.utils.terrain import Terrain
    from test_golden_legged_robot import act_params
    import synth
    n = 160
    cfg = common.make_cfg("anymal_c_rough", n, add_noise=True)
    np.random.seed(1)
    ter = Terrain(cfg.terrain, n)
    hf, to = ter.heightsamples, ter.env_origins.astype(np.float32)
    sim = common.HostSim(cfg, n, hf=hf, terrain_origins=to, actuator=act_params(), seed=9)
    rs = np.random.RandomState(9)
    sim.terrain_level[:] = 5 + np.arange(n) % 5
    sim.terrain_type[:] = 4 + (np.arange(n) // 5) % 12          # stairs: choice in [0.2, 0.8) (legged_gym/utils/terrain.py:109-145)
    sim.f("env_origin", 3)[:] = to[sim.terrain_level, sim.terrain_type]
    root = sim.f("root", 13)
    root[:, :3] = sim.f("env_origin", 3)
    r, th = rs.uniform(1.9, 3.4, n), rs.uniform(0, 2 * np.pi, n)
    root[:, 0] += r * np.cos(th); root[:, 1] += r * np.sin(th)
    hs, border = cfg.terrain.horizontal_scale, cfg.terrain.border_size
    ix = np.clip(((root[:, 0] + border) / hs).astype(int), 4, hf.shape[0] - 5)
    iy = np.clip(((root[:, 1] + border) / hs).astype(int), 4, hf.shape[1] - 5)
    zmax = np.max(np.stack([hf[ix + a, iy + b] for a in range(-4, 5) for b in range(-4, 5)]), axis=0) * cfg.terrain.vertical_scale
    root[:, 2] = zmax + 0.60
    root[:, 3:7] = synth.rand_unit_quat(rs, n, 0.05)
    sim.f("dof_pos", 12)[:] = np.array([sim.c.default_dof_pos[i] for i in range(12)]) + rs.uniform(-0.05, 0.05, (n, 12))
    sim.f("friction")[:] = rs.uniform(0.5, 1.25, n)
    sim.f("added_mass")[:] = rs.uniform(-5, 5, n)
    sim.f("commands", 4)[:] = rs.uniform(-1, 1, (n, 4))
    sim.episode_length[:] = rs.choice([3, 200, 497, 498, 499, 700], n)
    rr = _anymal_full_step_parity("test_hip_anymal_on_curriculum_stairs", sim, rs, n, 24, count_outliers=True)
    assert rr["face_contacts"] > 0, "no quadruped touched a riser"
    assert rr["many_contacts"] > 0, "no env-step with more than 6 contacts: the raised cap is not exercised"
    # measured: 1 of 3840 env-steps (one leg sliding along a riser at 6.6 rad/s: 1.6e-4 on one joint velocity term, every recorded
    # decision equal) -- a friction-cone branch; bounded at 0.2 %
    assert rr["outliers"] <= 0.002 * 24 * n
    assert rr["loose"] <= 0.004 * rr["checks"]
    assert rr["flipped"] <= 0.01 * 24 * n
    assert rr["scan_flips"] <= 12


def test_hip_anymal_rough_8192_properties():
    """Config 5 at its stated size (8192 envs per GPU): determinism, bounded obs, reset bookkeeping, finite state."""
    n = 8192
    sim, rs = _anymal_rough_sim(n, seed=8)
    sim2 = sim.clone()
    hip, hip2 = common.HipMirror(sim), common.HipMirror(sim2)
    t0 = sim.c.push_interval - 10
    resets = 0
    for t in range(60):
        act = rs.normal(0, 0.6, (n, 12)).astype(np.float32)
        hip.step(act, t0 + t)
        if t < 12:
            hip2.step(act, t0 + t)
            np.testing.assert_array_equal(sim.obs, sim2.obs)
            np.testing.assert_array_equal(sim.state, sim2.state)
        assert np.isfinite(sim.obs).all() and np.isfinite(sim.rew).all()
        assert np.abs(sim.obs).max() <= 100.0 and (sim.rew >= 0).all()          # only_positive_rewards
        assert (sim.episode_length[sim.reset != 0] == 0).all()
        assert (sim.episode_length[sim.reset == 0] > 0).all()
        assert int(sim.stats[common._abi.PF_MAX_REWARDS]) == int(sim.reset.sum())
        assert (sim.terrain_level >= 0).all() and (sim.terrain_level < 10).all()
        L = sim.L
        fresh = sim.reset != 0
        assert (sim.state[fresh][:, L.lstm_state:L.lstm_state + 384] == 0).all()   # actuator-net state cleared on reset
        resets += int(fresh.sum())
    assert resets > 0
    assert np.isfinite(sim.state).all()
    hip.close(); hip2.close()


@pytest.mark.gpu
def test_calibration_hooks_read_what_the_host_says():
    """include/pointfoot_hip_dev.h: pf_debug_calib_gather / pf_debug_calib_read, the FETCH_SIZE calibration kernels behind
    tools/calib_gather.py and tools/calib_fetch.py.  Both keep their loads alive by comparing a sum with a value ordinary data cannot
    produce and writing the sink only on a hit: the gather hook must run over in-range cells and leave the sink alone, the read hook
    must hit when one lane's sum IS the sentinel (so it reads the floats the byte count assumes)."""
    import torch
    lib = common._abi.load()
    lib.pf_debug_calib_gather.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    lib.pf_debug_calib_read.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    field = torch.full((64, 64), 1, dtype=torch.int16, device="cuda:0")
    base = torch.tensor([[5, 7], [40, 50]], dtype=torch.int32, device="cuda:0")      # 50 + 11 < 64: inside the field
    sink = torch.zeros(4, device="cuda:0")
    assert lib.pf_debug_calib_gather(field.data_ptr(), 64, base.data_ptr(), 2, sink.data_ptr(), s) == 0, lib.pf_last_error()
    torch.cuda.synchronize()
    assert float(sink.abs().sum()) == 0.0
    src = torch.zeros(64, 160, device="cuda:0")
    src[3, 0] = 1234.5678          # lane 0 of row 3 sums to exactly the sentinel: the sink is written
    assert lib.pf_debug_calib_read(src.data_ptr(), 64, 160, sink.data_ptr(), s) == 0, lib.pf_last_error()
    torch.cuda.synchronize()
    assert abs(float(sink[0]) - 1234.5678) < 1e-3
