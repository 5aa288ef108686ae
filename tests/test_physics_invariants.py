"""The physics substep has no reference artefact (Isaac Gym/PhysX is a closed, absent binary): parity
unpinned.  These tests pin the oracle's own articulated-body step to physics instead — momentum and
energy in free flight, free-fall acceleration, static support force — and check that the host build of the
HIP step program agrees with the oracle on full steps (same algorithm, float32)."""
import ctypes as C

import numpy as np
import pytest

import common


def energy(sim, i):
    out = (C.c_double * 5)()
    common.oracle_lib().pfo_energy(C.byref(sim.model), C.byref(sim.c), C.byref(sim.L), sim.state[i].ctypes.data, out)
    return np.array(list(out))


def flying_sim(n=8, gravity=True, seed=0):
    cfg = common.make_cfg("flat", n, control_type="T", add_noise=False)
    cfg.domain_rand.push_robots = False
    if not gravity:
        cfg.sim.gravity = [0., 0., 0.]
    sim = common.HostSim(cfg, n)
    rs = np.random.RandomState(seed)
    import synth
    root = sim.f("root", 13)
    root[:, 0:2] = rs.uniform(-3, 3, (n, 2))
    root[:, 2] = 50.0
    root[:, 3:7] = synth.rand_unit_quat(rs, n, 0.8)
    root[:, 7:10] = rs.normal(0, 1.0, (n, 3))
    root[:, 10:13] = rs.normal(0, 1.5, (n, 3))
    sim.f("dof_pos", 6)[:] = rs.uniform(-0.2, 0.2, (n, 6))
    sim.f("dof_vel", 6)[:] = rs.normal(0, 1.0, (n, 6))
    return sim


@pytest.mark.parametrize("bits", [32, 64])
def test_free_flight_conserves_momentum_and_energy(bits):
    sim = flying_sim(gravity=False)
    n = sim.n
    e0 = np.stack([energy(sim, i) for i in range(n)])
    for t in range(10):       # 40 substeps of 5 ms
        sim.step("oracle", np.zeros((n, 6), np.float32), t + 1, physics_bits=bits)
        assert not sim.reset.any()
    e1 = np.stack([energy(sim, i) for i in range(n)])
    tol = 2e-3 if bits == 32 else 1e-3     # semi-implicit Euler drift over 0.2 s + float32 state records
    np.testing.assert_allclose(e1[:, 2:5], e0[:, 2:5], atol=tol * 20)        # linear momentum (kg m/s)
    np.testing.assert_allclose(e1[:, 0], e0[:, 0], rtol=5e-3)                # kinetic energy


def test_free_fall_acceleration():
    sim = flying_sim(gravity=True)
    n = sim.n
    e0 = np.stack([energy(sim, i) for i in range(n)])
    steps = 5
    for t in range(steps):
        sim.step("oracle", np.zeros((n, 6), np.float32), t + 1, physics_bits=64)
    e1 = np.stack([energy(sim, i) for i in range(n)])
    total_mass = sum(sim.model.mass[i] for i in range(sim.model.nb))
    T = steps * sim.c.dt
    np.testing.assert_allclose(e1[:, 4] - e0[:, 4], -9.81 * total_mass * T, rtol=2e-3)
    np.testing.assert_allclose(e1[:, 2:4], e0[:, 2:4], atol=2e-2)
    # total mechanical energy (KE + PE) conserved
    np.testing.assert_allclose(e1[:, 0] + e1[:, 1], e0[:, 0] + e0[:, 1], rtol=2e-3)


def standing_sim(n=4, task="flat"):
    cfg = common.make_cfg(task, n, add_noise=False)
    cfg.domain_rand.push_robots = False
    sim = common.HostSim(cfg, n)
    sim.f("root", 13)[:, 2] = 0.66
    sim.f("friction")[:] = 1.0
    return sim


def test_standing_support_force_and_no_tunnelling():
    sim = standing_sim()
    n = sim.n
    total_mass = sum(sim.model.mass[i] for i in range(sim.model.nb))
    fz = []
    for t in range(25):
        sim.step("oracle", np.zeros((n, 6), np.float32), t + 1)
        cf = sim.f("contact_forces", 27).reshape(n, 9, 3)
        fz.append(cf[:, :, 2].sum(axis=1))
        feet_z = sim.f("feet_pos", 6).reshape(n, 2, 3)[:, :, 2]
        assert (feet_z > 0.03 - 0.02).all(), "foot sphere sank through the plane"
    fz = np.array(fz)[5:]
    # averaged over the settle the ground carries the weight (within 25 %: the robot is sagging on soft PD gains)
    assert abs(fz.mean() / (total_mass * 9.81) - 1.0) < 0.25


def test_emu_matches_oracle_full_physics_steps():
    """Same algorithm, float32 both sides: the HIP program (host build) vs the oracle over 20 policy steps."""
    a = standing_sim(8)
    rs = np.random.RandomState(3)
    a.f("dof_pos", 6)[:] = rs.uniform(-0.1, 0.1, (8, 6))
    b = a.clone()
    for t in range(20):
        act = rs.normal(0, 0.5, (8, 6)).astype(np.float32)
        a.step("oracle", act, t + 1)
        b.step("emu", act, t + 1)
        np.testing.assert_array_equal(a.reset, b.reset)
        np.testing.assert_allclose(b.obs, a.obs, atol=2e-3, rtol=0)
        np.testing.assert_allclose(b.rew, a.rew, atol=2e-3, rtol=0)
    # single-step agreement from identical state is tight
    b.state[:] = a.state
    b.episode_length[:] = a.episode_length
    act = rs.normal(0, 0.5, (8, 6)).astype(np.float32)
    a.step("oracle", act, 100)
    b.step("emu", act, 100)
    np.testing.assert_allclose(b.obs, a.obs, atol=1e-4, rtol=0)
    np.testing.assert_allclose(b.rew, a.rew, atol=1e-4, rtol=0)
    np.testing.assert_allclose(b.state, a.state, atol=2e-4, rtol=1e-4)


# ------------------------------------------------------------------------------- collision geometry (a3 contract)
# Vertical faces (legged_gym/utils/terrain.py:69-73, slope_treshold) and leg-leg self-collision
# (pointfoot_rough_config.py:107) are part of the reference's physics contract.  Own model, parity unpinned; pinned
# here by what the geometry has to do.
def riser_sim(n=2, trimesh=True, height_raw=40):
    """Frictionless ground at z = 0 for x < 5 m, a 0.2 m step up at x = 5 m; robots slide towards it at 1 m/s."""
    import synth
    cfg = common.make_cfg("rough", n, add_noise=False)
    cfg.domain_rand.push_robots = False
    cfg.terrain.static_friction = 0.0
    if not trimesh:
        cfg.terrain.mesh_type = "heightfield"      # no slope_treshold correction: the step is a 0.1 m wide ramp
    hf = np.zeros((synth.TOT_ROWS, synth.TOT_COLS), np.int16)
    hf[300:, :] = height_raw                        # grid line 300 = x 5.0 m
    to = synth.synth_terrain_origins().astype(np.float32)
    to[:, :, 2] = 0
    sim = common.HostSim(cfg, n, hf=hf, terrain_origins=to)
    root = sim.f("root", 13)
    root[:, 0] = 4.80
    root[:, 1] = np.linspace(3.0, 3.5, n)
    root[:, 2] = 0.66
    root[:, 7] = 1.0
    sim.f("friction")[:] = 0.0
    sim.f("env_origin", 3)[:] = root[:, :3] * [1, 1, 0]
    return sim


@pytest.mark.parametrize("backend", ["oracle", "emu"])
def test_foot_pushed_against_a_riser_does_not_climb(backend):
    sim = riser_sim()
    n, nsph = sim.n, sim.model.nsph
    assert sim.c.wall_threshold == pytest.approx(0.075)
    touched = False
    for t in range(14):        # until the body starts to topple over the blocked feet
        sim.step(backend, np.zeros((n, 6), np.float32), t + 1)
        fp = sim.f("feet_pos", 6).reshape(n, 2, 3)
        assert (fp[:, :, 0] <= 5.0 - 0.03 + 0.012).all(), f"step {t}: a foot entered the riser: {fp[:, :, 0]}"      # 1 m/s impact: < 1 substep of travel
        assert t < 2 or (fp[:, :, 2] <= 0.03 + 0.004).all(), f"step {t}: a foot climbed: {fp[:, :, 2]}"      # (the robot is set down during the first two steps)
        face_bits = (sim.contact_set[:, -1, 0] >> np.uint64(nsph)) & np.uint64(3)      # face candidates of the two feet
        touched = touched or bool((face_bits == 3).all())
        assert not sim.reset.any()
    assert touched, "the feet never reached the vertical face"
    # the same field as a plain bilinear height field (mesh_type 'heightfield'): the feet stop at the foot of the ramp
    ramp = riser_sim(trimesh=False)
    assert ramp.c.wall_threshold == 0.0
    for t in range(10):
        ramp.step(backend, np.zeros((n, 6), np.float32), t + 1)
    assert (ramp.f("feet_pos", 6).reshape(n, 2, 3)[:, :, 0] < 4.92).all()


@pytest.mark.parametrize("backend", ["oracle", "emu"])
def test_ground_next_to_a_riser_stays_low(backend):
    """A robot set down inside the steep grid cell stands on the low ground (the trimesh's stretched low cell), not half
    way up a ramp."""
    sim = riser_sim()
    n = sim.n
    root = sim.f("root", 13)
    root[:, 7] = 0.0
    root[:, 0] = 4.95 + 0.065     # the feet hang 6.5 cm behind the base origin: they come down at x ~ 4.95, inside grid cell 299
    sim.f("friction")[:] = 2.0     # mu = 0.5 * (2 + 0) = 1: no sliding
    for t in range(8):
        sim.step(backend, np.zeros((n, 6), np.float32), t + 1)
    fp = sim.f("feet_pos", 6).reshape(n, 2, 3)
    assert ((fp[:, :, 0] > 4.9) & (fp[:, :, 0] < 5.0 - 0.03 + 0.004)).all(), fp[:, :, 0]
    np.testing.assert_allclose(fp[:, :, 2], 0.03, atol=0.004)


def crossing_sim(self_collisions, n=2):
    """Zero gravity, torque control: both abad joints are driven inwards with 20 N m."""
    cfg = common.make_cfg("flat", n, control_type="T", add_noise=False)
    cfg.domain_rand.push_robots = False
    cfg.sim.gravity = [0., 0., 0.]
    cfg.asset.self_collisions = 0 if self_collisions else 1
    sim = common.HostSim(cfg, n)
    sim.f("root", 13)[:, 2] = 50.0
    act = np.zeros((n, 6), np.float32)
    act[:, 0], act[:, 3] = -40.0, 40.0
    return sim, act


@pytest.mark.parametrize("backend", ["oracle", "emu"])
def test_crossed_legs_are_pushed_apart(backend):
    sim, act = crossing_sim(True)
    n = sim.n
    assert sim.model.npair == 16 and sim.c.self_collisions == 1
    e0 = np.stack([energy(sim, i) for i in range(n)])
    pressed = 0
    for t in range(16):
        sim.step(backend, act, t + 1)
        fp = sim.f("feet_pos", 6).reshape(n, 2, 3)
        # left foot stays on the left: the two 3 cm foot spheres may touch (centres 6 cm apart) but not pass
        assert (fp[:, 0, 1] - fp[:, 1, 1] >= 0.06 - 0.004).all(), f"step {t}: feet interpenetrate {fp[:, :, 1]}"
        cf = sim.f("contact_forces", 27).reshape(n, 9, 3)
        pair_bits = sim.contact_set[:, -1, 0] >> np.uint64(2 * sim.model.nsph)
        if (pair_bits != 0).all() and (np.abs(cf).sum(axis=(1, 2)) > 1.0).all():
            pressed += 1
            # action = reaction: the net contact forces of the two legs cancel
            left, right = cf[:, 1:5].sum(axis=1), cf[:, 5:9].sum(axis=1)
            np.testing.assert_allclose(left, -right, atol=1e-3 * np.abs(left).max() + 1e-4)
            assert (left[:, 1] > 0).all(), "the left leg must be pushed to the left (+y)"
        assert not np.isnan(sim.state).any()
    assert pressed >= 4
    e1 = np.stack([energy(sim, i) for i in range(n)])
    np.testing.assert_allclose(e1[:, 2:5], e0[:, 2:5], atol=0.15)      # internal forces: linear momentum stays (integrator drift only)
    # without self-collision the same drive swings the legs through each other
    free, act = crossing_sim(False)
    assert free.model.npair == 0 and free.c.self_collisions == 0
    for t in range(8):
        free.step(backend, act, t + 1)
    fp = free.f("feet_pos", 6).reshape(n, 2, 3)
    assert (fp[:, 0, 1] < fp[:, 1, 1]).all()


def test_emu_matches_oracle_on_faces_and_self_contacts():
    """HIP step program (host build) vs oracle on the two scenarios above: same contact sets, same motion."""
    for make, acts in ((lambda: (riser_sim(4), None), 14), (lambda: crossing_sim(True, 4), 16)):
        a, act = make()
        b = a.clone()
        if act is None:
            act = np.zeros((a.n, 6), np.float32)
        for t in range(acts):
            a.step("oracle", act, t + 1)
            b.step("emu", act, t + 1)
            np.testing.assert_array_equal(a.contact_set, b.contact_set)
            np.testing.assert_allclose(b.state, a.state, atol=5e-3, rtol=1e-3)
            b.state[:] = a.state


def quad_crossing_sim(self_collisions, n=2):
    """ANYmal-C in zero gravity, torque control: all four hip-abduction joints are driven inwards (the quadruped instance with the 12
    capsules / 54 pairs of asset.self_collisions = 0, anymal_c_flat_config.py:42)."""
    cfg = common.make_cfg("anymal_c_flat", n, control_type="T", add_noise=False)
    cfg.domain_rand.push_robots = False
    cfg.sim.gravity = [0., 0., 0.]
    cfg.asset.self_collisions = 0 if self_collisions else 1
    sim = common.HostSim(cfg, n)
    sim.c.use_actuator_net = 0
    sim.f("root", 13)[:, 2] = 50.0
    act = np.zeros((n, 12), np.float32)
    act[:, 0], act[:, 3], act[:, 6], act[:, 9] = -40.0, 40.0, -40.0, 40.0      # LF / RF / LH / RH hip abduction
    return sim, act


@pytest.mark.parametrize("backend", ["oracle", "emu"])
def test_quadruped_legs_do_not_pass_through_each_other(backend):
    sim, act = quad_crossing_sim(True)
    n, nsph = sim.n, sim.model.nsph
    assert sim.model.ncap == 12 and sim.model.npair == 54 and sim.c.self_collisions == 1
    assert 2 * nsph + sim.model.npair > 64      # the candidate set needs its second word
    pressed = 0
    for t in range(16):
        sim.step(backend, act, t + 1)
        fp = sim.f("feet_pos", 12).reshape(n, 4, 3)
        # left feet stay on the left of the right feet (front pair, hind pair): 3 cm foot spheres may touch, not pass
        assert (fp[:, 0, 1] - fp[:, 1, 1] >= 0.06 - 0.004).all() and (fp[:, 2, 1] - fp[:, 3, 1] >= 0.06 - 0.004).all(), f"step {t}: {fp[:, :, 1]}"
        lo, hi = sim.contact_set[:, -1, 0] >> np.uint64(2 * nsph), sim.contact_set[:, -1, 3]
        cf = sim.f("contact_forces", 51).reshape(n, 17, 3)
        if ((lo != 0) & (hi != 0)).all() and (np.abs(cf).sum(axis=(1, 2)) > 1.0).all():
            pressed += 1
            # action = reaction inside each pair of legs: LF (bodies 1-4) against RF (5-8), LH (9-12) against RH (13-16)
            for a, b in ((1, 5), (9, 13)):
                left, right = cf[:, a:a + 4].sum(axis=1), cf[:, b:b + 4].sum(axis=1)
                np.testing.assert_allclose(left, -right, atol=1e-3 * np.abs(left).max() + 1e-4)
                assert (left[:, 1] > 0).all(), "a left leg must be pushed to the left (+y)"
        assert not np.isnan(sim.state).any()
    assert pressed >= 4
    # built without: the same drive swings the legs through each other
    free, act = quad_crossing_sim(False)
    assert free.model.npair == 0 and free.c.self_collisions == 0
    for t in range(8):
        free.step(backend, act, t + 1)
    fp = free.f("feet_pos", 12).reshape(n, 4, 3)
    assert (fp[:, 0, 1] < fp[:, 1, 1]).all() and (fp[:, 2, 1] < fp[:, 3, 1]).all()


def test_emu_matches_oracle_on_quadruped_self_contacts():
    a, act = quad_crossing_sim(True, 4)
    b = a.clone()
    for t in range(14):
        a.step("oracle", act, t + 1)
        b.step("emu", act, t + 1)
        np.testing.assert_array_equal(a.contact_set, b.contact_set)
        np.testing.assert_allclose(b.state, a.state, atol=5e-3, rtol=1e-3)
        b.state[:] = a.state


@pytest.mark.parametrize("task", ["flat", "anymal_c_flat"])
def test_lumped_model_is_the_same_rigid_body_system(task):
    """pf_lump.h (what pf_create hands the step kernel): bodies behind fixed joints -- the feet -- are merged into their parents.  For random
    joint angles the lumped tree must put every collision sphere and every foot origin where the full tree puts them, and carry the same total
    mass, centre of mass and rotational inertia about the base origin (computed here in float64 from both models)."""
    import ctypes as C
    cfg = common.make_cfg(task, 1)
    sim = common.HostSim(cfg, 1)
    full = sim.model
    lumped = type(full)()
    class Lump(C.Structure):
        _fields_ = [("sph_rep", C.c_int32 * common._abi.PF_MAX_SPHERES), ("feet_dyn", C.c_int32 * common._abi.PF_MAX_FEET),
                    ("feet_off", (C.c_float * 3) * common._abi.PF_MAX_FEET)]
    lump = Lump()
    common.emu_lib().pfe_lump(C.byref(full), C.byref(lumped), C.byref(lump))
    nfixed = sum(1 for i in range(1, full.nb) if full.dof_of_body[i] < 0)
    assert nfixed == full.nfeet and lumped.nb == full.nb - nfixed and lumped.nd == full.nd

    def rodrigues(a, th):
        a = np.asarray(a, np.float64); K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
        return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)

    def fk(m, q):      # world frame of every body (base at the origin, identity)
        R, r = [np.eye(3)], [np.zeros(3)]
        for i in range(1, m.nb):
            p = m.parent[i]
            Rj = R[p] @ np.array(m.joint_rot[i], np.float64).reshape(3, 3)
            k = m.dof_of_body[i]
            if k >= 0:
                Rj = Rj @ rodrigues(m.joint_axis[i], q[k])
            R.append(Rj); r.append(r[p] + R[p] @ np.array(m.joint_pos[i], np.float64))
        return R, r

    def mass_props(m, R, r):      # total mass, first moment, inertia about the origin
        M, h, I = 0.0, np.zeros(3), np.zeros((3, 3))
        for i in range(m.nb):
            mi = float(m.mass[i]); c = r[i] + R[i] @ np.array(m.com[i], np.float64)
            a = m.inertia[i]; Il = np.array([[a[0], a[3], a[4]], [a[3], a[1], a[5]], [a[4], a[5], a[2]]], np.float64)
            M += mi; h += mi * c; I += R[i] @ Il @ R[i].T + mi * (c @ c * np.eye(3) - np.outer(c, c))
        return M, h, I

    rs = np.random.RandomState(3)
    for _ in range(5):
        q = rs.uniform(-1.0, 1.0, full.nd)
        Rf, rf = fk(full, q); Rl, rl = fk(lumped, q)
        for s in range(full.nsph):
            pf_ = rf[full.sph_body[s]] + Rf[full.sph_body[s]] @ np.array(full.sph_pos[s], np.float64)
            pl_ = rl[lumped.sph_body[s]] + Rl[lumped.sph_body[s]] @ np.array(lumped.sph_pos[s], np.float64)
            np.testing.assert_allclose(pl_, pf_, atol=2e-6)
            assert lump.sph_rep[s] == full.sph_body[s]
        for f in range(full.nfeet):
            b = lump.feet_dyn[f]
            np.testing.assert_allclose(rl[b] + Rl[b] @ np.array(lump.feet_off[f], np.float64), rf[full.feet_body[f]], atol=2e-6)
        Mf, hf, If = mass_props(full, Rf, rf); Ml, hl, Il = mass_props(lumped, Rl, rl)
        assert abs(Mf - Ml) < 1e-5 * Mf
        np.testing.assert_allclose(hl, hf, atol=1e-5 * Mf)
        np.testing.assert_allclose(Il, If, atol=2e-5 * np.abs(If).max())
