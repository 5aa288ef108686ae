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
