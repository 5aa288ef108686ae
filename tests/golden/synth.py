"""Deterministic synthetic inputs shared by the golden generator and the tests.

Nothing here comes from the reference: these are seeded numpy recipes for the
"sim state" tensors that Isaac Gym would hand to the reference's Python layer
(root state, DOF state, net contact forces, rigid-body state) and for the
int16 height field.  `numpy.random.RandomState` (legacy MT19937 streams) is
bit-stable across numpy versions, so the generator (run in the build
container, where /root/reference exists) and the tests (run anywhere) see the
same inputs.
"""
import numpy as np

TOT_ROWS, TOT_COLS = 1300, 2100        # (10*80 + 2*250, 20*80 + 2*250), reference terrain.py:53-60
BORDER = 25.0
HSCALE, VSCALE = 0.1, 0.005


def synth_heightfield(seed=7):
    """Synthetic int16 (1300, 2100) height field: smooth bumps + steps + noise."""
    rs = np.random.RandomState(seed)
    x = np.arange(TOT_ROWS, dtype=np.float64)[:, None]
    y = np.arange(TOT_COLS, dtype=np.float64)[None, :]
    h = 40.0 * np.sin(x / 37.0) * np.cos(y / 53.0) + 25.0 * np.sin((x + 2 * y) / 11.0)
    h += 30.0 * ((x // 9 + y // 13) % 3)             # discrete steps
    h += rs.randint(-6, 7, size=(TOT_ROWS, TOT_COLS))
    return np.rint(h).astype(np.int16)


def synth_terrain_origins(seed=11, rows=10, cols=20):
    rs = np.random.RandomState(seed)
    o = np.zeros((rows, cols, 3), dtype=np.float64)
    for i in range(rows):
        for j in range(cols):
            o[i, j] = [(i + 0.5) * 8.0, (j + 0.5) * 8.0, rs.uniform(-0.3, 0.6)]
    return o


def rand_unit_quat(rs, n, tilt=0.5):
    """xyzw quaternions: yaw uniform, roll/pitch ~ N(0, tilt)."""
    yaw = rs.uniform(-np.pi, np.pi, n)
    roll = rs.normal(0, tilt, n)
    pitch = rs.normal(0, tilt, n)
    cy, sy = np.cos(yaw / 2), np.sin(yaw / 2)
    cr, sr = np.cos(roll / 2), np.sin(roll / 2)
    cp, sp = np.cos(pitch / 2), np.sin(pitch / 2)
    q = np.stack([sr * cp * cy - cr * sp * sy,
                  cr * sp * cy + sr * cp * sy,
                  cr * cp * sy - sr * sp * cy,
                  cr * cp * cy + sr * sp * sy], axis=1)
    return q / np.linalg.norm(q, axis=1, keepdims=True)


def synth_sim_state(rs, n, nb, nd, origins, feet_idx, rough):
    """One synthetic post-simulate snapshot.  Returns dict of float32 arrays."""
    root = np.zeros((n, 13))
    root[:, 0:2] = origins[:, 0:2] + rs.uniform(-5.0, 5.0, (n, 2))
    root[:, 2] = origins[:, 2] + rs.uniform(0.3, 0.9, n)
    root[:, 3:7] = rand_unit_quat(rs, n)
    root[:, 7:13] = rs.normal(0, 1.0, (n, 6))
    dof = np.zeros((n, nd, 2))
    dof[:, :, 0] = rs.uniform(-1.2, 1.2, (n, nd))
    dof[:, :, 1] = rs.normal(0, 4.0, (n, nd))
    # sparse contact forces: feet often loaded, other bodies rarely
    cf = np.zeros((n, nb, 3))
    for b in range(nb):
        p = 0.6 if b in feet_idx else 0.08
        on = rs.uniform(size=n) < p
        mag = rs.choice([0.05, 0.5, 3.0, 60.0, 400.0], size=n)
        cf[:, b, :] = on[:, None] * mag[:, None] * rs.normal(0, 1.0, (n, 3))
        if b in feet_idx:
            cf[:, b, 2] = np.abs(cf[:, b, 2])
    body = np.zeros((n, nb, 13))
    body[:, :, 0:3] = root[:, None, 0:3] + rs.uniform(-0.4, 0.4, (n, nb, 3))
    for f in feet_idx:
        body[:, f, 2] = origins[:, 2] + rs.uniform(0.0, 0.25, n)
    body[:, :, 3:7] = rand_unit_quat(rs, n * nb).reshape(n, nb, 4)
    body[:, :, 7:13] = rs.normal(0, 1.0, (n, nb, 6))
    f32 = lambda a: a.astype(np.float32)
    return dict(root=f32(root), dof=f32(dof), contact=f32(cf), body=f32(body))
