"""Diagnostic: where does the HIP step kernel differ from the float32 oracle by more than 5e-5 on envs whose discrete
decisions agree, and how does that compare with the oracle's own float32-vs-float64 gap for the same env (a measure of
how ill-conditioned that env's contact problem is)?   python tools/parity_probe.py [rough|anymal]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np
import common
import test_gpu_env as T

which = sys.argv[1] if len(sys.argv) > 1 else "rough"
if which == "rough":
    n, steps, na, c0 = 256, 30, 6, 1
    sim, rs = T._physics_sim("rough", n)
else:
    n, steps, na = 128, 16, 12
    sim, rs = T._anymal_rough_sim(n)
    c0 = sim.c.push_interval - 4
ref, ref64 = sim.clone(), sim.clone()
hip = common.HipMirror(sim)
nprop = sim.c.num_obs if which == "rough" else 48
tot = out = 0
for t in range(steps):
    act = rs.normal(0, 0.7 if which == "rough" else 0.4, (n, na)).astype(np.float32)
    for k in common.HipMirror.FIELDS:
        getattr(sim, k)[...] = getattr(ref, k); getattr(ref64, k)[...] = getattr(ref, k)
    hip.upload()
    ref.step("oracle", act, c0 + t); ref64.step("oracle", act, c0 + t, physics_bits=64)
    hip.step(act, c0 + t)
    m = T._same_decisions(sim, ref)
    m64 = T._same_decisions(ref64, ref)
    e = np.abs(sim.obs[:, :nprop] - ref.obs[:, :nprop]).max(axis=1)
    g = np.abs(ref64.obs[:, :nprop] - ref.obs[:, :nprop]).max(axis=1)
    tot += int(m.sum())
    for i in np.where(m & (e > 5e-5))[0]:
        out += 1
        j = int(np.abs(sim.obs[i, :nprop] - ref.obs[i, :nprop]).argmax())
        print(f"step {t:2d} env {i:3d}: |hip-o32| {e[i]:.2e} at obs[{j}]  |o64-o32| {g[i]:.2e}  o64 same decisions: {bool(m64[i])}  contacts "
              + " ".join(hex(int(v)) for v in ref.contact_set[i, :, 0]) + "  limits " + " ".join(hex(int(v)) for v in ref.contact_set[i, :, 1]))
    print(f"step {t:2d}: matched {int(m.sum())}/{n}  p50 {np.median(e[m]):.1e} p99 {np.percentile(e[m], 99):.1e} max {e[m].max():.1e} | oracle f64 gap p50 {np.median(g[m & m64]):.1e} p99 {np.percentile(g[m & m64], 99):.1e}")
print(f"outliers (> 5e-5) {out} of {tot} matched env-steps")
