"""Diagnostic: where one policy step of pf_step_kernel spends its cycles (s_memtime stamps, lane 0 of every wave).
Builds a separate libpointfoot_hip_stamps.so with -DPF_STAMPS (the product library contains no stamp), runs
`python tools/step_stamps.py [N] [task]` (default pointfoot_rough, N=4096) and prints the mean cycles per phase.  Read SHARES, not absolute times."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib_dbg = os.path.join(ROOT, "gpurun_out", "libpointfoot_hip_stamps.so")
os.makedirs(os.path.dirname(lib_dbg), exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-slp-vectorize", "-mllvm", "-inline-threshold=225",
                       "-fPIC", "-shared", "-DPF_STAMPS", *os.environ.get("PF_EXTRA_FLAGS", "").split(), "-o", lib_dbg,
                       os.path.join(ROOT, "pointfoot_amd/csrc/pf_env.hip"), os.path.join(ROOT, "pointfoot_amd/csrc/pf_env_quad.hip"),
                       os.path.join(ROOT, "pointfoot_amd/csrc/pf_ppo.hip")])
from pointfoot_amd import _abi
_abi.lib_path = lambda: lib_dbg
import torch
from pointfoot_amd.envs import task_registry
from pointfoot_amd.utils import get_args
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
task = sys.argv[2] if len(sys.argv) > 2 else "pointfoot_rough"
args = get_args(["--task", task, "--headless", "--num_envs", str(N)])
env, _ = task_registry.make_env(task, args=args)
env.reset()
lib = _abi.load()
stamps = torch.zeros(N, 16, dtype=torch.int32, device="cuda:0")
lib.pf_debug_set_stamps.argtypes = [C.c_void_p, C.c_void_p]
lib.pf_debug_set_stamps(env._h, stamps.data_ptr())
g = torch.Generator(device="cuda:0").manual_seed(0)
acc = torch.zeros(16, dtype=torch.float64, device="cuda:0")
K = 40
tot_hist = []           # per-wave totals of the timed steps: the launch ends with its SLOWEST wave (all waves are resident at once)
for i in range(K + 10):
    env.step(torch.randn(N, env.num_actions, device="cuda:0", generator=g) * 0.5)
    if i >= 10:
        acc += stamps.double().mean(0)
        per_wave = stamps.double().sum(1)
        per_wave = per_wave[per_wave > 0]          # lane 0 of a wave writes the row of its first robot only
        sd = stamps.double()[stamps.double().sum(1) > 0]
        top = sd[sd.sum(1) >= sd.sum(1).quantile(0.99)]
        top_acc = top.mean(0) if i == 10 else top_acc + top.mean(0)
        rs = env.reset_buf.bool()
        tot_hist.append((per_wave.mean().item(), per_wave.median().item(), per_wave.quantile(0.99).item(), per_wave.max().item(),
                         stamps.double()[:, 14][stamps.double().sum(1) > 0].max().item(), int(rs.sum())))
torch.cuda.synchronize()
names = ["loop/ext bookkeeping", "kinematics (FK+vel)", "inertia+composite+F", "mass matrix+bias", "cholesky", "contact detect",
         "contact rows", "solve rows (M^-1 J^T)", "free vel + limits", "delassus blocks", "PGS 4 iters", "integrate+forces",
         "PD torques", "final FK + feet", "post_physics (python layer)", "record store"]
m = (acc / K).cpu().numpy()
tot = m.sum()
print(f"mean cycles per wave per policy step: {tot:.0f} (s_memtime ticks)")
for i, n in enumerate(names):
    per = "x4 substeps" if i in (0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12) else ""
    print(f"  [{i:2d}] {n:30s} {m[i]:9.0f}  {100 * m[i] / tot:5.1f} %  {per}")

import numpy as np
h = np.array(tot_hist)
print(f"per-wave total ticks over {len(h)} steps: mean {h[:, 0].mean():.0f}, median {h[:, 1].mean():.0f}, p99 {h[:, 2].mean():.0f}, max {h[:, 3].mean():.0f} "
      f"(slowest wave / mean wave = {h[:, 3].mean() / h[:, 0].mean():.3f}); post_physics max {h[:, 4].mean():.0f}; resets per step {h[:, 5].mean():.0f}")

tm = (top_acc / K).cpu().numpy()
print("slowest 1 % of the waves, mean ticks per phase (vs all waves):")
for i, n in enumerate(names):
    print(f"  [{i:2d}] {n:30s} {tm[i]:9.0f}  ({2 * m[i]:9.0f})")
