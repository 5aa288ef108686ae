"""Diagnostic: where one policy step of pf_step_kernel spends its cycles (s_memtime stamps, lane 0 of every wave).
Builds a separate libpointfoot_hip_stamps.so with -DPF_STAMPS (the product library contains no stamp), runs
`python tools/step_stamps.py [N] [task]` (default pointfoot_rough, N=4096) and prints the mean cycles per phase.  Read SHARES, not absolute times."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib_dbg = os.path.join(ROOT, "gpurun_out", "libpointfoot_hip_stamps.so")
os.makedirs(os.path.dirname(lib_dbg), exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-slp-vectorize", "-mllvm", "-inline-threshold=225",
                       "-mllvm", "-amdgpu-sched-strategy=iterative-ilp", "-fPIC", "-shared", "-DPF_STAMPS", "-o", lib_dbg,
                       os.path.join(ROOT, "pointfoot_amd/csrc/pf_env.hip"), os.path.join(ROOT, "pointfoot_amd/csrc/pf_ppo.hip")])
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
for i in range(K + 10):
    env.step(torch.randn(N, env.num_actions, device="cuda:0", generator=g) * 0.5)
    if i >= 10:
        acc += stamps.double().mean(0)
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
