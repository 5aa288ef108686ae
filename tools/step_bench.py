"""Micro-benchmark of the fused env-step kernel alone (pointfoot_rough, N envs, random actions):
   python tools/step_bench.py [N] [launches] [task]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointfoot_amd import _abi
if os.environ.get("PF_LIB_VARIANT"):      # a variant build of the library (tools/build_variant.sh)
    _abi.lib_path = lambda: os.path.abspath(os.environ["PF_LIB_VARIANT"])
from pointfoot_amd.envs import task_registry
from pointfoot_amd.utils import get_args

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 200
task = sys.argv[3] if len(sys.argv) > 3 else "pointfoot_rough"
args = get_args(["--task", task, "--headless", "--num_envs", str(N)])
env, _ = task_registry.make_env(task, args=args)
env.reset()
g = torch.Generator(device="cuda:0").manual_seed(0)
acts = [torch.randn(N, env.num_actions, device="cuda:0", generator=g) * 0.5 for _ in range(8)]
for i in range(30):
    env.step(acts[i % 8])
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter(); s.record()
for i in range(K):
    env.step(acts[i % 8])
e.record(); torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / K * 1e6
print(f"N={N} {task}: {s.elapsed_time(e) / K * 1e3:.1f} us/launch (event), {wall:.1f} us wall; "
      f"{N / (s.elapsed_time(e) / K * 1e-3) / 1e6:.2f} M env-steps/s kernel-only; resets/step ~{float(env.reset_buf.float().mean()) * N:.0f}")
