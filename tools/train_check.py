"""Functional check: does the loop learn?  Trains pointfoot_flat / pointfoot_rough for a few hundred iterations and prints
mean episode length / reward every 25 iterations."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointfoot_amd.envs import task_registry
from pointfoot_amd.utils import get_args
task = sys.argv[1] if len(sys.argv) > 1 else "pointfoot_flat"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 300
args = get_args(["--task", task, "--headless", "--num_envs", "4096"])
env, _ = task_registry.make_env(task, args=args)
runner, _ = task_registry.make_alg_runner(env, task, args=args, log_root=None)
runner.env.episode_length_buf = torch.randint_like(runner.env.episode_length_buf, high=int(runner.env.max_episode_length))
obs, cobs = env.get_observations(), env.get_privileged_observations()
cobs = cobs if cobs is not None else obs
t0 = time.time()
for it in range(iters):
    obs, cobs, _ = runner.rollout(obs, cobs)
    vl, sl = runner.alg.update()
    if it % 25 == 0 or it == iters - 1:
        st = runner.alg.ep_stats.tolist(); runner.alg.ep_stats.zero_()
        n = max(st[2], 1)
        lvl = float(env.terrain_levels.float().mean()) if "rough" in task else 0.0
        print(f"it {it:4d}  episodes {int(st[2]):6d}  mean_len {st[1]/n:7.1f}  mean_ret {st[0]/n:8.2f}  vloss {float(vl):8.3f}  lr {float(runner.alg.current_learning_rate):.2e}  "
              f"std {float(runner.alg.actor_critic.std.detach().mean()):.3f}  terrain_lvl {lvl:.2f}  t {time.time()-t0:5.1f}s", flush=True)
