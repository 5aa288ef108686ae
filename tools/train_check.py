"""Functional check: does the loop learn?  Trains pointfoot_flat / pointfoot_rough for a few hundred iterations and prints
mean episode length / reward every 25 iterations -- and how often the contact solver's cap bites: the fraction of (env,
substep) pairs with more active contact candidates than the PF_MAXC the solve keeps (stats slots PF_STAT_CAP_*).
   python tools/train_check.py TASK [ITERS] [NUM_ENVS]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointfoot_amd.envs import task_registry
from pointfoot_amd.utils import get_args
task = sys.argv[1] if len(sys.argv) > 1 else "pointfoot_flat"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 300
nenv = sys.argv[3] if len(sys.argv) > 3 else "4096"
args = get_args(["--task", task, "--headless", "--num_envs", nenv])
from pointfoot_amd import _abi
env, _ = task_registry.make_env(task, args=args)
runner, _ = task_registry.make_alg_runner(env, task, args=args, log_root=None)
runner.env.episode_length_buf = torch.randint_like(runner.env.episode_length_buf, high=int(runner.env.max_episode_length))
obs, cobs = env.get_observations(), env.get_privileged_observations()
cobs = cobs if cobs is not None else obs
t0 = time.time()
cap_hits = cap_drop = 0.0
T, dec = runner.num_steps_per_env, env.cfg.control.decimation
cap_max_frac = 0.0
for it in range(iters):
    obs, cobs, _ = runner.rollout(obs, cobs)
    ring = env._stats_ring          # one slot per step of the iteration just collected (others are zero)
    h, d = float(ring[:, _abi.PF_STAT_CAP_HITS].sum()), float(ring[:, _abi.PF_STAT_CAP_DROPPED].sum())
    ring[:, _abi.PF_STAT_CAP_HITS:_abi.PF_STAT_CAP_DROPPED + 1] = 0
    cap_hits += h; cap_drop += d
    cap_max_frac = max(cap_max_frac, h / (env.num_envs * T * dec))
    vl, sl = runner.alg.update()
    if it % 25 == 0 or it == iters - 1:
        st = runner.alg.ep_stats.tolist(); runner.alg.ep_stats.zero_()
        n = max(st[2], 1)
        lvl = float(env.terrain_levels.float().mean()) if "rough" in task else 0.0
        print(f"it {it:4d}  episodes {int(st[2]):6d}  mean_len {st[1]/n:7.1f}  mean_ret {st[0]/n:8.2f}  vloss {float(vl):8.3f}  lr {float(runner.alg.current_learning_rate):.2e}  "
              f"std {float(runner.alg.actor_critic.std.detach().mean()):.3f}  terrain_lvl {lvl:.2f}  t {time.time()-t0:5.1f}s  "
              f"cap_hit_frac {cap_hits / ((it + 1) * env.num_envs * T * dec):.2e}", flush=True)
tot = iters * env.num_envs * T * dec
print(f"contact cap ({task}, {env.num_envs} envs, {iters} iterations): {int(cap_hits)} of {tot} (env, substep) pairs had more than PF_MAXC active candidates "
      f"= {cap_hits / tot:.3e} (worst iteration {cap_max_frac:.3e}); {int(cap_drop)} candidates dropped = {cap_drop / max(cap_hits, 1):.2f} per hit")
