"""python -m pointfoot_amd.scripts.play --task=pointfoot_flat [--load_run X --checkpoint K]
(mirror of reference legged_gym/scripts/play.py:42-111, headless: no Logger plots / camera / frame recording)."""
import os

import torch

from pointfoot_amd import LEGGED_GYM_ROOT_DIR
from pointfoot_amd.envs import *  # noqa: F401,F403
from pointfoot_amd.utils import get_args, export_policy_as_jit, task_registry

EXPORT_POLICY = True


def play(args, num_steps=None):
    env_cfg, train_cfg = task_registry.get_cfgs(name=args.task)
    # override some parameters for testing (play.py:45-51)
    env_cfg.env.num_envs = min(env_cfg.env.num_envs, 50)
    env_cfg.terrain.num_rows = 5
    env_cfg.terrain.num_cols = 5
    env_cfg.terrain.curriculum = False
    env_cfg.noise.add_noise = False
    env_cfg.domain_rand.randomize_friction = False
    env_cfg.domain_rand.push_robots = False
    env, _ = task_registry.make_env(name=args.task, args=args, env_cfg=env_cfg)
    obs = env.get_observations()
    train_cfg.runner.resume = True
    ppo_runner, train_cfg = task_registry.make_alg_runner(env=env, name=args.task, args=args, train_cfg=train_cfg)
    policy = ppo_runner.get_inference_policy(device=env.device)
    if EXPORT_POLICY:
        path = os.path.join(LEGGED_GYM_ROOT_DIR, 'logs', train_cfg.runner.experiment_name, 'exported', 'policies')
        export_policy_as_jit(ppo_runner.alg.actor_critic, path)
        print('Exported policy as jit script to: ', path)
    steps = num_steps if num_steps is not None else 10 * int(env.max_episode_length)
    rew_sum, n_done = torch.zeros((), device=env.device), torch.zeros((), device=env.device)
    for i in range(steps):
        actions = policy(obs.detach())
        obs, _, rews, dones, infos = env.step(actions.detach())
        rew_sum += rews.sum()
        n_done += dones.sum()
    print(f"played {steps} steps x {env.num_envs} envs: mean reward/step {float(rew_sum) / (steps * env.num_envs):.4f}, episodes finished {int(n_done)}")
    return env


if __name__ == '__main__':
    play(get_args())
