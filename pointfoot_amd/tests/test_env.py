"""Mirror of the reference's only test (legged_gym/tests/test_env.py:42-52): build the env through the registry with
at most 10 envs and step it with zero actions; here with assertions."""
import torch

from pointfoot_amd.envs import *  # noqa: F401,F403
from pointfoot_amd.utils import get_args, task_registry


def test_env(args, steps=None):
    env_cfg, train_cfg = task_registry.get_cfgs(name=args.task)
    env_cfg.env.num_envs = min(env_cfg.env.num_envs, 10)
    env, _ = task_registry.make_env(name=args.task, args=args, env_cfg=env_cfg)
    n = steps if steps is not None else int(10 * env.max_episode_length)
    for i in range(n):
        actions = 0. * torch.ones(env.num_envs, env.num_actions, device=env.device)
        obs, _, rew, done, info = env.step(actions)
    torch.cuda.synchronize()
    assert torch.isfinite(obs).all() and torch.isfinite(rew).all()
    print("Done")
    return env


if __name__ == '__main__':
    test_env(get_args())
