"""python -m pointfoot_amd.scripts.train --task=pointfoot_rough --headless [--num_envs N ...]
(mirror of reference legged_gym/scripts/train.py:41-49).  Under `torchrun` each rank trains its env shard on its own
GPU and gradients are all-reduced over RCCL."""
import os

import torch
import torch.distributed as dist

from pointfoot_amd.envs import *  # noqa: F401,F403  (registers the tasks)
from pointfoot_amd import dp
from pointfoot_amd.utils import get_args, task_registry


def _maybe_init_distributed(args):
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and not dist.is_initialized():
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        dp.init_process_group("nccl", local)     # RCCL on ROCm, bound to this rank's GPU (device_id)
        args.sim_device = args.rl_device = f"cuda:{local}"


def train(args):
    _maybe_init_distributed(args)
    env, env_cfg = task_registry.make_env(name=args.task, args=args)
    ppo_runner, train_cfg = task_registry.make_alg_runner(env=env, name=args.task, args=args)
    ppo_runner.learn(num_learning_iterations=train_cfg.runner.max_iterations, init_at_random_ep_len=True)


if __name__ == '__main__':
    train(get_args())
