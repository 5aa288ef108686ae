"""Env-sharded data parallelism helpers.  One process per GPU; backend "nccl" is RCCL on ROCm (xGMI
inside a node), "gloo" in the CPU tests.  The path shards over independent environments, so the only data-path
collectives are: the flat-gradient all-reduce per minibatch (with the KL sum and sample count riding in the tail so every
rank takes the same adaptive-LR decision), a 3-scalar all-reduce for global advantage statistics, and a one-time
parameter broadcast."""
import os

import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard(num_envs_per_rank, rank=None, world_size=None):
    """global env index range owned by `rank`: [base, base + n)"""
    if rank is None or world_size is None:
        r, w = world()
        rank = r if rank is None else rank
        world_size = w if world_size is None else world_size
    return rank * num_envs_per_rank, num_envs_per_rank * world_size


def sync_params(flat):
    rank, ws = world()
    if ws > 1:
        dist.broadcast(flat, src=0)
    return flat


def allreduce_grads(flat_with_tail):
    """sum over ranks, in place; the consumer divides by world_size (ppo_minibatch_apply: gscale = 1/world_size)"""
    rank, ws = world()
    if ws > 1:
        dist.all_reduce(flat_with_tail)
    return flat_with_tail


def allreduce_adv_stats(stats3):
    rank, ws = world()
    if ws > 1:
        dist.all_reduce(stats3)
    return stats3


def init_from_env(backend="nccl"):
    """torchrun-style environment (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*)."""
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and not dist.is_initialized():
        dist.init_process_group(backend)
    return world()
