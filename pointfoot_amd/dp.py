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


def collectives_on():
    """True when the collectives of the path have to be issued: more than one rank -- or a group of ONE rank with
    PF_DP_SINGLE_RANK_COLLECTIVES=1, the hardware rehearsal of the N>1 code path on a one-GPU box (every all-reduce /
    broadcast of the path then really goes through RCCL and its stream ordering; the sums are identities)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("PF_DP_SINGLE_RANK_COLLECTIVES", "0") == "1"


def shard(num_envs_per_rank, rank=None, world_size=None):
    """global env index range owned by `rank`: [base, base + n)"""
    if rank is None or world_size is None:
        r, w = world()
        rank = r if rank is None else rank
        world_size = w if world_size is None else world_size
    return rank * num_envs_per_rank, num_envs_per_rank * world_size


def sync_params(flat):
    if collectives_on():
        dist.broadcast(flat, src=0)
    return flat


def allreduce_grads(flat_with_tail):
    """sum over ranks, in place; the consumer divides by world_size (ppo_minibatch_apply: gscale = 1/world_size)"""
    if collectives_on():
        dist.all_reduce(flat_with_tail)
    return flat_with_tail


def allreduce_async(t):
    """start the sum over ranks of one contiguous gradient range; returns the work handle (None on one rank).  On the nccl
    (= RCCL) backend the collective is ordered after the kernels already issued on the current stream and runs on the process
    group's own stream -- beside whatever is issued next; `wait()` makes the current stream wait for it (no host sync)."""
    if collectives_on():
        return dist.all_reduce(t, async_op=True)
    return None


def allreduce_adv_stats(stats3):
    if collectives_on():
        dist.all_reduce(stats3)
    return stats3


def allreduce_sum(t):
    """in-place sum over ranks of a small statistics tensor (episode sums and counts); a no-op on one rank"""
    if collectives_on():
        dist.all_reduce(t)
    return t


def reduce_episode_infos(ep_infos, device):
    """The iteration's `extras["episode"]` entries -> one dict of floats, as ONE process with all the envs would log it
    (reference: legged_gym/envs/pointfoot/point_foot.py:276-285 takes the mean over the envs reset on a step; rsl_rl's
    runner averages those per-step means over the iteration).  Each rank holds, per step, the SUMS over its own reset envs
    and their COUNT (`_EpisodeInfo.raw()`); summing both over ranks before dividing is the count-weighted mean across shards
     -- averaging the ranks' means would weight a rank with one reset env like a rank with a hundred.
    Collective: every rank must call it (with the same number of entries) when world_size > 1."""
    import torch
    rank, ws = world()
    out = {}
    raws = [e.raw() for e in ep_infos if hasattr(e, "raw") and e.raw is not None]
    if raws and len(raws) == len(ep_infos):
        names = raws[0]["names"]
        n = len(names)
        extra_keys = sorted(raws[0]["mean_extras"].keys())
        M = torch.stack([torch.cat([r["sums"][:n].reshape(-1).float(), r["count"].reshape(1).float(),
                                    torch.stack([torch.as_tensor(r["mean_extras"][k], dtype=torch.float, device=r["sums"].device).reshape(()) for k in extra_keys]).reshape(-1)
                                    if extra_keys else r["sums"].new_zeros(0)]) for r in raws]).to(device)
        allreduce_sum(M)
        cnt = M[:, n:n + 1]
        per_step = M[:, :n] / cnt / raws[0]["scale"]          # NaN on steps where no env was reset anywhere
        vals = torch.nanmean(per_step, dim=0)
        for i, k in enumerate(names):
            out["rew_" + k] = float(vals[i])
        for j, k in enumerate(extra_keys):
            out[k] = float(M[:, n + 1 + j].mean() / ws)         # per-rank means over equal shard sizes
        for k, v in raws[0]["const_extras"].items():
            out[k] = float(v)
        return out
    # entries without raw statistics (another VecEnv): per-rank values, averaged over steps; ranks are averaged unweighted
    if ep_infos:
        for key in ep_infos[0]:
            vals = [ep[key] for ep in ep_infos if key in ep]
            v = torch.nanmean(torch.stack([torch.as_tensor(x, dtype=torch.float).reshape(()).to(device) for x in vals])).reshape(1)
            if collectives_on():
                v = torch.nan_to_num(v); dist.all_reduce(v); v = v / ws
            out[key] = float(v)
    return out


def merge_episode_rings(ring, maxlen=100):
    """The device-side ring of finished episodes (PpoBuffers.d_ep_ring: header {next slot, held, seen, steps}, then per slot
    {return bits, length bits, policy-step index, sequence number}) -> (returns, lengths) as rsl_rl's OnPolicyRunner would hold
    them in `rewbuffer` / `lenbuffer = deque(maxlen=100)`: the last 100 FINISHED EPISODES, oldest first.  Under torch.distributed
    the ranks' rings are concatenated: ONE process with all the envs appends the episodes of a step in global env order, i.e.
    ordered by (policy step, rank, position inside the rank); the last 100 of that sequence are among the ranks' own last 100s.
    Collective when collectives are on: every rank must call it."""
    import torch
    rank, ws = world()
    rings = [ring]
    if collectives_on():
        rings = [torch.empty_like(ring) for _ in range(ws)]
        dist.all_gather(rings, ring.contiguous())
    entries = []
    for r, t in enumerate(rings):
        host = t.detach().cpu()
        held = int(host[1])
        body = host[4:].view(-1, 4)
        vals = body[:, :2].contiguous().view(torch.float32)
        head = int(host[0])
        for k in range(held):              # oldest first: the `held` slots that end just before `head`
            slot = (head - held + k) % body.shape[0]
            entries.append((int(body[slot, 2]), r, int(body[slot, 3]), float(vals[slot, 0]), float(vals[slot, 1])))
    entries.sort(key=lambda e: e[:3])
    entries = entries[-maxlen:]
    return [e[3] for e in entries], [e[4] for e in entries]


def init_process_group(backend="nccl", local_rank=None):
    """`dist.init_process_group` for one process per GPU.  On the nccl (= RCCL) backend the process group is BOUND to this
    rank's device (`device_id`): RCCL then creates its communicator eagerly on that GPU instead of guessing the device at the
    first collective ("can cause a hang if rank to GPU mapping is heterogeneous"), and `barrier()` knows which device to use.
    Returns (rank, world_size)."""
    import torch
    if dist.is_initialized():
        return world()
    if local_rank is None:
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        if os.environ.get("PF_DP_NO_DEVICE_ID") == "1":      # (A/B switch: the unbound form of rounds 1-3)
            dist.init_process_group(backend)
        else:
            dist.init_process_group(backend, device_id=torch.device(f"cuda:{local_rank}"))
    else:
        dist.init_process_group(backend)
    return world()


def backend_name():
    """the collective library the gradient all-reduce really goes through ("nccl" = RCCL on ROCm), or None"""
    if dist.is_available() and dist.is_initialized():
        return str(dist.get_backend())
    return None


def init_from_env(backend="nccl"):
    """torchrun-style environment (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*)."""
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and not dist.is_initialized():
        init_process_group(backend)
    return world()
