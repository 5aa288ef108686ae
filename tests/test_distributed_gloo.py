"""world_size-2 tests of the N>1 path on CPU (gloo): (1) env sharding — two ranks of N envs produce exactly the
observations of one process with 2N envs (Philox streams and terrain types are keyed by the GLOBAL env index);
(2) data-parallel update — all-reducing per-shard gradients (+ KL tail) through pointfoot_amd.dp and applying
clip + Adam gives the single-process result on the concatenated batch, with identical learning-rate decisions."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import common

sys.path.insert(0, os.path.join(common.ROOT, "oracle"))


def _env_shard_sim(n, base, n_global):
    import synth
    cfg = common.make_cfg("rough", n)
    hf, to = synth.synth_heightfield(), synth.synth_terrain_origins().astype(np.float32)
    sim = common.HostSim(cfg, n, hf=hf, terrain_origins=to, seed=5)
    sim.c.env_index_base, sim.c.num_envs_global = base, n_global
    rs = np.random.RandomState(11)
    full = dict(root=rs.uniform(-1, 1, (n_global, 2)), q=rs.uniform(-0.1, 0.1, (n_global, 6)), lvl=rs.randint(0, 6, n_global))
    g = np.arange(base, base + n)
    sim.terrain_level[:] = full["lvl"][g]
    sim.terrain_type[:] = g // max(n_global // 20, 1) % 20
    sim.f("env_origin", 3)[:] = to[sim.terrain_level, sim.terrain_type]
    sim.f("root", 13)[:, :3] = sim.f("env_origin", 3)
    sim.f("root", 13)[:, :2] += full["root"][g]
    sim.f("root", 13)[:, 2] += 0.75
    sim.f("dof_pos", 6)[:] = full["q"][g]
    sim.f("friction")[:] = 0.8
    return sim, g


def test_env_shards_equal_one_big_run():
    n, W = 16, 2
    big, _ = _env_shard_sim(n * W, 0, n * W)
    shards = [_env_shard_sim(n, r * n, n * W) for r in range(W)]
    rs = np.random.RandomState(2)
    for t in range(6):
        act = rs.normal(0, 0.6, (n * W, 6)).astype(np.float32)
        big.step("emu", act, t + 1)
        for sim, g in shards:
            sim.step("emu", act[g], t + 1)
            np.testing.assert_array_equal(sim.obs, big.obs[g])
            np.testing.assert_array_equal(sim.priv, big.priv[g])
            np.testing.assert_array_equal(sim.rew, big.rew[g])


def _dp_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import ppo_ref
    from pointfoot_amd import dp
    torch.manual_seed(100 + rank)                      # ranks start from DIFFERENT weights ...
    ac = ppo_ref.RefActorCritic(27, 27, 6, [32, 16], [32, 16])
    flat = ac.flat().clone()
    dp.sync_params(flat)                               # ... and are made identical by the rank-0 broadcast
    ac.load_flat(flat)
    g = torch.Generator().manual_seed(7)               # same global batch everywhere, each rank takes its slice
    B = 64
    data = dict(obs=torch.randn(B * world, 27, generator=g), critic_obs=torch.randn(B * world, 27, generator=g),
                actions=torch.randn(B * world, 6, generator=g), values=torch.randn(B * world, generator=g),
                returns=torch.randn(B * world, generator=g), advantages=torch.randn(B * world, generator=g),
                old_log_prob=-8 + torch.randn(B * world, generator=g) * 0.1, old_mu=torch.randn(B * world, 6, generator=g) * 0.1,
                old_sigma=torch.ones(B * world, 6))
    mb = {k: v[rank * B:(rank + 1) * B] for k, v in data.items()}
    loss, kl, _, _ = ppo_ref.minibatch_loss(ac, mb, 0.2, 1.0, 0.01)
    ac.zero_grad(); loss.backward()
    grads = torch.cat([ac.flat_grad(), torch.tensor([kl.item() * B, float(B), 0.0, 0.0])])     # tail: KL sum, count
    dp.allreduce_grads(grads)
    P = grads.numel() - 4
    gmean = grads[:P] / world                          # what ppo_minibatch_apply does (gscale = 1/world_size)
    kl_global = (grads[P] / grads[P + 1]).item()
    stats = torch.tensor([float(rank + 1), 2.0, 3.0]); dp.allreduce_adv_stats(stats)
    if rank == 0:
        torch.save(dict(flat=flat, gmean=gmean, kl=kl_global, stats=stats, data=data), out)
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_gradient_matches_single_process(tmp_path):
    import ppo_ref
    out = str(tmp_path / "dp.pt")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_dp_worker, args=(2, port, out), nprocs=2, join=True)
    r = torch.load(out, weights_only=True)
    ac = ppo_ref.RefActorCritic(27, 27, 6, [32, 16], [32, 16])
    ac.load_flat(r["flat"])
    loss, kl, _, _ = ppo_ref.minibatch_loss(ac, r["data"], 0.2, 1.0, 0.01)      # the concatenated global batch
    ac.zero_grad(); loss.backward()
    torch.testing.assert_close(r["gmean"], ac.flat_grad(), atol=1e-6, rtol=1e-4)
    assert abs(r["kl"] - kl.item()) < 1e-6
    torch.testing.assert_close(r["stats"], torch.tensor([3.0, 4.0, 6.0]))


class _FakeInfo(dict):
    """stand-in for hip_env._EpisodeInfo: the per-step sums over this rank's reset envs, their count, a per-rank mean"""
    def __init__(self, sums, count, level):
        super().__init__()
        self._r = dict(names=["a", "b"], sums=torch.tensor(sums + [0.0]), count=torch.tensor(float(count)), scale=2.0,
                       mean_extras={"terrain_level": torch.tensor(level)}, const_extras={"max_command_x": 1.5})
        self.raw = lambda: self._r


def _ep_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pointfoot_amd import dp
    # step 0: rank 0 reset 1 env (sums 2, 4), rank 1 reset 3 envs (sums 10, 20); step 1: nobody reset anything;
    # step 2: only rank 1 reset (2 envs)
    infos = [[_FakeInfo([2.0, 4.0], 1, 1.0), _FakeInfo([0.0, 0.0], 0, 1.0), _FakeInfo([0.0, 0.0], 0, 1.0)],
             [_FakeInfo([10.0, 20.0], 3, 3.0), _FakeInfo([0.0, 0.0], 0, 3.0), _FakeInfo([6.0, 8.0], 2, 3.0)]][rank]
    red = dp.reduce_episode_infos(infos, "cpu")
    tot = dp.allreduce_sum(torch.tensor([1.0 + rank, 10.0, 2.0]))
    if rank == 0:
        torch.save(dict(red=red, tot=tot), out)
    dist.barrier()
    dist.destroy_process_group()


def test_episode_statistics_are_count_weighted_across_ranks(tmp_path):
    """the logged episode means are those of ONE process with all envs -- sums and counts are reduced over the
    shards before dividing (a rank with 1 reset env must not weigh like a rank with 3)."""
    out = str(tmp_path / "ep.pt")
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_ep_worker, args=(2, port, out), nprocs=2, join=True)
    r = torch.load(out, weights_only=True)
    red = r["red"]
    # step 0: (2 + 10) / 4 / scale = 1.5, step 2: 6 / 2 / 2 = 1.5 (a), step 1 is NaN and skipped
    assert abs(red["rew_a"] - 1.5) < 1e-6
    assert abs(red["rew_b"] - ((4 + 20) / 4 / 2 + 8 / 2 / 2) / 2) < 1e-6
    assert abs(red["terrain_level"] - 2.0) < 1e-6 and red["max_command_x"] == 1.5
    torch.testing.assert_close(r["tot"], torch.tensor([3.0, 20.0, 4.0]))
    # one rank alone: the plain per-step means
    from pointfoot_amd import dp
    solo = dp.reduce_episode_infos([_FakeInfo([2.0, 4.0], 1, 1.0), _FakeInfo([0.0, 0.0], 0, 1.0)], "cpu")
    assert abs(solo["rew_a"] - 1.0) < 1e-6 and abs(solo["rew_b"] - 2.0) < 1e-6


def _ring(entries, seen_steps):
    """a PpoBuffers.d_ep_ring image holding `entries` = [(return, length, step, seq)], oldest first"""
    import numpy as np
    from pointfoot_amd import _abi
    r = np.zeros(_abi.PPO_EP_RING_INTS, np.int32)
    n = len(entries)
    assert n <= _abi.PPO_EP_RING
    head = 37                                    # somewhere in the ring: the merge must honour the wrap-around
    for k, (ret, ln, stp, seq) in enumerate(entries):
        slot = (head - n + k) % _abi.PPO_EP_RING
        r[4 + 4 * slot: 8 + 4 * slot] = [np.float32(ret).view(np.int32), np.float32(ln).view(np.int32), stp, seq]
    r[0], r[1], r[2], r[3] = head, n, (entries[-1][3] + 1 if entries else 0), seen_steps
    return torch.from_numpy(r)


def _ring_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pointfoot_amd import dp
    import random
    rng = random.Random(5)
    # one process with 2 x 40 envs: per step a few envs finish; env e belongs to rank e // 40
    seq = [0, 0]
    mine, merged = [], []
    for step in range(60):
        for e in sorted(rng.sample(range(80), rng.choice([0, 1, 3, 9]))):
            r = e // 40
            item = (float(step * 100 + e), float(e + 1), step, seq[r])
            seq[r] += 1
            merged.append(item)
            if r == rank:
                mine.append(item)
    rets, lens = dp.merge_episode_rings(_ring(mine[-100:], 60))
    if rank == 0:
        torch.save(dict(rets=rets, lens=lens, want_r=[m[0] for m in merged[-100:]], want_l=[m[1] for m in merged[-100:]]), out)
    dist.barrier()
    dist.destroy_process_group()


def test_episode_rings_of_two_ranks_merge_into_one_process_deque(tmp_path):
    """Train/mean_reward and Train/mean_episode_length under data parallelism: the ranks' rings of finished episodes are
    concatenated in the order ONE process with all the envs would have appended them -- (policy step, rank, position) -- and the
    last 100 kept (rsl_rl: deque(maxlen=100))."""
    out = str(tmp_path / "ring.pt")
    port = 33500 + (os.getpid() % 2000)
    mp.spawn(_ring_worker, args=(2, port, out), nprocs=2, join=True)
    r = torch.load(out, weights_only=True)
    assert len(r["rets"]) == 100 and r["rets"] == r["want_r"] and r["lens"] == r["want_l"]
    # a single process: the ring itself, oldest first, across the wrap-around
    from pointfoot_amd import dp
    ents = [(1.5 * k, float(k), k // 3, k) for k in range(70)]
    rets, lens = dp.merge_episode_rings(_ring(ents, 24))
    assert rets == [e[0] for e in ents] and lens == [e[1] for e in ents]
    assert dp.merge_episode_rings(_ring([], 0)) == ([], [])
