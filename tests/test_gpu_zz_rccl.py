"""The multi-GPU code path through RCCL itself, on the one GPU a test box has: a process group of ONE rank on
the `nccl` backend (= RCCL on ROCm) with PF_DP_SINGLE_RANK_COLLECTIVES=1, so that every collective of the N>1 path -- the
parameter broadcast, the advantage-statistics all-reduce, the two asynchronous gradient all-reduces under the split backward
pass (or the one blocking all-reduce), each ordered against the HIP kernels by RCCL's own stream events -- is really issued.
On one rank every sum is an identity: the update must leave bit-identical parameters to the same update without a process
group.  (Two ranks cannot share one device under RCCL; the world_size-2 rehearsals are the gloo tests.)"""
import os
import socket

import pytest
import torch
import torch.distributed as dist

import test_gpu_ppo as tp

pytestmark = pytest.mark.gpu


def _update(seed=11):
    N, T, O, P, A, hidden = 128, 8, 27, 148, 6, [512, 256, 128]
    alg, ref = tp.make(N, T, O, P, A, hidden, seed=seed)
    gen = torch.Generator(device="cuda:0").manual_seed(seed + 1)
    tp.fill_rollout(alg, ref, N, T, O, P, A, gen)
    torch.manual_seed(seed + 2)      # the update's randperm
    alg.update()
    torch.cuda.synchronize()
    return alg.params.clone(), float(alg.scalars[0]), alg.storage.advantages.clone()


def test_update_through_a_single_rank_rccl_group_is_the_plain_update(monkeypatch):
    from pointfoot_amd import dp
    # the reference run takes the separate backward / apply calls too (ppo_minibatch_step, PF_PPO_FUSED_STEP=1, sums the squared
    # gradient norm in another order: tests/test_gpu_ppo.py compares the two)
    monkeypatch.setenv("PF_PPO_FUSED_STEP", "0")
    base_params, base_lr, base_adv = _update()
    assert not dp.collectives_on()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    torch.cuda.set_device(0)
    try:
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    except Exception as ex:      # an image without a usable RCCL transport: nothing to rehearse
        pytest.skip(f"nccl process group of one rank did not come up: {ex!r}")
    try:
        monkeypatch.setenv("PF_DP_SINGLE_RANK_COLLECTIVES", "1")
        assert dp.collectives_on()
        t = torch.arange(8, device="cuda:0", dtype=torch.float32)
        dist.all_reduce(t)
        assert torch.equal(t.cpu(), torch.arange(8, dtype=torch.float32))
        for overlap in ("1", "0"):      # two async all-reduces under the split backward pass / one blocking all-reduce
            monkeypatch.setenv("PF_PPO_OVERLAP_ALLREDUCE", overlap)
            params, lr, adv = _update()
            assert torch.equal(adv, base_adv), f"overlap={overlap}: advantages differ (advantage-statistics all-reduce)"
            assert lr == base_lr, f"overlap={overlap}: adaptive learning rate differs (KL tail of the gradient all-reduce)"
            assert torch.equal(params, base_params), f"overlap={overlap}: parameters differ after 20 optimiser steps"
    finally:
        monkeypatch.delenv("PF_DP_SINGLE_RANK_COLLECTIVES", raising=False)
        dist.destroy_process_group()
    assert not dp.collectives_on()
