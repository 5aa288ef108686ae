"""GPU parity tests of the HIP PPO kernels (through the ppo_* C ABI) against the plain-PyTorch fp32 restatement of
rsl_rl's PPO in oracle/ppo_ref.py (autograd reference; rsl_rl itself is absent: parity unpinned)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch

import common

sys.path.insert(0, os.path.join(common.ROOT, "oracle"))
import ppo_ref  # noqa: E402

pytestmark = pytest.mark.gpu

ALG = dict(value_loss_coef=1.0, use_clipped_value_loss=True, clip_param=0.2, entropy_coef=0.01, num_learning_epochs=5,
           num_mini_batches=4, learning_rate=1e-3, schedule="adaptive", gamma=0.99, lam=0.95, desired_kl=0.01, max_grad_norm=1.0)


def make(N, T, O, P, A, hidden, seed=0):
    from pointfoot_amd.rsl_rl.modules import ActorCritic
    from pointfoot_amd.rsl_rl.algorithms import PPO
    torch.manual_seed(seed)
    ac = ActorCritic(O, P, A, hidden, hidden, "elu", 1.0)
    alg = PPO(ac, device="cuda:0", **ALG)
    alg.init_storage(N, T, [O], [P], [A])
    ref = ppo_ref.RefActorCritic(O, P, A, hidden, hidden).to("cuda:0")
    ref.load_flat(alg.params, alg.param_offsets)
    return alg, ref


def fill_rollout(alg, ref, N, T, O, P, A, gen):
    dev = "cuda:0"
    R = lambda *s: torch.randn(*s, device=dev, generator=gen)
    recs = []
    for t in range(T):
        obs, cobs, noise = R(N, O), R(N, P), R(N, A)
        alg.inject_normal = noise
        a = alg.act(obs, cobs).clone()
        r = ppo_ref.act(ref, obs, cobs, noise)
        torch.testing.assert_close(a, r["actions"], atol=2e-5, rtol=1e-5)
        st = alg.storage
        torch.testing.assert_close(st.values[t, :, 0], r["values"][:, 0], atol=2e-5, rtol=1e-5)
        torch.testing.assert_close(st.actions_log_prob[t, :, 0], r["log_prob"], atol=1e-4, rtol=1e-5)
        torch.testing.assert_close(st.mu[t], r["mu"], atol=2e-5, rtol=1e-5)
        torch.testing.assert_close(st.observations[t], obs)
        torch.testing.assert_close(st.privileged_observations[t], cobs)
        rew = R(N)
        dones = torch.rand(N, device=dev, generator=gen) < 0.1
        tos = dones & (torch.rand(N, device=dev, generator=gen) < 0.5)
        alg.process_env_step(rew, dones, {"time_outs": tos})
        rr = ppo_ref.process_rewards(rew, r["values"], tos, ALG["gamma"])
        torch.testing.assert_close(st.rewards[t, :, 0], rr, atol=1e-5, rtol=1e-5)
        recs.append(dict(rewards=rr, dones=dones.float(), values=r["values"][:, 0]))
    last = R(N, P)
    alg.compute_returns(last)
    with torch.no_grad():
        lv = ref.critic(last)[:, 0]
    V = torch.stack([x["values"] for x in recs]); Rw = torch.stack([x["rewards"] for x in recs]); D = torch.stack([x["dones"] for x in recs])
    ret, adv, _ = ppo_ref.compute_returns(V, Rw, D, lv, ALG["gamma"], ALG["lam"])
    torch.testing.assert_close(alg.storage.returns[:, :, 0], ret, atol=1e-4, rtol=1e-4)
    torch.testing.assert_close(alg.storage.advantages[:, :, 0], adv, atol=2e-4, rtol=1e-4)
    return last


def ref_store(alg):
    st = alg.storage
    f = lambda x: x.flatten(0, 1).clone()
    return dict(obs=f(st.observations), critic_obs=f(st.privileged_observations), actions=f(st.actions),
                values=f(st.values)[:, 0], returns=f(st.returns)[:, 0], advantages=f(st.advantages)[:, 0],
                old_log_prob=f(st.actions_log_prob)[:, 0], old_mu=f(st.mu), old_sigma=f(st.sigma))


# execution variants of the same arithmetic (read by ppo_create): default = dense layers by exact bf16 split on the bf16 matrix
# pipes (pf_gemm3.h), actor and critic layers paired in one launch, fused head kernels; PF_PPO_GEMM=1 selects the plain fp32
# MFMA GEMMs; the other fallbacks run the critic on a second stream / the separate head, loss and sampling kernels
VARIANTS = [{}, {"PF_PPO_GEMM": "1"}, {"PF_PPO_GEMM": "2"}, {"PF_PPO_PAIRED": "0", "PF_PPO_GEMM": "2"}, {"PF_PPO_PAIRED": "0", "PF_PPO_GEMM": "1"},
            {"PF_PPO_FUSED_HEADS": "0"}, {"PF_PPO_FUSED_HEADS": "0", "PF_PPO_DUAL_STREAM": "0"}]


@pytest.mark.parametrize("variant", VARIANTS, ids=lambda v: "+".join(f"{k[7:]}={x}" for k, x in v.items()) or "default")
@pytest.mark.parametrize("N,T,O,P,A,hidden", [(96, 8, 27, 27, 6, [128, 64, 32]), (160, 6, 27, 148, 6, [512, 256, 128])])
def test_rollout_gae_and_full_update_match_autograd(N, T, O, P, A, hidden, variant, monkeypatch):
    for k, v in variant.items():
        monkeypatch.setenv(k, v)
    alg, ref = make(N, T, O, P, A, hidden)
    gen = torch.Generator(device="cuda:0").manual_seed(1)
    fill_rollout(alg, ref, N, T, O, P, A, gen)
    store = ref_store(alg)
    perm = torch.randperm(N * T, device="cuda:0", generator=gen)
    # --- one minibatch: gradients
    alg.perm.copy_(perm.to(torch.int32))
    common._abi.check(alg._lib.ppo_minibatch_backward(alg._h, 1, alg._stream()))
    B = N * T // 4
    idx = perm[B:2 * B].long()
    mb = {k: v[idx] for k, v in store.items()}
    loss, kl, vl, sl = ppo_ref.minibatch_loss(ref, mb, ALG["clip_param"], ALG["value_loss_coef"], ALG["entropy_coef"])
    ref.zero_grad()
    loss.backward()
    g_ref = ref.flat_grad(alg.param_offsets, alg.num_params)
    g = alg.grads[:alg.num_params]
    scale = g_ref.abs().max().item()
    assert (g - g_ref).abs().max().item() <= 2e-4 * scale + 1e-7, ((g - g_ref).abs().max().item(), scale)
    assert abs(alg.grads[alg.num_params].item() / alg.grads[alg.num_params + 1].item() - kl.item()) <= 1e-5 + 1e-3 * abs(kl.item())
    # --- the whole update (5 epochs x 4 minibatches, adaptive LR, clip, Adam)
    opt = torch.optim.Adam(ref.parameters(), lr=ALG["learning_rate"])
    cfg = dict(ALG)
    lr, mv, ms = ppo_ref.update(ref, opt, store, perm, cfg, ALG["learning_rate"])
    orig = torch.randperm
    torch.randperm = lambda n, **kw: perm
    try:
        v_loss, s_loss = alg.update()
    finally:
        torch.randperm = orig
    torch.cuda.synchronize()
    assert abs(alg.current_learning_rate.item() - lr) <= 1e-9 + 1e-6 * lr, (alg.current_learning_rate.item(), lr)
    p_ref = ref.flat(alg.param_offsets, alg.num_params)
    err = (alg.params - p_ref).abs().max().item()
    assert err <= 3e-4, err
    assert abs(v_loss.item() - mv) <= 1e-3 * max(1.0, abs(mv)) and abs(s_loss.item() - ms) <= 1e-3 * max(1.0, abs(ms))
    # nn.Module views see the kernel's updates (flat-buffer aliasing)
    torch.testing.assert_close(alg.actor_critic.actor[0].weight, ref.actor[0].weight, atol=3e-4, rtol=0)


def test_full_size_minibatch_gradient():
    """BASELINE shapes (4096 envs x 24 steps, 512/256/128, 148-dim critic): one minibatch forward/backward vs autograd."""
    N, T, O, P, A, hidden = 4096, 24, 27, 148, 6, [512, 256, 128]
    alg, ref = make(N, T, O, P, A, hidden)
    dev = "cuda:0"
    gen = torch.Generator(device=dev).manual_seed(3)
    st = alg.storage
    R = lambda *s: torch.randn(*s, device=dev, generator=gen)
    st.observations.copy_(R(T, N, O)); st.privileged_observations.copy_(R(T, N, P))
    with torch.no_grad():
        mu = ref.actor(st.observations.flatten(0, 1)).view(T, N, A)
        st.mu.copy_(mu + 0.05 * R(T, N, A)); st.sigma.fill_(1.0)
        st.actions.copy_(mu + R(T, N, A))
        st.values.copy_(ref.critic(st.privileged_observations.flatten(0, 1)).view(T, N, 1) + 0.1 * R(T, N, 1))
        st.actions_log_prob.copy_((-0.5 * (st.actions - st.mu) ** 2 - 0.9189385).sum(-1, keepdim=True))
    st.returns.copy_(R(T, N, 1)); st.advantages.copy_(R(T, N, 1))
    perm = torch.randperm(N * T, device=dev, generator=gen)
    alg.perm.copy_(perm.to(torch.int32))
    common._abi.check(alg._lib.ppo_minibatch_backward(alg._h, 2, alg._stream()))
    store = ref_store(alg)
    B = N * T // 4
    mb = {k: v[perm[2 * B:3 * B].long()] for k, v in store.items()}
    loss, kl, _, _ = ppo_ref.minibatch_loss(ref, mb, ALG["clip_param"], ALG["value_loss_coef"], ALG["entropy_coef"])
    ref.zero_grad(); loss.backward()
    g_ref, g = ref.flat_grad(alg.param_offsets, alg.num_params), alg.grads[:alg.num_params]
    scale = g_ref.abs().max().item()
    assert (g - g_ref).abs().max().item() <= 3e-4 * scale, ((g - g_ref).abs().max().item(), scale)
    # determinism: a second launch gives bit-identical gradients (atomic-free reductions)
    g1 = g.clone()
    common._abi.check(alg._lib.ppo_minibatch_backward(alg._h, 2, alg._stream()))
    assert torch.equal(g1, alg.grads[:alg.num_params])
    # storage permuted once (ppo_begin_update): every minibatch is a slice of it, same bits as gathering per minibatch
    common._abi.check(alg._lib.ppo_begin_update(alg._h, alg._stream()))
    common._abi.check(alg._lib.ppo_minibatch_backward(alg._h, 2, alg._stream()))
    assert torch.equal(g1, alg.grads[:alg.num_params])
    common._abi.check(alg._lib.ppo_minibatch_backward(alg._h, 0, alg._stream()))
    g0 = alg.grads[:alg.num_params].clone()
    common._abi.check(alg._lib.ppo_normalize_advantages(alg._h, alg._stream()))      # drops the copy
    st.advantages.copy_(store["advantages"].view(T, N, 1))
    common._abi.check(alg._lib.ppo_minibatch_backward(alg._h, 0, alg._stream()))
    assert torch.equal(g0, alg.grads[:alg.num_params])


def test_gemm_shapes_against_torch():
    """act_inference on odd batch sizes exercises the GEMM's M/N/K edge handling (K=27/148, M not a tile multiple)."""
    alg, ref = make(200, 4, 27, 148, 6, [512, 256, 128])
    for n in (1, 50, 199, 200):
        x = torch.randn(n, 27, device="cuda:0")
        out = torch.zeros(n, 6, device="cuda:0")
        common._abi.check(alg._lib.ppo_act_inference(alg._h, x.data_ptr(), n, out.data_ptr(), alg._stream()))
        with torch.no_grad():
            torch.testing.assert_close(out, ref.actor(x), atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize("path", ["fp32_mfma", "bf16x6", "bf16x6_planes", "bf16x6_dma", "bf16x6_dma_nothin"])
@pytest.mark.parametrize("mode", [0, 1, 2], ids=["fwd_bias_elu", "bwd_x_delu", "bwd_w_splitk_colsum"])
def test_gemm_kernel_family_on_ragged_shapes(mode, path, monkeypatch):
    """Both GEMM families through the test hook -- the fp32 MFMA kernels (pf_gemm.h) and the fp32-by-exact-bf16-split
    kernels (pf_gemm3.h, "bf16x6") -- every tile variant, on shapes that exercise the edges: rows / columns / reduction
    lengths that are not multiples of the tile, of 4 (scalar loads and write-out) or of the k-slice, single-row
    problems, and split-K with a ragged last split.  Reference: FLOAT64 matmul; the tolerance is the float32 rounding
    budget of a length-K dot product and is THE SAME for both paths (the split drops only terms below 2^-23 |a b|)."""
    if path in ("bf16x6_planes", "bf16x6_dma", "bf16x6_dma_nothin") and mode == 2:
        pytest.skip("the weight gradient has no pre-split operand (both operands are activations)")
    # launches of at most 128 64-column workgroups (all of these shapes) are moved to the 32-column instance of the LDS-DMA kernel
    # -- the rollout's last hidden layer; "_nothin" switches that off, so that tile 1 / -1 reach the 64-column instance too
    monkeypatch.setenv("PF_GEMM3F_THIN", "0" if path.endswith("_nothin") else "1")
    path = path.replace("_nothin", "")
    lib = common._abi.load()
    lib.pf_gemm_test.argtypes = [C.c_int] * 4 + [C.c_void_p] * 6 + [C.c_int, C.c_int, C.c_void_p]
    dev = "cuda:0"
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    gen = torch.Generator(device=dev).manual_seed(7)
    R = lambda *s: torch.rand(*s, device=dev, generator=gen) * 2 - 1
    shapes = [(1, 5, 3), (64, 64, 32), (130, 70, 45), (257, 96, 100), (200, 132, 260), (96, 28, 1000), (33, 512, 27), (300, 148, 4100)]
    if path == "bf16x6_dma":
        # the LDS-DMA kernel (pf_gemm3f.h) moves 16-byte chunks: K and N multiples of 4 (the PPO's padded layers); everything else --
        # ragged M / N against the 128 x {128, 64} tiles, a ragged last k-slice, one-row problems -- is its own edge handling
        shapes = [(1, 8, 4), (64, 64, 32), (130, 68, 44), (257, 96, 100), (200, 132, 260), (96, 28, 1000), (33, 512, 28), (300, 148, 4100), (385, 260, 36)]
    base = {"fp32_mfma": 0, "bf16x6": 10, "bf16x6_planes": 20, "bf16x6_dma": 30}[path]
    worst = 0.0
    for (M, N, K) in shapes:
        for tile in ((0, 1, 2, -1) if path == "fp32_mfma" else (0, 1, 2, -1) if path == "bf16x6_dma" else (0, 1, 2, 3, -1)):
            for splits in ((1, 7, 64) if mode == 2 else (0,)):
                if mode == 0: A, B = R(M, K), R(N, K)
                elif mode == 1: A, B = R(M, K), R(K, N)
                else: A, B = R(K, M), R(K, N)
                if (M, N, K) == (257, 96, 100):          # wide dynamic range: exercises the low-order split terms
                    A = A * torch.exp(6 * R(*A.shape)); B = B * torch.exp(6 * R(*B.shape))
                bias, H = R(N), R(M, N)
                out = torch.full((M * N + M,), float("nan"), device=dev)
                ws = torch.zeros(max(max(splits, 1) * (M * N + M), 2 * N * (K + 8)) + 64, device=dev)
                rc = lib.pf_gemm_test(mode + base, M, N, K, A.data_ptr(), B.data_ptr(), out.data_ptr(),
                                      bias.data_ptr(), H.data_ptr(), ws.data_ptr(), max(splits, 1), tile, stream)
                assert rc == 0, lib.pf_last_error()
                torch.cuda.synchronize()
                A64, B64 = A.double(), B.double()
                if mode == 0: ref = torch.nn.functional.elu(A64 @ B64.t() + bias.double()); mag = A64.abs() @ B64.abs().t()
                elif mode == 1: ref = (A64 @ B64) * torch.where(H > 0, torch.ones_like(H), H + 1).double(); mag = A64.abs() @ B64.abs()
                else: ref = A64.t() @ B64; mag = A64.abs().t() @ B64.abs()
                got = out[:M * N].view(M, N).double()
                assert torch.isfinite(got).all(), (path, mode, M, N, K, tile, splits)
                # |error| <= c * eps32 * sqrt(K) * sum_k |a_k b_k| (+ one rounding of the result)
                bound = 2.0 ** -24 * (4.0 * max(1.0, K) ** 0.5) * mag + 2.0 ** -23 * ref.abs() + 1e-30
                ratio = float(((got - ref).abs() / bound).max())
                worst = max(worst, ratio)
                assert ratio <= 1.0, (path, mode, M, N, K, tile, splits, ratio)
                if mode == 2:
                    cs = A64.sum(0)
                    assert float((out[M * N:].double() - cs).abs().max()) <= 2e-5 * max(1.0, float(cs.abs().max())) * max(1.0, K / 64) ** 0.5
    print(f"[gemm {path} mode {mode}] worst error / float32 dot-product bound: {worst:.3f}")
    common.record_parity(f"test_gemm_kernel_family_on_ragged_shapes[{path},mode{mode}]", worst_error_over_float32_bound=worst)


def test_two_rank_shards_match_one_full_batch_handle():
    """The N>1 path through the HIP kernels, without a second GPU: two PpoAlg handles created with world_size = 2, each
    holding one env shard, against ONE handle holding all envs.  The collectives are done by hand exactly as
    pointfoot_amd.dp does them (sum of the 3 advantage scalars; sum of the flat gradient + KL tail), so this
    exercises gscale = 1 / world_size in ppo_minibatch_apply, the KL tail read by k_prepare_step and the global
    advantage normalisation.  Ranks must stay in lock-step bit for bit, and equal the full-batch run."""
    from pointfoot_amd.rsl_rl.modules import ActorCritic
    from pointfoot_amd.rsl_rl.algorithms import PPO
    n, T, O, P, A, hidden = 128, 8, 27, 148, 6, [128, 64, 32]
    dev = "cuda:0"

    def build(N, world, rank):
        torch.manual_seed(0)
        alg = PPO(ActorCritic(O, P, A, hidden, hidden, "elu", 1.0), device=dev, **ALG)
        alg.world_size, alg.rank = world, rank
        alg.init_storage(N, T, [O], [P], [A])
        assert alg._cfg.world_size == world
        return alg
    full, shards = build(2 * n, 1, 0), [build(n, 2, r) for r in range(2)]
    for s in shards:
        assert torch.equal(s.params, full.params)
    lib = full._lib
    gen = torch.Generator(device=dev).manual_seed(5)
    R = lambda *s: torch.randn(*s, device=dev, generator=gen)
    # a synthetic rollout written straight into the full storage; the shards get their env slices
    fs = full.storage
    fs.observations.copy_(R(T, 2 * n, O)); fs.privileged_observations.copy_(R(T, 2 * n, P))
    fs.actions.copy_(R(T, 2 * n, A)); fs.mu.copy_(fs.actions + 0.1 * R(T, 2 * n, A)); fs.sigma.fill_(1.0)
    fs.actions_log_prob.copy_((-0.5 * (fs.actions - fs.mu) ** 2 - 0.9189385).sum(-1, keepdim=True))
    fs.values.copy_(R(T, 2 * n, 1)); fs.rewards.copy_(R(T, 2 * n, 1))
    fs.dones.copy_((torch.rand(T, 2 * n, 1, device=dev, generator=gen) < 0.1).float())
    last = R(2 * n, P)
    names = ("observations", "privileged_observations", "actions", "mu", "sigma", "actions_log_prob", "values", "rewards", "dones")
    for r, s in enumerate(shards):
        for k in names:
            getattr(s.storage, k).copy_(getattr(fs, k)[:, r * n:(r + 1) * n])
    # returns + GLOBAL advantage normalisation (the 3-scalar all-reduce done by hand)
    common._abi.check(lib.ppo_compute_returns(full._h, last.data_ptr(), full._stream()))
    common._abi.check(lib.ppo_normalize_advantages(full._h, full._stream()))
    for r, s in enumerate(shards):
        common._abi.check(lib.ppo_compute_returns(s._h, last[r * n:(r + 1) * n].contiguous().data_ptr(), s._stream()))
    tot = shards[0].scalars[10:13] + shards[1].scalars[10:13]
    for s in shards:
        s.scalars[10:13] = tot
        common._abi.check(lib.ppo_normalize_advantages(s._h, s._stream()))
    for r, s in enumerate(shards):
        torch.testing.assert_close(s.storage.advantages, fs.advantages[:, r * n:(r + 1) * n], atol=2e-5, rtol=1e-5)
        torch.testing.assert_close(s.storage.returns, fs.returns[:, r * n:(r + 1) * n], atol=1e-6, rtol=1e-6)
    # minibatches: the full handle's minibatch k is the union of the two ranks' minibatch k
    B = n * T // 4
    local = [torch.randperm(n * T, device=dev, generator=gen) for _ in range(2)]
    to_full = lambda r, p: (p // n) * (2 * n) + r * n + (p % n)
    fperm = torch.cat([torch.cat([to_full(r, local[r][k * B:(k + 1) * B]) for r in range(2)]) for k in range(4)])
    assert torch.equal(torch.sort(fperm).values, torch.arange(2 * n * T, device=dev))
    full.perm.copy_(fperm.to(torch.int32))
    for r, s in enumerate(shards):
        s.perm.copy_(local[r].to(torch.int32))
    Pn = full.num_params
    for epoch in range(ALG["num_learning_epochs"]):
        for mb in range(4):
            common._abi.check(lib.ppo_minibatch_backward(full._h, mb, full._stream()))
            common._abi.check(lib.ppo_minibatch_apply(full._h, full._stream()))
            for s in shards:
                common._abi.check(lib.ppo_minibatch_backward(s._h, mb, s._stream()))
            gsum = shards[0].grads + shards[1].grads            # dist.all_reduce(SUM) over the flat gradient and its KL tail
            for s in shards:
                s.grads.copy_(gsum)
                common._abi.check(lib.ppo_minibatch_apply(s._h, s._stream()))
            torch.cuda.synchronize()
            assert torch.equal(shards[0].params, shards[1].params), "ranks left lock-step"
            assert torch.equal(shards[0].scalars[:3], shards[1].scalars[:3])       # lr, mean KL, grad norm
            assert abs(shards[0].scalars[1].item() - full.scalars[1].item()) <= 1e-6 + 1e-4 * abs(full.scalars[1].item())
    assert shards[0].scalars[0].item() == full.scalars[0].item(), "adaptive learning-rate decisions differ"
    assert shards[0].scalars[5].item() == full.scalars[5].item() == 20.0
    assert (shards[0].params - full.params).abs().max().item() <= 2e-5
    assert not torch.equal(full.params[:Pn], build(2 * n, 1, 0).params[:Pn])      # the update did move the parameters


def test_minibatch_size_is_floored_like_rsl_rl():
    """rsl_rl: mini_batch_size = batch_size // num_mini_batches (the remainder of the permutation is dropped); 7 envs x 5 steps
    in 4 minibatches -> 8 samples each, 3 unused."""
    N, T, O, P, A, hidden = 7, 5, 27, 27, 6, [64, 32]
    alg, ref = make(N, T, O, P, A, hidden)
    gen = torch.Generator(device="cuda:0").manual_seed(2)
    fill_rollout(alg, ref, N, T, O, P, A, gen)
    store = ref_store(alg)
    perm = torch.randperm(N * T, device="cuda:0", generator=gen)
    alg.perm.copy_(perm.to(torch.int32))
    common._abi.check(alg._lib.ppo_minibatch_backward(alg._h, 3, alg._stream()))
    B = (N * T) // 4
    assert B == 8
    mb = {k: v[perm[3 * B:4 * B].long()] for k, v in store.items()}
    loss, kl, _, _ = ppo_ref.minibatch_loss(ref, mb, ALG["clip_param"], ALG["value_loss_coef"], ALG["entropy_coef"])
    ref.zero_grad(); loss.backward()
    g_ref, g = ref.flat_grad(alg.param_offsets, alg.num_params), alg.grads[:alg.num_params]
    assert (g - g_ref).abs().max().item() <= 3e-4 * g_ref.abs().max().item() + 1e-7
    assert alg.grads[alg.num_params + 1].item() == 8.0          # sample count in the KL tail


@pytest.mark.parametrize("hidden", [[512, 256, 128], [64, 32]])
def test_backward_in_two_parts_equals_the_whole_pass(hidden):
    """Multi-GPU path: ppo_minibatch_backward_part(0) + (1) -- the pass cut where the EARLY gradient range is
    final, so that its all-reduce can run under the rest -- must leave bit-identical gradients to ppo_minibatch_backward,
    the two ranges of ppo_grad_groups must tile the buffer, and after part 0 the early range must already hold its final
    values while the late range (the two input layers) is still untouched."""
    N, T, O, P, A = 128, 8, 27, 148, 6
    alg, ref = make(N, T, O, P, A, hidden, seed=3)
    gen = torch.Generator(device="cuda:0").manual_seed(5)
    fill_rollout(alg, ref, N, T, O, P, A, gen)
    lib, h, s = alg._lib, alg._h, alg._stream()
    early, late = alg._grad_groups
    assert early.numel() + late.numel() == alg.grads.numel() and late.data_ptr() == alg.grads.data_ptr()
    assert early.data_ptr() == alg.grads.data_ptr() + 4 * late.numel()
    # the late range is exactly the two input layers (W, b of actor layer 0 and critic layer 0, 16-byte padded)
    assert late.numel() >= hidden[0] * (O + 1) + hidden[0] * (P + 1) and late.numel() <= hidden[0] * (O + 1) + hidden[0] * (P + 1) + 16
    alg.perm.copy_(torch.randperm(N * T, device="cuda:0", generator=gen).to(torch.int32))
    common._abi.check(lib.ppo_begin_update(h, s), lib)
    for mb in range(alg.num_mini_batches):
        SENT = 12345.0      # (the padding floats between tensors are never written: they keep the sentinel on both sides)
        alg.grads.fill_(SENT)
        common._abi.check(lib.ppo_minibatch_backward(h, mb, s), lib)
        whole = alg.grads.clone()
        assert torch.isfinite(whole).all() and int((whole != SENT).sum()) >= alg.num_params - 64
        alg.grads.fill_(SENT)
        common._abi.check(lib.ppo_minibatch_backward_part(h, mb, 0, s), lib)
        torch.cuda.synchronize()
        assert torch.equal(early, whole[late.numel():]), "early range not final after part 0"
        assert bool((late == SENT).all()), "part 0 wrote into the late range"
        # what the overlap relies on: part 1 never touches the EARLY range (its all-reduce is in flight by then -- a part 1 that
        # rewrote it, even with the same local values, would overwrite the reduced sums on N > 1 ranks)
        early.fill_(SENT)
        common._abi.check(lib.ppo_minibatch_backward_part(h, mb, 1, s), lib)
        torch.cuda.synchronize()
        assert bool((early == SENT).all()), "part 1 wrote into the early range"
        assert torch.equal(late, whole[:late.numel()])
    assert lib.ppo_minibatch_backward_part(h, 0, 2, s) != 0


@pytest.mark.parametrize("env", [{"PF_PPO_PAIRED": "0"}, {"PF_PPO_FUSED_HEADS": "0"}])
def test_update_under_collectives_without_the_paired_mode_keeps_the_blocking_path(env, monkeypatch):
    """A handle that cannot cut the backward pass (ppo_supports_parts == 0) must say so: PPO.update then runs the whole pass and
    one blocking all-reduce instead of failing inside ppo_minibatch_backward_part."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    N, T, O, P, A = 64, 4, 27, 148, 6
    alg, ref = make(N, T, O, P, A, [64, 32], seed=2)
    assert alg._lib.ppo_supports_parts(alg._h) == 0 and alg._grad_groups is None
    assert alg._lib.ppo_minibatch_backward_part(alg._h, 0, 0, alg._stream()) != 0      # the library still refuses the cut
    for k in env:
        monkeypatch.delenv(k)
    alg2, _ = make(N, T, O, P, A, [64, 32], seed=2)
    assert alg2._lib.ppo_supports_parts(alg2._h) == 1 and alg2._grad_groups is not None


@pytest.mark.parametrize("N,T", [(300, 6), (23000, 23)])      # the second: more than 512 x 1024 flags per rollout (1024-flag chunks)
def test_episode_ring_is_the_runners_deque_of_finished_episodes(N, T):
    """rsl_rl v1.0.2 OnPolicyRunner.learn: cur_reward_sum += rewards; cur_episode_length += 1; for the envs done on this step,
    in env order, rewbuffer / lenbuffer (deque(maxlen=100)) get the finished episode's return / length (call site
    legged_gym/scripts/train.py:44).  The device-side ring (k_process + k_episode_ring, no host sync) must hold exactly that
    deque after scripted episode ends -- sparse steps, a step with more than 100 ends at once, several rollouts."""
    from collections import deque
    from pointfoot_amd import dp
    O, P, A = 27, 148, 6
    alg, ref = make(N, T, O, P, A, [64, 32], seed=4)
    gen = torch.Generator(device="cuda:0").manual_seed(9)
    rewbuffer, lenbuffer = deque(maxlen=100), deque(maxlen=100)
    cur_r, cur_l = np.zeros(N, np.float32), np.zeros(N, np.float32)
    rates = [0.0, 0.002, 0.05, 0.6, 0.01, 0.0, 0.3, 0.003]
    step = 0
    totals = np.zeros(3)          # what the runner logs from: sum of finished returns, of their lengths, their count (PpoBuffers.d_ep_stats)
    for rollout in range(5 if N < 1000 else 2):
        for t in range(T):
            obs, cobs = torch.randn(N, O, device="cuda:0", generator=gen), torch.randn(N, P, device="cuda:0", generator=gen)
            alg.act(obs, cobs)
            rew = torch.randn(N, device="cuda:0", generator=gen)
            dones = torch.rand(N, device="cuda:0", generator=gen) < rates[step % len(rates)] * (1.0 if N < 1000 else 0.002)
            step += 1
            alg.process_env_step(rew, dones, {"time_outs": dones & (torch.rand(N, device="cuda:0", generator=gen) < 0.5)})
            cur_r += rew.cpu().numpy(); cur_l += 1
            ids = np.nonzero(dones.cpu().numpy())[0]
            rewbuffer.extend(cur_r[ids].tolist()); lenbuffer.extend(cur_l[ids].tolist())
            totals += np.array([cur_r[ids].sum(), cur_l[ids].sum(), len(ids)], np.float64)
            cur_r[ids] = 0; cur_l[ids] = 0
        alg.compute_returns(torch.randn(N, P, device="cuda:0", generator=gen))
        alg.storage.clear()
        rets, lens = dp.merge_episode_rings(alg.ep_ring)
        assert lens == list(lenbuffer), f"rollout {rollout}: episode lengths differ from the deque"
        np.testing.assert_allclose(rets, list(rewbuffer), rtol=1e-5, atol=1e-5)
        assert int(alg.ep_ring[3]) == step and int(alg.ep_ring[1]) == len(rewbuffer)
        np.testing.assert_allclose(alg.ep_stats[:3].cpu().numpy(), totals, rtol=1e-5, atol=1e-3)
    assert len(rewbuffer) == 100


def test_deferred_process_env_step_is_the_immediate_one():
    """ppo_process_env_step_deferred (the transition processing in the tail of the next act()'s head kernel, or flushed by
    compute_returns) against ppo_process_env_step on the same scripted rollouts: stored rewards (with the time-out bootstrap), dones,
    episode sums / lengths, the finished-episode ring and the GAE outputs are bit-identical.  The env outputs are persistent
    buffers rewritten in place once per step, as HipEnv.step hands them out."""
    N, T, O, P, A = 300, 6, 27, 148, 6
    algs = [make(N, T, O, P, A, [64, 32], seed=4)[0] for _ in range(2)]
    algs[1].env_outputs_persist = True
    gen = torch.Generator(device="cuda:0").manual_seed(11)
    bufs = [(torch.zeros(N, device="cuda:0"), torch.zeros(N, device="cuda:0", dtype=torch.bool), torch.zeros(N, device="cuda:0", dtype=torch.bool))
            for _ in range(2)]
    for rollout in range(3):
        for t in range(T):
            obs, cobs = torch.randn(N, O, device="cuda:0", generator=gen), torch.randn(N, P, device="cuda:0", generator=gen)
            noise = torch.randn(N, A, device="cuda:0", generator=gen)
            rew = torch.randn(N, device="cuda:0", generator=gen)
            dones = torch.rand(N, device="cuda:0", generator=gen) < (0.5 if t == 2 else 0.03)
            tos = dones & (torch.rand(N, device="cuda:0", generator=gen) < 0.5)
            for alg, (b_rew, b_done, b_to) in zip(algs, bufs):
                alg.inject_normal = noise
                alg.act(obs, cobs)
                b_rew.copy_(rew); b_done.copy_(dones); b_to.copy_(tos)       # "env.step": the buffers change only here
                alg.process_env_step(b_rew, b_done, {"time_outs": b_to})
        last = torch.randn(N, P, device="cuda:0", generator=gen)
        for alg in algs:
            alg.compute_returns(last)
        a, b = algs
        for name in ("rewards", "dones", "values", "returns", "advantages"):
            assert torch.equal(getattr(a.storage, name), getattr(b.storage, name)), name
        assert torch.equal(a.ep_ring, b.ep_ring) and torch.equal(a.ep_fin, b.ep_fin)
        assert torch.equal(a.ep_stats, b.ep_stats)       # summed once per rollout by k_episode_ring, in a fixed order
        for alg in algs:
            alg.storage.clear()
    assert int(algs[1].ep_ring[1]) == 100


def test_fused_optimiser_step_matches_the_separate_calls(monkeypatch):
    """ppo_minibatch_step (one process: the gradient norm's first stage inside the slab reduction, clip + adaptive-KL learning rate +
    Adam as one launch with a last-block ticket) against ppo_minibatch_backward + ppo_minibatch_apply: the same update up to the
    summation order of the squared norm -- same learning-rate decisions, parameters equal to ~1e-6, Adam step counter and the
    published scalars (norm, clip coefficient, KL) in place."""
    N, T, O, P, A, hidden = 128, 8, 27, 148, 6, [512, 256, 128]
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("PF_PPO_FUSED_STEP", mode)
        alg, ref = make(N, T, O, P, A, hidden, seed=21)
        gen = torch.Generator(device="cuda:0").manual_seed(22)
        fill_rollout(alg, ref, N, T, O, P, A, gen)
        torch.manual_seed(23)
        alg.update()
        torch.cuda.synchronize()
        out[mode] = (alg.params.clone(), alg.scalars.clone(), alg.adam_m.clone(), alg.adam_v.clone())
    pf, sf, mf, vf = out["1"]
    pu, su, mu, vu = out["0"]
    assert float(sf[0]) == float(su[0]), "adaptive learning rate took a different path"
    assert float(sf[5]) == float(su[5]) == 20.0 and float(sf[9]) == float(su[9])          # Adam steps / update count
    torch.testing.assert_close(sf[1:3], su[1:3], rtol=1e-5, atol=1e-7)                     # mean KL, gradient norm of the last minibatch
    torch.testing.assert_close(pf, pu, rtol=0, atol=2e-6)
    torch.testing.assert_close(mf, mu, rtol=1e-4, atol=1e-7)
    torch.testing.assert_close(vf, vu, rtol=1e-4, atol=1e-9)
    assert not torch.equal(pf, torch.zeros_like(pf))
