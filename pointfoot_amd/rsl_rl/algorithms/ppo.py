"""PPO with the rsl_rl v1.0.2 surface (act / process_env_step / compute_returns / update), executed by the HIP kernels
behind the ppo_* C ABI.  Multi-GPU: env-sharded data parallelism — every rank runs its own rollout and, per
minibatch, the flat gradient (+ the KL sum in its tail) is all-reduced over RCCL before the fused clip + Adam step,
so all ranks take identical parameter and learning-rate steps."""
import ctypes as C
import os

import torch
import torch.distributed as dist

from pointfoot_amd import _abi
from pointfoot_amd import dp
from pointfoot_amd.rsl_rl.storage import RolloutStorage

SC_LR, SC_KL, SC_GNORM, SC_VLOSS, SC_SLOSS, SC_STEP = 0, 1, 2, 3, 4, 5
SC_ADV_SUM = 10


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(None)


class PPO:
    def __init__(self, actor_critic, num_learning_epochs=1, num_mini_batches=1, clip_param=0.2, gamma=0.998, lam=0.95,
                 value_loss_coef=1.0, entropy_coef=0.0, learning_rate=1e-3, max_grad_norm=1.0, use_clipped_value_loss=True,
                 schedule="fixed", desired_kl=0.01, device='cpu', seed=1):
        if not str(device).startswith("cuda"):
            raise RuntimeError(f"PPO (pointfoot_amd) runs on the MI355X HIP kernels only; got device={device!r}")
        if actor_critic.activation_name != "elu":
            raise NotImplementedError("the fused MLP kernels implement ELU (the activation both pointfoot tasks use)")
        self.device = device
        self.desired_kl = desired_kl
        self.schedule = schedule
        self.learning_rate = learning_rate
        self.actor_critic = actor_critic
        self.storage = None
        self.transition = RolloutStorage.Transition()
        self.clip_param = clip_param
        self.num_learning_epochs = num_learning_epochs
        self.num_mini_batches = num_mini_batches
        self.value_loss_coef = value_loss_coef
        self.entropy_coef = entropy_coef
        self.gamma = gamma
        self.lam = lam
        self.max_grad_norm = max_grad_norm
        self.use_clipped_value_loss = use_clipped_value_loss
        self.seed = seed
        self.world_size = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self._lib = _abi.load()
        self._h = None
        self._sample_counter = 0
        self.inject_normal = None      # tests: (N, A) standard-normal draws replacing Philox
        self.env_outputs_persist = False      # see process_env_step; OnPolicyRunner sets it from env.step_outputs_persist

    # ------------------------------------------------------------------------------------------------ setup
    def init_storage(self, num_envs, num_transitions_per_env, actor_obs_shape, critic_obs_shape, action_shape):
        dev, lib = self.device, self._lib
        ac = self.actor_critic
        self.storage = st = RolloutStorage(num_envs, num_transitions_per_env, actor_obs_shape, critic_obs_shape, action_shape, dev)
        c = _abi.PpoConfig()
        c.num_envs, c.num_steps = num_envs, num_transitions_per_env
        c.num_obs, c.num_critic_obs, c.num_actions = ac.actor_dims[0], ac.critic_dims[0], ac.actor_dims[-1]
        c.n_actor_layers, c.n_critic_layers = len(ac.actor_dims) - 1, len(ac.critic_dims) - 1
        if max(c.n_actor_layers, c.n_critic_layers) > _abi.PPO_MAX_LAYERS:
            raise ValueError("at most 3 hidden layers")
        for i, d in enumerate(ac.actor_dims):
            c.actor_dims[i] = d
        for i, d in enumerate(ac.critic_dims):
            c.critic_dims[i] = d
        c.num_learning_epochs, c.num_mini_batches = self.num_learning_epochs, self.num_mini_batches
        c.use_clipped_value_loss = int(self.use_clipped_value_loss)
        c.adaptive_schedule = int(self.schedule == "adaptive")
        c.clip_param, c.value_loss_coef, c.entropy_coef = self.clip_param, self.value_loss_coef, self.entropy_coef
        c.gamma, c.lam, c.desired_kl = self.gamma, self.lam, (self.desired_kl if self.desired_kl is not None else 0.0)
        c.max_grad_norm, c.learning_rate = self.max_grad_norm, self.learning_rate
        c.adam_beta1, c.adam_beta2, c.adam_eps = 0.9, 0.999, 1e-8
        c.world_size = self.world_size
        c.seed_lo, c.seed_hi = (self.seed * 2654435761 + self.rank) & 0xFFFFFFFF, 0x85EBCA6B
        self._cfg = c
        P = lib.ppo_num_params(C.byref(c))
        if P < 0:
            raise RuntimeError("ppo: " + lib.pf_last_error().decode())
        self.num_params = P
        z = lambda n, dt=torch.float32: torch.zeros(n, device=dev, dtype=dt)
        self.params = z(P)
        ac.to(dev)
        offs = (C.c_int64 * 32)()
        nt = lib.ppo_param_offsets(C.byref(c), offs, 32)
        self.param_offsets = [int(offs[i]) for i in range(nt)]
        assert len(ac.kernel_order()) == nt and ac.flatten_into(self.params, self.param_offsets) <= P
        dp.sync_params(self.params)                   # identical initial policy on every rank
        self.grads = z(P + 4)                         # tail: [P] = KL sum, [P+1] = sample count
        self.adam_m, self.adam_v = z(P), z(P)
        self.scalars = z(16)
        self.scalars[SC_LR] = self.learning_rate
        ws = lib.ppo_workspace_floats(C.byref(c))
        self.workspace = z(ws)
        self.perm = z(num_envs * num_transitions_per_env, torch.int32)
        torch.cuda.set_device(torch.device(dev))        # kernels launch on the thread's current device (checked by the library)
        self._h = lib.ppo_create(C.byref(c), torch.device(dev).index or 0)
        if not self._h:
            raise RuntimeError("ppo_create: " + lib.pf_last_error().decode())
        b = _abi.PpoBuffers()
        b.d_params, b.d_grads, b.d_adam_m, b.d_adam_v = _ptr(self.params), _ptr(self.grads), _ptr(self.adam_m), _ptr(self.adam_v)
        b.d_obs, b.d_critic_obs, b.d_actions = _ptr(st.observations), _ptr(st.privileged_observations), _ptr(st.actions)
        b.d_rewards, b.d_dones, b.d_values = _ptr(st.rewards), _ptr(st.dones), _ptr(st.values)
        b.d_returns, b.d_advantages, b.d_log_prob = _ptr(st.returns), _ptr(st.advantages), _ptr(st.actions_log_prob)
        b.d_mu, b.d_sigma, b.d_perm = _ptr(st.mu), _ptr(st.sigma), _ptr(self.perm)
        b.d_workspace, b.workspace_floats, b.d_scalars = _ptr(self.workspace), ws, _ptr(self.scalars)
        self.ep_return, self.ep_length, self.ep_stats = z(num_envs), z(num_envs), z(4)
        b.d_ep_return, b.d_ep_length, b.d_ep_stats = _ptr(self.ep_return), _ptr(self.ep_length), _ptr(self.ep_stats)
        # the runner's deque(maxlen=100) of finished episodes, kept on the device (filled once per rollout by ppo_compute_returns)
        self.ep_fin = z(2 * num_transitions_per_env * num_envs)
        self.ep_ring = z(_abi.PPO_EP_RING_INTS, torch.int32)
        b.d_ep_fin, b.d_ep_ring = _ptr(self.ep_fin), _ptr(self.ep_ring)
        self._bufs = b
        _abi.check(lib.ppo_bind_buffers(self._h, C.byref(b)), lib)
        self._actions = torch.zeros(num_envs, c.num_actions, device=dev)
        # multi-GPU: the gradient buffer as two contiguous ranges in the order the backward pass finishes them
        rng = (C.c_int64 * 4)()
        self._grad_groups = None
        # the handle itself says whether it can cut the backward pass (paired execution; PF_PPO_PAIRED=0 / PF_PPO_FUSED_HEADS=0
        # switch it off): without it update() keeps ppo_minibatch_backward + one blocking all-reduce
        if lib.ppo_supports_parts(self._h) == 1 and lib.ppo_grad_groups(C.byref(c), rng) == 0:
            e0, en, l0, ln = (int(v) for v in rng)
            assert l0 == 0 and e0 == ln and e0 + en == P + 4
            self._grad_groups = (self.grads[e0:e0 + en], self.grads[l0:l0 + ln])

    def test_mode(self):
        self.actor_critic.eval()

    def train_mode(self):
        self.actor_critic.train()

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ------------------------------------------------------------------------------------------------ rollout
    def act(self, obs, critic_obs):
        t = self.storage.next_slot()
        obs = obs.contiguous()
        critic_obs = critic_obs.contiguous()
        self._sample_counter += 1
        _abi.check(self._lib.ppo_act(self._h, t, obs.data_ptr(), critic_obs.data_ptr(), self._actions.data_ptr(),
                                     self._sample_counter, _ptr(self.inject_normal), self._stream()), self._lib)
        self.transition.actions = self._actions
        return self._actions

    def act_hidden(self, obs, critic_obs):
        """PPO.act up to the last hidden layer, for an env that finishes the step itself (HipEnv.step_policy): returns the filled
        PfPolicyHeads -- heads, sampling, transition store and a pending deferred process_env_step then run inside the env-step launch --
        or None when this handle cannot hand its heads over (the caller falls back to act() + env.step())."""
        t = self.storage.next_slot()
        obs = obs.contiguous()
        critic_obs = critic_obs.contiguous()
        heads = _abi.PfPolicyHeads()
        rc = self._lib.ppo_act_hidden(self._h, t, obs.data_ptr(), critic_obs.data_ptr(), self._actions.data_ptr(), self._sample_counter + 1,
                                      _ptr(self.inject_normal), C.byref(heads), self._stream())
        if rc == 2:
            return None
        _abi.check(rc, self._lib)
        self._sample_counter += 1
        self._hidden_obs = (obs, critic_obs)      # the step launch reads them: keep them alive until it has been enqueued
        self.transition.actions = self._actions
        return heads

    def process_env_step(self, rewards, dones, infos):
        t = self.storage.next_slot()
        to = infos.get('time_outs', None) if isinstance(infos, dict) else None
        # env_outputs_persist (set by the runner for an env that says so): rewards / dones / time_outs are the env's own buffers, rewritten
        # only by its next step -- the processing then rides in the next act()'s head kernel instead of a launch of its own
        fn = self._lib.ppo_process_env_step_deferred if self.env_outputs_persist else self._lib.ppo_process_env_step
        _abi.check(fn(self._h, t, rewards.data_ptr(), dones.data_ptr(), _ptr(to), self._stream()), self._lib)
        self.storage.step += 1
        self.transition.clear()
        self.actor_critic.reset(dones)

    def compute_returns(self, last_critic_obs):
        lib = self._lib
        _abi.check(lib.ppo_compute_returns(self._h, last_critic_obs.contiguous().data_ptr(), self._stream()), lib)
        dp.allreduce_adv_stats(self.scalars[SC_ADV_SUM:SC_ADV_SUM + 3])   # global advantage statistics
        _abi.check(lib.ppo_normalize_advantages(self._h, self._stream()), lib)

    # ------------------------------------------------------------------------------------------------ update
    def update(self):
        lib, st = self._lib, self.storage
        n = st.num_envs * st.num_transitions_per_env
        # rsl_rl mini_batch_generator: one randperm per update, reused by every epoch
        self.perm.copy_(torch.randperm(n, device=self.device).to(torch.int32))
        self.scalars[SC_VLOSS:SC_SLOSS + 1] = 0.0
        _abi.check(lib.ppo_begin_update(self._h, self._stream()), lib)      # storage into permutation order, once for all epochs
        # PF_PPO_OVERLAP_ALLREDUCE=1: the backward pass in two parts with an asynchronous all-reduce behind each.  Default OFF since
        # round 4: on a one-rank RCCL group (every collective issued, nothing to move) the split costs +0.95 ms per update against
        # +0.13 ms for ONE blocking all-reduce per minibatch (profiles/r4_bench_rccl_single_rank*.json) -- four cross-stream hand-overs and
        # a second slab reduction per minibatch; what it could hide on 8 GPUs (20 x ~40 us of a 1.68 MB ring all-reduce) is no larger.
        # To be re-decided on a node with more than one GPU.
        overlap = dp.collectives_on() and self._grad_groups is not None and os.environ.get("PF_PPO_OVERLAP_ALLREDUCE", "0") == "1"
        # PF_PPO_FUSED_STEP=1 (one process only): backward + clip + adaptive LR + Adam as ppo_minibatch_step -- the norm's first stage
        # rides in the slab reduction, the rest is one launch with a last-block ticket: two launches fewer per minibatch.  Measured on
        # MI355X (profiles/r4_ppo_small_kernel_folds.txt): update 9.99 ms against 9.94 ms with the separate calls -- the launches it
        # removes were hidden in the queue, the block reductions it adds are not.  Off by default; kept as a tested switch.
        fused = not dp.collectives_on() and self.world_size == 1 and os.environ.get("PF_PPO_FUSED_STEP", "0") == "1"
        for _epoch in range(self.num_learning_epochs):
            for mb in range(self.num_mini_batches):
                if fused:
                    _abi.check(lib.ppo_minibatch_step(self._h, mb, self._stream()), lib)
                    continue
                if overlap:
                    # the backward pass in two parts: the all-reduce of the gradients finished by the first (every layer but the
                    # two input layers, std, the KL tail: 90 % of the 1.68 MB for pointfoot_rough) runs over xGMI while the last
                    # data-gradient GEMM and the input layers' weight gradients are still computing; the host waits for nothing,
                    # the stream waits for both collectives just before the optimiser step
                    early, late = self._grad_groups
                    _abi.check(lib.ppo_minibatch_backward_part(self._h, mb, 0, self._stream()), lib)
                    w0 = dp.allreduce_async(early)
                    _abi.check(lib.ppo_minibatch_backward_part(self._h, mb, 1, self._stream()), lib)
                    w1 = dp.allreduce_async(late)
                    for w in (w0, w1):
                        if w is not None:
                            w.wait()
                else:
                    _abi.check(lib.ppo_minibatch_backward(self._h, mb, self._stream()), lib)
                    dp.allreduce_grads(self.grads)           # RCCL over xGMI: 1.68 MB for pointfoot_rough
                _abi.check(lib.ppo_minibatch_apply(self._h, self._stream()), lib)
        num_updates = self.num_learning_epochs * self.num_mini_batches
        st.clear()
        # 0-d device tensors: no host sync here; the runner converts them when it logs
        return self.scalars[SC_VLOSS] / num_updates, self.scalars[SC_SLOSS] / num_updates

    @property
    def current_learning_rate(self):
        return self.scalars[SC_LR]

    # ------------------------------------------------------------------------------------------------ checkpoints
    def optimizer_state_dict(self):
        """torch.optim.Adam-compatible layout, parameters in `actor_critic.parameters()` order (std first)."""
        order = list(self.actor_critic.parameters())
        offs = {id(p): (o, p.numel(), p.shape) for p, o in zip(self.actor_critic.kernel_order(), self.param_offsets)}
        state = {}
        for i, p in enumerate(order):
            a, n, shp = offs[id(p)]
            state[i] = {"step": self.scalars[SC_STEP].detach().cpu().clone(),
                        "exp_avg": self.adam_m[a:a + n].view(shp).clone(), "exp_avg_sq": self.adam_v[a:a + n].view(shp).clone()}
        group = {"lr": float(self.scalars[SC_LR].item()), "betas": (0.9, 0.999), "eps": 1e-8, "weight_decay": 0, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "params": list(range(len(order)))}
        return {"state": state, "param_groups": [group]}

    def load_optimizer_state_dict(self, sd):
        order = list(self.actor_critic.parameters())
        offs = {id(p): (o, p.numel()) for p, o in zip(self.actor_critic.kernel_order(), self.param_offsets)}
        for i, p in enumerate(order):
            s = sd["state"].get(i)
            if s is None:
                continue
            a, n = offs[id(p)]
            self.adam_m[a:a + n].copy_(s["exp_avg"].reshape(-1))
            self.adam_v[a:a + n].copy_(s["exp_avg_sq"].reshape(-1))
            self.scalars[SC_STEP] = float(s["step"])
        if sd.get("param_groups"):
            self.scalars[SC_LR] = float(sd["param_groups"][0]["lr"])

    def __del__(self):
        try:
            if self._h:
                self._lib.ppo_destroy(self._h)
                self._h = None
        except Exception:
            pass
