"""TEST INFRASTRUCTURE — plain-PyTorch fp32 restatement of rsl_rl v1.0.2's PPO (ActorCritic, RolloutStorage GAE,
PPO.update) as described in SURVEY.md Appendix B.  PARITY UNPINNED: `rsl_rl` is an un-vendored dependency of the
reference (setup.py:13; call sites legged_gym/utils/task_registry.py:147, scripts/train.py:44) and is absent from
/root/reference and from this image, so this file restates the published algorithm and serves as the autograd
reference the HIP PPO kernels are checked against.  Only tests/, smoke() and bench.py's cpu_baseline leg import it.
"""
import torch
import torch.nn as nn
from torch.distributions import Normal


def mlp(sizes):
    layers = []
    for i in range(len(sizes) - 1):
        layers.append(nn.Linear(sizes[i], sizes[i + 1]))
        if i < len(sizes) - 2:
            layers.append(nn.ELU())
    return nn.Sequential(*layers)


class RefActorCritic(nn.Module):
    def __init__(self, num_obs, num_critic_obs, num_actions, actor_hidden, critic_hidden, init_noise_std=1.0):
        super().__init__()
        self.actor = mlp([num_obs] + list(actor_hidden) + [num_actions])
        self.critic = mlp([num_critic_obs] + list(critic_hidden) + [1])
        self.std = nn.Parameter(init_noise_std * torch.ones(num_actions))

    def kernel_order(self):
        ps = []
        for net in (self.actor, self.critic):
            for m in net:
                if isinstance(m, nn.Linear):
                    ps += [m.weight, m.bias]
        return ps + [self.std]

    def _spans(self, offsets):
        """(tensor, start) in flat order; offsets=None: packed, else the starts reported by ppo_param_offsets"""
        off, out = 0, []
        for i, p in enumerate(self.kernel_order()):
            if offsets is not None:
                off = offsets[i]
            out.append((p, off))
            off += p.numel()
        return out

    def load_flat(self, flat, offsets=None):
        with torch.no_grad():
            for p, off in self._spans(offsets):
                p.copy_(flat[off:off + p.numel()].view(p.shape))

    def _gather(self, get, offsets, total):
        if offsets is None:
            return torch.cat([get(p).reshape(-1) for p in self.kernel_order()])
        out = torch.zeros(total, dtype=self.std.dtype, device=self.std.device)
        for p, off in self._spans(offsets):
            out[off:off + p.numel()] = get(p).reshape(-1)
        return out

    def flat(self, offsets=None, total=None):
        return self._gather(lambda p: p.detach(), offsets, total)

    def flat_grad(self, offsets=None, total=None):
        return self._gather(lambda p: p.grad, offsets, total)


def act(ac, obs, critic_obs, normal_noise):
    """PPO.act with the sample expressed as mean + std * noise."""
    with torch.no_grad():
        mean = ac.actor(obs)
        dist = Normal(mean, mean * 0. + ac.std)
        actions = mean + ac.std * normal_noise
        return dict(actions=actions, values=ac.critic(critic_obs), log_prob=dist.log_prob(actions).sum(-1),
                    mu=mean, sigma=dist.stddev)


def process_rewards(rewards, values, time_outs, gamma):
    """PPO.process_env_step bootstrapping: rewards += gamma * squeeze(values * time_outs)"""
    return rewards + gamma * (values.squeeze(-1) * time_outs.float())


def compute_returns(values, rewards, dones, last_values, gamma, lam):
    """RolloutStorage.compute_returns; tensors [T, N]."""
    T = values.shape[0]
    returns = torch.zeros_like(values)
    adv = torch.zeros_like(last_values)
    for t in reversed(range(T)):
        nv = last_values if t == T - 1 else values[t + 1]
        nnt = 1.0 - dones[t]
        delta = rewards[t] + nnt * gamma * nv - values[t]
        adv = delta + nnt * gamma * lam * adv
        returns[t] = adv + values[t]
    advantages = returns - values
    raw = advantages.clone()
    advantages = (advantages - advantages.mean()) / (advantages.std() + 1e-8)
    return returns, advantages, raw


def minibatch_loss(ac, mb, clip, vcoef, ecoef, clipped_value=True):
    mean = ac.actor(mb["obs"])
    dist = Normal(mean, mean * 0. + ac.std)
    logp = dist.log_prob(mb["actions"]).sum(-1)
    value = ac.critic(mb["critic_obs"]).squeeze(-1)
    entropy = dist.entropy().sum(-1)
    sigma = dist.stddev
    with torch.inference_mode():
        kl = torch.sum(torch.log(sigma / mb["old_sigma"] + 1.e-5) +
                       (torch.square(mb["old_sigma"]) + torch.square(mb["old_mu"] - mean)) / (2.0 * torch.square(sigma)) - 0.5, dim=-1)
        kl_mean = torch.mean(kl)
    ratio = torch.exp(logp - mb["old_log_prob"])
    surrogate = -mb["advantages"] * ratio
    surrogate_clipped = -mb["advantages"] * torch.clamp(ratio, 1.0 - clip, 1.0 + clip)
    surrogate_loss = torch.max(surrogate, surrogate_clipped).mean()
    if clipped_value:
        value_clipped = mb["values"] + (value - mb["values"]).clamp(-clip, clip)
        value_loss = torch.max((value - mb["returns"]).pow(2), (value_clipped - mb["returns"]).pow(2)).mean()
    else:
        value_loss = (mb["returns"] - value).pow(2).mean()
    loss = surrogate_loss + vcoef * value_loss - ecoef * entropy.mean()
    return loss, kl_mean, value_loss, surrogate_loss


def update(ac, optimizer, store, perm, cfg, lr):
    """PPO.update: `num_learning_epochs` passes over `num_mini_batches` slices of one permutation.  `store` holds the
    flattened [T*N, .] tensors; returns the final learning rate and the loss means."""
    B = perm.numel() // cfg["num_mini_batches"]
    mv = ms = 0.0
    for _ in range(cfg["num_learning_epochs"]):
        for i in range(cfg["num_mini_batches"]):
            idx = perm[i * B:(i + 1) * B].long()
            mb = {k: v[idx] for k, v in store.items()}
            loss, kl_mean, vl, sl = minibatch_loss(ac, mb, cfg["clip_param"], cfg["value_loss_coef"], cfg["entropy_coef"],
                                                   cfg["use_clipped_value_loss"])
            if cfg["desired_kl"] is not None and cfg["schedule"] == "adaptive":
                if kl_mean > cfg["desired_kl"] * 2.0:
                    lr = max(1e-5, lr / 1.5)
                elif kl_mean < cfg["desired_kl"] / 2.0 and kl_mean > 0.0:
                    lr = min(1e-2, lr * 1.5)
                for g in optimizer.param_groups:
                    g["lr"] = lr
            optimizer.zero_grad()
            loss.backward()
            nn.utils.clip_grad_norm_(ac.parameters(), cfg["max_grad_norm"])
            optimizer.step()
            mv += vl.item()
            ms += sl.item()
    n = cfg["num_learning_epochs"] * cfg["num_mini_batches"]
    return lr, mv / n, ms / n
