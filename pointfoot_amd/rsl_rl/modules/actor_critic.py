"""ActorCritic with the rsl_rl v1.0.2 surface: `actor` / `critic` are plain
`nn.Sequential[Linear, ELU, ..., Linear]` and `std` an `nn.Parameter`, so `state_dict()` keys, TorchScript / ONNX export
of `.actor` (helpers.py:180-190, export_policy_as_onnx.py) and checkpoints stay compatible.

MI355X twist: every parameter is a *view* into one flat fp32 buffer laid out as the HIP kernels expect
(actor W0,b0,..., critic W0,b0,..., std) — `flatten_into()` re-homes them — so the fused kernels, the optimiser and
the RCCL gradient all-reduce work on a single contiguous tensor with no per-step packing.
"""
import torch
import torch.nn as nn
from torch.distributions import Normal


def get_activation(act_name):
    table = {"elu": nn.ELU, "selu": nn.SELU, "relu": nn.ReLU, "lrelu": nn.LeakyReLU, "tanh": nn.Tanh, "sigmoid": nn.Sigmoid}
    if act_name == "crelu":
        raise NotImplementedError("crelu")
    if act_name not in table:
        print("invalid activation function!")
        return None
    return table[act_name]()


def _mlp(sizes, activation):
    layers = []
    for i in range(len(sizes) - 1):
        layers.append(nn.Linear(sizes[i], sizes[i + 1]))
        if i < len(sizes) - 2:
            layers.append(get_activation(activation))
    return nn.Sequential(*layers)


class ActorCritic(nn.Module):
    is_recurrent = False

    def __init__(self, num_actor_obs, num_critic_obs, num_actions, actor_hidden_dims=[256, 256, 256],
                 critic_hidden_dims=[256, 256, 256], activation='elu', init_noise_std=1.0, **kwargs):
        if kwargs:
            print("ActorCritic.__init__ got unexpected arguments, which will be ignored: " + str([key for key in kwargs.keys()]))
        super().__init__()
        self.activation_name = activation
        self.actor_dims = [num_actor_obs] + list(actor_hidden_dims) + [num_actions]
        self.critic_dims = [num_critic_obs] + list(critic_hidden_dims) + [1]
        self.actor = _mlp(self.actor_dims, activation)
        self.critic = _mlp(self.critic_dims, activation)
        print(f"Actor MLP: {self.actor}")
        print(f"Critic MLP: {self.critic}")
        self.std = nn.Parameter(init_noise_std * torch.ones(num_actions))
        self.distribution = None
        Normal.set_default_validate_args = False
        self.flat = None

    # ---- flat storage -------------------------------------------------------------------------------------
    def kernel_order(self):
        """parameters in the order of the flat buffer: actor (W,b)*, critic (W,b)*, std"""
        ps = []
        for net in (self.actor, self.critic):
            for m in net:
                if isinstance(m, nn.Linear):
                    ps += [m.weight, m.bias]
        return ps + [self.std]

    def flatten_into(self, flat, offsets=None):
        """Re-home every parameter as a view of `flat` (1-D float32 tensor on the target device).  `offsets`: start of
        each tensor in `kernel_order()` (from `ppo_param_offsets`: 16-byte aligned); None = packed."""
        off, end = 0, 0
        with torch.no_grad():
            for i, p in enumerate(self.kernel_order()):
                n = p.numel()
                if offsets is not None:
                    off = offsets[i]
                view = flat[off:off + n].view(p.shape)
                view.copy_(p.data.to(flat.device))
                p.data = view
                off += n
                end = max(end, off)
        self.flat = flat
        return end

    # ---- rsl_rl surface (torch path: used for export / inference on arbitrary devices) -----------------------
    def reset(self, dones=None):
        pass

    def forward(self):
        raise NotImplementedError

    @property
    def action_mean(self):
        return self.distribution.mean

    @property
    def action_std(self):
        return self.distribution.stddev

    @property
    def entropy(self):
        return self.distribution.entropy().sum(dim=-1)

    def update_distribution(self, observations):
        mean = self.actor(observations)
        self.distribution = Normal(mean, mean * 0. + self.std)

    def act(self, observations, **kwargs):
        self.update_distribution(observations)
        return self.distribution.sample()

    def get_actions_log_prob(self, actions):
        return self.distribution.log_prob(actions).sum(dim=-1)

    def act_inference(self, observations):
        return self.actor(observations)

    def evaluate(self, critic_observations, **kwargs):
        return self.critic(critic_observations)
