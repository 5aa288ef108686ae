"""Rollout storage, time-major [T, N, .] like rsl_rl's RolloutStorage.  The tensors are filled and consumed by the HIP
kernels (ppo_act / ppo_process_env_step / ppo_compute_returns / ppo_minibatch_backward); this class owns the memory
and the step counter (including rsl_rl's "Rollout buffer overflow" error)."""
import torch


class RolloutStorage:
    class Transition:
        def __init__(self):
            self.observations = None
            self.critic_observations = None
            self.actions = None
            self.rewards = None
            self.dones = None
            self.values = None
            self.actions_log_prob = None
            self.action_mean = None
            self.action_sigma = None
            self.hidden_states = None

        def clear(self):
            self.__init__()

    def __init__(self, num_envs, num_transitions_per_env, obs_shape, privileged_obs_shape, actions_shape, device='cpu'):
        self.device = device
        self.obs_shape = obs_shape
        self.privileged_obs_shape = privileged_obs_shape
        self.actions_shape = actions_shape
        T, N = num_transitions_per_env, num_envs
        z = lambda *s: torch.zeros(*s, device=device, dtype=torch.float32)
        self.observations = z(T, N, *obs_shape)
        pshape = privileged_obs_shape if privileged_obs_shape[0] is not None else obs_shape
        self.privileged_observations = z(T, N, *pshape)
        self.rewards = z(T, N, 1)
        self.actions = z(T, N, *actions_shape)
        self.dones = z(T, N, 1)
        self.actions_log_prob = z(T, N, 1)
        self.values = z(T, N, 1)
        self.returns = z(T, N, 1)
        self.advantages = z(T, N, 1)
        self.mu = z(T, N, *actions_shape)
        self.sigma = z(T, N, *actions_shape)
        self.num_transitions_per_env = T
        self.num_envs = N
        self.step = 0

    def next_slot(self):
        if self.step >= self.num_transitions_per_env:
            raise AssertionError("Rollout buffer overflow")
        return self.step

    def clear(self):
        self.step = 0

    def get_statistics(self):
        done = self.dones
        done[-1] = 1
        flat_dones = done.permute(1, 0, 2).reshape(-1, 1)
        done_indices = torch.cat((flat_dones.new_tensor([-1], dtype=torch.int64), flat_dones.nonzero(as_tuple=False)[:, 0]))
        trajectory_lengths = (done_indices[1:] - done_indices[:-1])
        return trajectory_lengths.float().mean(), self.rewards.mean()
