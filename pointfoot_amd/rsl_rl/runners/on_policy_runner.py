"""OnPolicyRunner with the rsl_rl v1.0.2 surface used by the reference (task_registry.py:147-154, train.py:44,
play.py:58-64): learn / save / load / get_inference_policy, checkpoint keys model_state_dict /
optimizer_state_dict / iter / infos, `fps = T*N / (collection + learn)` as the headline metric.

Differences for the MI355X hot path: the rollout loop never synchronises with the host — episode statistics are
accumulated on the device and only read when an iteration is logged; under torch.distributed every rank runs its env
shard and rank 0 logs / checkpoints."""
import os
import statistics
import time
from collections import deque

import torch
import torch.distributed as dist

from pointfoot_amd import dp
from pointfoot_amd.rsl_rl.algorithms import PPO
from pointfoot_amd.rsl_rl.modules import ActorCritic


class OnPolicyRunner:
    def __init__(self, env, train_cfg, log_dir=None, device='cpu'):
        self.cfg = train_cfg["runner"]
        self.alg_cfg = train_cfg["algorithm"]
        self.policy_cfg = train_cfg["policy"]
        self.device = device
        self.env = env
        self.world_size = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        num_critic_obs = self.env.num_privileged_obs if self.env.num_privileged_obs is not None else self.env.num_obs
        if self.cfg["policy_class_name"] != "ActorCritic":
            raise NotImplementedError(f"policy class {self.cfg['policy_class_name']} (only the feed-forward ActorCritic is on the hot path)")
        if self.cfg["algorithm_class_name"] != "PPO":
            raise NotImplementedError(f"algorithm {self.cfg['algorithm_class_name']}")
        policy_kwargs = {k: v for k, v in self.policy_cfg.items() if k != "init_member_classes"}
        actor_critic = ActorCritic(self.env.num_obs, num_critic_obs, self.env.num_actions, **policy_kwargs).to(self.device)
        alg_kwargs = {k: v for k, v in self.alg_cfg.items() if k != "init_member_classes"}
        self.alg = PPO(actor_critic, device=self.device, seed=int(train_cfg.get("seed", 1)), **alg_kwargs)
        # an env whose step() hands out its own buffers (rew_buf / reset_buf / time_out_buf, rewritten by the next step only) lets
        # process_env_step ride in the next act()'s head kernel (ppo_process_env_step_deferred); PF_PPO_DEFER_PROCESS=0: own launch
        self.alg.env_outputs_persist = bool(getattr(env, "step_outputs_persist", False)) and os.environ.get("PF_PPO_DEFER_PROCESS", "1") == "1"
        self.num_steps_per_env = self.cfg["num_steps_per_env"]
        self.save_interval = self.cfg["save_interval"]
        self.alg.init_storage(self.env.num_envs, self.num_steps_per_env, [self.env.num_obs], [self.env.num_privileged_obs],
                              [self.env.num_actions])
        if hasattr(self.env, "ensure_stats_ring"):
            self.env.ensure_stats_ring(self.num_steps_per_env)     # lazily-read episode statistics must outlive one iteration
        self.log_dir = log_dir
        self.writer = None
        self.tot_timesteps = 0
        self.tot_time = 0
        self.current_learning_iteration = 0
        self.log_interval = int(self.cfg.get("log_interval", 1))
        N, dev = self.env.num_envs, self.device
        self._ep_stats = self.alg.ep_stats          # [sum return, sum length, count] of finished episodes since the last log
        # rsl_rl v1.0.2: the returns / lengths of the last 100 FINISHED EPISODES (Train/mean_reward, Train/mean_episode_length).
        # They are collected on the device (PpoBuffers.d_ep_ring, filled once per rollout without a host sync) and copied into
        # these deques when an iteration is logged.
        self.rewbuffer = deque(maxlen=100)
        self.lenbuffer = deque(maxlen=100)
        _, _ = self.env.reset()

    # ------------------------------------------------------------------------------------------------ training
    def rollout(self, obs, critic_obs):
        """num_steps_per_env env steps + GAE; no host synchronisation inside."""
        alg, env = self.alg, self.env
        ep_infos = []
        # The policy's heads, the sampling, the transition store and the deferred process_env_step ride in the env-step launch
        # (PPO.act_hidden + HipEnv.step_policy: one launch fewer per policy step, same bits -- tests/test_gpu_train.py); needs an env with
        # step_policy whose outputs persist.  PF_FUSED_POLICY_STEP=0: the two-call form  actions = alg.act(...); env.step(actions).
        fused = os.environ.get("PF_FUSED_POLICY_STEP", "1") == "1" and hasattr(env, "step_policy") and getattr(alg, "env_outputs_persist", False)
        for _ in range(self.num_steps_per_env):
            heads = alg.act_hidden(obs, critic_obs) if fused else None
            if heads is not None:
                obs, privileged_obs, rewards, dones, infos = env.step_policy(heads)
            else:
                actions = alg.act(obs, critic_obs)
                obs, privileged_obs, rewards, dones, infos = env.step(actions)
            critic_obs = privileged_obs if privileged_obs is not None else obs
            alg.process_env_step(rewards, dones, infos)
            if (self.log_dir is not None or dp.collectives_on()) and 'episode' in infos:
                ep_infos.append(infos['episode'])
            # (episode return / length bookkeeping happens inside ppo_process_env_step, on the device)
        alg.compute_returns(critic_obs)
        return obs, critic_obs, ep_infos

    def learn(self, num_learning_iterations, init_at_random_ep_len=False):
        if self.log_dir is not None and self.rank == 0:
            os.makedirs(self.log_dir, exist_ok=True)
        if self.log_dir is not None and self.writer is None and self.rank == 0:
            try:
                from torch.utils.tensorboard import SummaryWriter
                self.writer = SummaryWriter(log_dir=self.log_dir, flush_secs=10)
            except Exception:
                self.writer = None
        if init_at_random_ep_len:
            self.env.episode_length_buf = torch.randint_like(self.env.episode_length_buf, high=int(self.env.max_episode_length))
        obs = self.env.get_observations()
        privileged_obs = self.env.get_privileged_observations()
        critic_obs = privileged_obs if privileged_obs is not None else obs
        obs, critic_obs = obs.to(self.device), critic_obs.to(self.device)
        self.alg.actor_critic.train()
        tot_iter = self.current_learning_iteration + num_learning_iterations
        for it in range(self.current_learning_iteration, tot_iter):
            start = time.time()
            obs, critic_obs, ep_infos = self.rollout(obs, critic_obs)
            torch.cuda.synchronize(self.device)
            stop = time.time()
            collection_time = stop - start
            start = stop
            mean_value_loss, mean_surrogate_loss = self.alg.update()
            torch.cuda.synchronize(self.device)
            stop = time.time()
            learn_time = stop - start
            if it % self.log_interval == 0:
                # every rank takes part: episode statistics are reduced over all env shards (count-weighted), rank 0 prints
                ep_means = dp.reduce_episode_infos(ep_infos, self.device)
                ep_totals = dp.allreduce_sum(self._ep_stats.clone()).tolist()
                self._ep_stats.zero_()
                self.refresh_episode_buffers()
                if self.rank == 0:
                    self.log(locals())
            if self.log_dir is not None and self.rank == 0 and it % self.save_interval == 0:
                self.save(os.path.join(self.log_dir, 'model_{}.pt'.format(it)))
            ep_infos.clear()
        self.current_learning_iteration += num_learning_iterations
        if self.log_dir is not None and self.rank == 0:
            self.save(os.path.join(self.log_dir, 'model_{}.pt'.format(self.current_learning_iteration)))

    def refresh_episode_buffers(self):
        """rewbuffer / lenbuffer <- the last 100 finished episodes of ALL env shards (a collective under torch.distributed)"""
        rets, lens = dp.merge_episode_rings(self.alg.ep_ring, maxlen=self.rewbuffer.maxlen)
        self.rewbuffer.clear(); self.rewbuffer.extend(rets)
        self.lenbuffer.clear(); self.lenbuffer.extend(lens)

    def log(self, locs, width=80, pad=35):
        n_total = self.num_steps_per_env * self.env.num_envs * self.world_size
        self.tot_timesteps += n_total
        self.tot_time += locs['collection_time'] + locs['learn_time']
        iteration_time = locs['collection_time'] + locs['learn_time']
        ep_string = ''
        if locs['ep_means']:
            # steps on which no env was reset anywhere carry NaN (no fresh episode entry in the reference): mean over the others
            for key, value in locs['ep_means'].items():
                if self.writer:
                    self.writer.add_scalar('Episode/' + key, value, locs['it'])
                ep_string += f"""{f'Mean episode {key}:':>{pad}} {value:.4f}\n"""
        mean_std = float(self.alg.actor_critic.std.detach().mean())
        fps = int(n_total / iteration_time)
        vl, sl, lr = float(locs['mean_value_loss']), float(locs['mean_surrogate_loss']), float(self.alg.current_learning_rate)
        if self.writer:
            w = self.writer
            w.add_scalar('Loss/value_function', vl, locs['it'])
            w.add_scalar('Loss/surrogate', sl, locs['it'])
            w.add_scalar('Loss/learning_rate', lr, locs['it'])
            w.add_scalar('Policy/mean_noise_std', mean_std, locs['it'])
            w.add_scalar('Perf/total_fps', fps, locs['it'])
            w.add_scalar('Perf/collection time', locs['collection_time'], locs['it'])
            w.add_scalar('Perf/learning_time', locs['learn_time'], locs['it'])
            if len(self.rewbuffer) > 0:
                w.add_scalar('Train/mean_reward', statistics.mean(self.rewbuffer), locs['it'])
                w.add_scalar('Train/mean_episode_length', statistics.mean(self.lenbuffer), locs['it'])
                w.add_scalar('Train/mean_reward/time', statistics.mean(self.rewbuffer), self.tot_time)
                w.add_scalar('Train/mean_episode_length/time', statistics.mean(self.lenbuffer), self.tot_time)
        head = f" \033[1m Learning iteration {locs['it']}/{self.current_learning_iteration + locs['num_learning_iterations']} \033[0m "
        log_string = (f"""{'#' * width}\n"""
                      f"""{head.center(width, ' ')}\n\n"""
                      f"""{'Computation:':>{pad}} {fps:.0f} steps/s (collection: {locs['collection_time']:.3f}s, learning {locs['learn_time']:.3f}s)\n"""
                      f"""{'Value function loss:':>{pad}} {vl:.4f}\n"""
                      f"""{'Surrogate loss:':>{pad}} {sl:.4f}\n"""
                      f"""{'Mean action noise std:':>{pad}} {mean_std:.2f}\n""")
        if len(self.rewbuffer) > 0:
            log_string += (f"""{'Mean reward:':>{pad}} {statistics.mean(self.rewbuffer):.2f}\n"""
                           f"""{'Mean episode length:':>{pad}} {statistics.mean(self.lenbuffer):.2f}\n""")
        log_string += ep_string
        log_string += (f"""{'-' * width}\n"""
                       f"""{'Total timesteps:':>{pad}} {self.tot_timesteps}\n"""
                       f"""{'Iteration time:':>{pad}} {iteration_time:.2f}s\n"""
                       f"""{'Total time:':>{pad}} {self.tot_time:.2f}s\n""")
        print(log_string)

    # ------------------------------------------------------------------------------------------------ checkpoints
    def save(self, path, infos=None):
        torch.save({'model_state_dict': self.alg.actor_critic.state_dict(),
                    'optimizer_state_dict': self.alg.optimizer_state_dict(),
                    'iter': self.current_learning_iteration,
                    'infos': infos}, path)

    def load(self, path, load_optimizer=True):
        loaded = torch.load(path, map_location=self.device, weights_only=True)      # tensors and plain containers only: nothing from the file is executed
        self.alg.actor_critic.load_state_dict(loaded['model_state_dict'])      # copies into the flat-buffer views
        if load_optimizer:
            self.alg.load_optimizer_state_dict(loaded['optimizer_state_dict'])
        self.current_learning_iteration = loaded['iter']
        return loaded['infos']

    def get_inference_policy(self, device=None):
        self.alg.actor_critic.eval()
        if device is not None and torch.device(device) != torch.device(self.device):
            # the reference calls `.to(device)` here (play.py:58-59 passes env.device, the training device).  The parameters of this
            # ActorCritic are VIEWS of the flat buffer the kernels read: moving the module would silently detach them.
            same = torch.device(device).type == "cuda" and torch.device(self.device).type == "cuda" and \
                (torch.device(device).index or 0) == (torch.device(self.device).index or 0)
            if not same:
                raise RuntimeError(f"get_inference_policy(device={device!r}): the policy lives in the kernels' flat parameter buffer on "
                                   f"{self.device}; export it (helpers.export_policy_as_jit) or copy actor_critic.state_dict() to run it elsewhere")
        return self.alg.actor_critic.act_inference
