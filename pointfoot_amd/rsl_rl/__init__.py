"""MI355X-native stand-in for the `rsl_rl` v1.0.2 package the reference depends on (setup.py:13, README.md:34-36):
same module layout and class names (`rsl_rl.runners.OnPolicyRunner`, `rsl_rl.algorithms.PPO`,
`rsl_rl.modules.ActorCritic`, `rsl_rl.storage.RolloutStorage`, `rsl_rl.env.VecEnv`), with the numerics on the HIP
kernels behind the `ppo_*` C ABI (include/pointfoot_hip.h)."""
