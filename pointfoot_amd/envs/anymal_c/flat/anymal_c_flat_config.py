"""`anymal_c_flat` (reference legged_gym/envs/anymal_c/flat/anymal_c_flat_config.py:32-74)."""
from pointfoot_amd.envs.anymal_c.mixed_terrains.anymal_c_rough_config import AnymalCRoughCfg, AnymalCRoughCfgPPO


class AnymalCFlatCfg(AnymalCRoughCfg):
    class env(AnymalCRoughCfg.env):
        num_observations = 48

    class terrain(AnymalCRoughCfg.terrain):
        mesh_type = 'plane'; measure_heights = False

    class asset(AnymalCRoughCfg.asset):
        self_collisions = 0

    class rewards(AnymalCRoughCfg.rewards):
        max_contact_force = 350.

        class scales(AnymalCRoughCfg.rewards.scales):
            orientation = -5.0; torques = -0.000025; feet_air_time = 2.

    class commands(AnymalCRoughCfg.commands):
        heading_command = False; resampling_time = 4.

        class ranges(AnymalCRoughCfg.commands.ranges):
            ang_vel_yaw = [-1.5, 1.5]

    class domain_rand(AnymalCRoughCfg.domain_rand):
        friction_range = [0., 1.5]


class AnymalCFlatCfgPPO(AnymalCRoughCfgPPO):
    class policy(AnymalCRoughCfgPPO.policy):
        actor_hidden_dims = [128, 64, 32]; critic_hidden_dims = [128, 64, 32]; activation = 'elu'

    class algorithm(AnymalCRoughCfgPPO.algorithm):
        entropy_coef = 0.01

    class runner(AnymalCRoughCfgPPO.runner):
        run_name = ''; experiment_name = 'flat_anymal_c'; load_run = -1; max_iterations = 300
