"""`pointfoot_flat`: the rough task on a plane, 27-dim critic, smaller networks
(reference legged_gym/envs/pointfoot/flat/pointfoot_flat_config.py:4-46)."""
from pointfoot_amd.envs.pointfoot.mixed_terrain.pointfoot_rough_config import PointFootRoughCfg, PointFootRoughCfgPPO


class PointFootFlatCfg(PointFootRoughCfg):
    class env(PointFootRoughCfg.env):
        num_privileged_obs = 27

    class terrain(PointFootRoughCfg.terrain):
        mesh_type = 'plane'; measure_heights_critic = False

    class asset(PointFootRoughCfg.asset):
        self_collisions = 0

    class rewards(PointFootRoughCfg.rewards):
        max_contact_force = 350.

        class scales(PointFootRoughCfg.rewards.scales):
            orientation = -5.0; torques = -0.000025; feet_air_time = 5.; unbalance_feet_air_time = 1.0; no_fly = 1.

    class commands(PointFootRoughCfg.commands):
        num_commands = 3; heading_command = False; resampling_time = 4.

        class ranges(PointFootRoughCfg.commands.ranges):
            ang_vel_yaw = [-1.5, 1.5]

    class domain_rand(PointFootRoughCfg.domain_rand):
        friction_range = [0., 1.5]


class PointFootFlatCfgPPO(PointFootRoughCfgPPO):
    class policy(PointFootRoughCfgPPO.policy):
        actor_hidden_dims = [128, 64, 32]; critic_hidden_dims = [128, 64, 32]

    class runner(PointFootRoughCfgPPO.runner):
        experiment_name = 'pointfoot_flat'; max_iterations = 30000
