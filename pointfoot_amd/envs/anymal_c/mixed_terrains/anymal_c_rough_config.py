"""`anymal_c_rough` (reference legged_gym/envs/anymal_c/mixed_terrains/anymal_c_rough_config.py:32-93)."""
from pointfoot_amd.envs.base.legged_robot_config import LeggedRobotCfg, LeggedRobotCfgPPO

_LEGS = ("LF", "LH", "RF", "RH")


class AnymalCRoughCfg(LeggedRobotCfg):
    class env(LeggedRobotCfg.env):
        num_envs = 4096; num_actions = 12

    class terrain(LeggedRobotCfg.terrain):
        mesh_type = 'trimesh'

    class init_state(LeggedRobotCfg.init_state):
        pos = [0.0, 0.0, 0.6]
        default_joint_angles = {**{f"{l}_HAA": (0.0 if l[0] == "L" else -0.0) for l in _LEGS},
                                **{f"{l}_HFE": (0.4 if l[1] == "F" else -0.4) for l in _LEGS},
                                **{f"{l}_KFE": (-0.8 if l[1] == "F" else 0.8) for l in _LEGS}}

    class control(LeggedRobotCfg.control):
        stiffness = {'HAA': 80., 'HFE': 80., 'KFE': 80.}
        damping = {'HAA': 2., 'HFE': 2., 'KFE': 2.}
        action_scale = 0.5; decimation = 4
        use_actuator_network = True
        actuator_net_file = "{LEGGED_GYM_ROOT_DIR}/resources/actuator_nets/anydrive_v3_lstm.pt"

    class asset(LeggedRobotCfg.asset):
        file = "{LEGGED_GYM_ROOT_DIR}/resources/robots/anymal_c/urdf/anymal_c.urdf"
        name = "anymal_c"; foot_name = "FOOT"
        penalize_contacts_on = ["SHANK", "THIGH"]; terminate_after_contacts_on = ["base"]
        self_collisions = 1

    class domain_rand(LeggedRobotCfg.domain_rand):
        randomize_base_mass = True; added_mass_range = [-5., 5.]

    class rewards(LeggedRobotCfg.rewards):
        base_height_target = 0.5; max_contact_force = 500.; only_positive_rewards = True

        class scales(LeggedRobotCfg.rewards.scales):
            pass


class AnymalCRoughCfgPPO(LeggedRobotCfgPPO):
    class runner(LeggedRobotCfgPPO.runner):
        run_name = ''; experiment_name = 'rough_anymal_c'; load_run = -1
