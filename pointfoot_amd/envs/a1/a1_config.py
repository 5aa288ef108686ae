"""`a1` (reference legged_gym/envs/a1/a1_config.py:32-79): Unitree A1 on the LeggedRobot variant, PD torques."""
from pointfoot_amd.envs.base.legged_robot_config import LeggedRobotCfg, LeggedRobotCfgPPO

_LEGS = ("FL", "RL", "FR", "RR")


class A1RoughCfg(LeggedRobotCfg):
    class init_state(LeggedRobotCfg.init_state):
        pos = [0.0, 0.0, 0.42]
        default_joint_angles = {**{f"{l}_hip_joint": (0.1 if l[1] == "L" else -0.1) for l in _LEGS},
                                **{f"{l}_thigh_joint": (0.8 if l[0] == "F" else 1.) for l in _LEGS},
                                **{f"{l}_calf_joint": -1.5 for l in _LEGS}}

    class control(LeggedRobotCfg.control):
        control_type = 'P'
        stiffness = {'joint': 20.}; damping = {'joint': 0.5}
        action_scale = 0.25; decimation = 4

    class asset(LeggedRobotCfg.asset):
        file = '{LEGGED_GYM_ROOT_DIR}/resources/robots/a1/urdf/a1.urdf'
        name = "a1"; foot_name = "foot"
        penalize_contacts_on = ["thigh", "calf"]; terminate_after_contacts_on = ["base"]
        self_collisions = 1

    class rewards(LeggedRobotCfg.rewards):
        soft_dof_pos_limit = 0.9; base_height_target = 0.25

        class scales(LeggedRobotCfg.rewards.scales):
            torques = -0.0002; dof_pos_limits = -10.0


class A1RoughCfgPPO(LeggedRobotCfgPPO):
    class algorithm(LeggedRobotCfgPPO.algorithm):
        entropy_coef = 0.01

    class runner(LeggedRobotCfgPPO.runner):
        run_name = ''; experiment_name = 'rough_a1'
