"""`pointfoot_rough` task and PPO tables (PF_P441A biped on the curriculum terrain).

Same names/values as reference legged_gym/envs/pointfoot/mixed_terrain/pointfoot_rough_config.py:3-259
(the API surface train.py / play.py users subclass); compact table form.
"""
from pointfoot_amd.envs.base.base_config import BaseConfig

_JOINTS = ("abad_L_Joint", "hip_L_Joint", "knee_L_Joint", "foot_L_Joint",
           "abad_R_Joint", "hip_R_Joint", "knee_R_Joint", "foot_R_Joint")


def _per_joint(actuated, passive=0.0):
    return {j: (passive if j.startswith("foot") else actuated) for j in _JOINTS}


class PointFootRoughCfg(BaseConfig):
    class env:
        num_envs = 1; num_propriceptive_obs = 27; num_privileged_obs = 148; num_actions = 6
        env_spacing = 3.; send_timeouts = True; episode_length_s = 20

    class terrain:
        mesh_type = 'trimesh'
        horizontal_scale = 0.1; vertical_scale = 0.005; border_size = 25
        curriculum = True
        static_friction = 0.4; dynamic_friction = 0.6; restitution = 0.8
        measure_heights_actor = False; measure_heights_critic = True
        measured_points_x = [-0.5, -0.4, -0.3, -0.2, -0.1, 0., 0.1, 0.2, 0.3, 0.4, 0.5]
        measured_points_y = [-0.5, -0.4, -0.3, -0.2, -0.1, 0., 0.1, 0.2, 0.3, 0.4, 0.5]
        selected = False; terrain_kwargs = None
        max_init_terrain_level = 5
        terrain_length = 8.; terrain_width = 8.; num_rows = 10; num_cols = 20
        terrain_proportions = [0.1, 0.1, 0.35, 0.25, 0.2]
        slope_treshold = 0.75

    class commands:
        curriculum = False; max_curriculum = 1.; num_commands = 4
        resampling_time = 10.; heading_command = True

        class ranges:
            lin_vel_x = [-1.0, 1.0]; lin_vel_y = [-0.2, 0.2]; ang_vel_yaw = [-1, 1]; heading = [-3.14, 3.14]

    class init_state:
        pos = [0.0, 0.0, 0.62]; rot = [0.0, 0.0, 0.0, 1.0]; lin_vel = [0.0, 0.0, 0.0]; ang_vel = [0.0, 0.0, 0.0]
        default_joint_angles = _per_joint(0.0)

    class control:
        control_type = 'P'
        stiffness = _per_joint(40)          # N*m/rad
        damping = _per_joint(1.5)           # N*m*s/rad
        action_scale = 0.5; decimation = 4

    class asset:
        file = '{LEGGED_GYM_ROOT_DIR}/resources/robots/PF_P441A/urdf/PF_P441A.urdf'
        name = "PF_P441A"; foot_name = 'foot'
        terminate_after_contacts_on = ["abad", "base"]
        penalize_contacts_on = ["base", "abad", "hip", "knee"]
        disable_gravity = False; collapse_fixed_joints = True; fix_base_link = False
        default_dof_drive_mode = 3; self_collisions = 0
        replace_cylinder_with_capsule = True; flip_visual_attachments = False
        density = 0.001; angular_damping = 0.; linear_damping = 0.
        max_angular_velocity = 1000.; max_linear_velocity = 1000.; armature = 0.; thickness = 0.01

    class domain_rand:
        randomize_friction = True; friction_range = [0.0, 1.6]
        randomize_base_mass = True; added_mass_range = [-1., 2.]
        randomize_base_com = True; rand_com_vec = [0.03, 0.02, 0.03]
        push_robots = True; push_interval_s = 7; max_push_vel_xy = 1.

    class rewards:
        class scales:
            action_rate = -0.01; ang_vel_xy = -0.05; base_height = -10.0; collision = -50.0; dof_acc = -2.5e-07
            dof_pos_limits = -0.0; dof_vel = -0.0; feet_air_time = 60; feet_contact_forces = -0.01
            feet_stumble = -0.0; lin_vel_z = -0.5; no_fly = 1.0; orientation = -5.0; stand_still = -1.0
            termination = -0.0; torque_limits = -0.1; torques = -2.5e-05; tracking_ang_vel = 5
            tracking_lin_vel = 10.0; unbalance_feet_air_time = -300.0; unbalance_feet_height = -60.0
            feet_distance = -100; survival = 100

        base_height_target = 0.62
        soft_dof_pos_limit = 0.95; soft_dof_vel_limit = 0.9; soft_torque_limit = 0.8
        max_contact_force = 200.
        only_positive_rewards = False
        min_feet_distance = 0.1; min_feet_air_time = 0.25; max_feet_air_time = 0.65
        tracking_sigma = 0.25

    class normalization:
        class obs_scales:
            lin_vel = 2.0; ang_vel = 0.25; dof_pos = 1.0; dof_vel = 0.05; height_measurements = 5.0

        clip_observations = 100.; clip_actions = 100.

    class noise:
        add_noise = True; noise_level = 1.0

        class noise_scales:
            dof_pos = 0.01; dof_vel = 1.5; lin_vel = 0.1; ang_vel = 0.2; gravity = 0.05; height_measurements = 0.1

    class viewer:
        ref_env = 0; pos = [10, 0, 6]; lookat = [11., 5, 3.]

    class sim:
        dt = 0.005; substeps = 1; gravity = [0., 0., -9.81]; up_axis = 1

        class physx:
            num_threads = 10; solver_type = 1
            num_position_iterations = 4; num_velocity_iterations = 0
            contact_offset = 0.01; rest_offset = 0.0
            bounce_threshold_velocity = 0.5; max_depenetration_velocity = 1.0
            max_gpu_contact_pairs = 2 ** 23; default_buffer_size_multiplier = 5; contact_collection = 2


class PointFootRoughCfgPPO(BaseConfig):
    seed = 1
    runner_class_name = 'OnPolicyRunner'

    class policy:
        init_noise_std = 1.0
        actor_hidden_dims = [512, 256, 128]; critic_hidden_dims = [512, 256, 128]
        activation = 'elu'

    class algorithm:
        value_loss_coef = 1.0; use_clipped_value_loss = True; clip_param = 0.2; entropy_coef = 0.01
        num_learning_epochs = 5; num_mini_batches = 4
        learning_rate = 1.e-3; schedule = 'adaptive'
        gamma = 0.99; lam = 0.95; desired_kl = 0.01; max_grad_norm = 1.

    class runner:
        policy_class_name = 'ActorCritic'; algorithm_class_name = 'PPO'
        num_steps_per_env = 24; max_iterations = 100000
        save_interval = 50; experiment_name = 'pointfoot_rough'; run_name = ''
        resume = False; load_run = -1; checkpoint = -1; resume_path = None
