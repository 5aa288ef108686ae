"""Config variants shared by the golden generator (applied to the REFERENCE's config instances) and the tests (applied to this
repo's mirror of them): same attribute names on both sides, so one function serves both.  Nothing here comes from the reference.

allrew   every `_reward_*` method the env class defines gets a non-zero scale (reference: 24 in point_foot.py:1132-1253, 19 in
         legged_robot.py:816-906), including the ones its configs ship with scale 0 or without an entry at all.  `feet_stumble`
         stays 0: the method is called `_reward_stumble`, so a non-zero `feet_stumble` makes the reference raise AttributeError
         (point_foot.py:814-815) -- pinned separately.
onlypos  rewards.only_positive_rewards = True (point_foot.py:312, legged_robot.py:201-202): the sum is clipped at 0 BEFORE the
         termination term is added.
"""

ALLREW_POINTFOOT = dict(dof_pos_limits=-2.0, dof_vel=-2.0e-4, termination=-3.0, dof_vel_limits=-1.5, stumble=-0.7)
ALLREW_LEGGED_ROBOT = dict(orientation=-0.5, dof_vel=-2.0e-4, base_height=-1.5, termination=-3.0, stand_still=-0.3,
                           dof_pos_limits=-2.0, dof_vel_limits=-1.5, torque_limits=-0.02, stumble=-0.7,
                           feet_contact_forces=-0.01)


def apply(cfg, variant):
    """`variant`: None or a string containing 'allrew' and / or 'onlypos'.  Sets attributes on the config INSTANCE (its nested
    classes were instantiated by BaseConfig.__init__), never on the classes."""
    if not variant:
        return cfg
    if "allrew" in variant:
        legged_robot = hasattr(cfg.env, "num_observations")
        for k, v in (ALLREW_LEGGED_ROBOT if legged_robot else ALLREW_POINTFOOT).items():
            setattr(cfg.rewards.scales, k, v)
        if legged_robot:
            # LeggedRobotCfg ships the soft limits at 1.0 (legged_robot_config.py:150-152): PD torques are clipped AT the limit, so
            # _reward_torque_limits could never be non-zero
            cfg.rewards.soft_torque_limit, cfg.rewards.soft_dof_vel_limit, cfg.rewards.soft_dof_pos_limit = 0.8, 0.9, 0.9
    if "onlypos" in variant:
        cfg.rewards.only_positive_rewards = True
    return cfg
