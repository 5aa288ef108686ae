"""`anymal_b` (reference legged_gym/envs/anymal_b/anymal_b_config.py:32-45): ANYmal-C task on the ANYmal-B asset."""
from pointfoot_amd.envs.anymal_c.mixed_terrains.anymal_c_rough_config import AnymalCRoughCfg, AnymalCRoughCfgPPO


class AnymalBRoughCfg(AnymalCRoughCfg):
    class asset(AnymalCRoughCfg.asset):
        file = '{LEGGED_GYM_ROOT_DIR}/resources/robots/anymal_b/urdf/anymal_b.urdf'
        name = "anymal_b"; foot_name = 'FOOT'

    class rewards(AnymalCRoughCfg.rewards):
        class scales(AnymalCRoughCfg.rewards.scales):
            pass


class AnymalBRoughCfgPPO(AnymalCRoughCfgPPO):
    class runner(AnymalCRoughCfgPPO.runner):
        run_name = ''; experiment_name = 'rough_anymal_b'; load_run = -1
