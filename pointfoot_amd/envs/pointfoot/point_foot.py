"""`PointFoot` — mirror of the reference env class legged_gym/envs/pointfoot/point_foot.py:16-1253 (PF_P441A biped).
All machinery lives in `HipEnv`; this class fixes the env kind: 27-dim proprioceptive obs + 27/148-dim privileged obs,
19-term reward stack, feet-state machine, force pushes, commands that never resample."""
from pointfoot_amd import _abi
from pointfoot_amd.envs.base.hip_env import HipEnv


class PointFoot(HipEnv):
    KIND = _abi.PF_KIND_POINTFOOT
