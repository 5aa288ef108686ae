"""`LeggedRobot` — mirror of the reference quadruped env legged_gym/envs/base/legged_robot.py:51-906 (+ BaseTask,
base_task.py:38-143): single 48(+187)-dim observation incl. base linear velocity, no privileged obs (the critic sees
the actor's obs), commands resampled every `resampling_time` and at reset, velocity pushes, the feet-air-time state
machine inside its reward, `only_positive_rewards`.  Differences to PointFoot: SURVEY.md Appendix E."""
from pointfoot_amd import _abi
from pointfoot_amd.envs.base.hip_env import HipEnv


class LeggedRobot(HipEnv):
    KIND = _abi.PF_KIND_LEGGED_ROBOT

    @property
    def num_observations(self):
        return self.num_obs
