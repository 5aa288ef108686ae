"""Task registration (reference legged_gym/envs/__init__.py:46-52).  The pointfoot tasks are built; the other five
names of the reference registry (`anymal_c_rough`, `anymal_c_flat`, `anymal_b`, `a1`, `cassie`) belong to the
LeggedRobot/Anymal variant, which SURVEY.md §8f lists as *next* — they are not registered until that kernel exists, so
`--task anymal_c_rough` fails loudly with the reference's own "was not registered" ValueError instead of silently
running something else."""
from .pointfoot.point_foot import PointFoot
from .pointfoot.mixed_terrain.pointfoot_rough_config import PointFootRoughCfg, PointFootRoughCfgPPO
from .pointfoot.flat.pointfoot_flat_config import PointFootFlatCfg, PointFootFlatCfgPPO
from .base.legged_robot_config import LeggedRobotCfg, LeggedRobotCfgPPO

from pointfoot_amd.utils.task_registry import task_registry

task_registry.register("pointfoot_rough", PointFoot, PointFootRoughCfg(), PointFootRoughCfgPPO())
task_registry.register("pointfoot_flat", PointFoot, PointFootFlatCfg(), PointFootFlatCfgPPO())
