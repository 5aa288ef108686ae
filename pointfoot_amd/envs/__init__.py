"""Task registration (reference legged_gym/envs/__init__.py:46-52).  Built: the two pointfoot tasks and the ANYmal-C
tasks (LeggedRobot variant + actuator network).  `anymal_b`, `a1` and `cassie` (out of scope,
no kernel instance for their trees) are not registered, so asking for them raises the reference's own
"was not registered" ValueError instead of silently running something else."""
from .pointfoot.point_foot import PointFoot
from .pointfoot.mixed_terrain.pointfoot_rough_config import PointFootRoughCfg, PointFootRoughCfgPPO
from .pointfoot.flat.pointfoot_flat_config import PointFootFlatCfg, PointFootFlatCfgPPO
from .base.legged_robot_config import LeggedRobotCfg, LeggedRobotCfgPPO
from .base.legged_robot import LeggedRobot
from .anymal_c.anymal import Anymal
from .anymal_c.mixed_terrains.anymal_c_rough_config import AnymalCRoughCfg, AnymalCRoughCfgPPO
from .anymal_c.flat.anymal_c_flat_config import AnymalCFlatCfg, AnymalCFlatCfgPPO

from pointfoot_amd.utils.task_registry import task_registry

task_registry.register("anymal_c_rough", Anymal, AnymalCRoughCfg(), AnymalCRoughCfgPPO())
task_registry.register("anymal_c_flat", Anymal, AnymalCFlatCfg(), AnymalCFlatCfgPPO())
task_registry.register("pointfoot_rough", PointFoot, PointFootRoughCfg(), PointFootRoughCfgPPO())
task_registry.register("pointfoot_flat", PointFoot, PointFootFlatCfg(), PointFootFlatCfgPPO())
