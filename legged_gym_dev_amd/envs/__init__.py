"""Task registrations (reference: legged_gym/envs/__init__.py:53-59): the hot-path tasks, the trajectory-tracking variants (:55-56) and the A1 / ANYmal-B robots."""
from legged_gym_dev_amd.utils.task_registry import task_registry
from .base.legged_robot import LeggedRobot
from .base.legged_robot_trajectory import LeggedRobotTrajectory
from .anymal_c.anymal import Anymal
from .anymal_c.anymal_trajectory import AnymalTrajectory
from .anymal_c.mixed_terrains_trajectory.anymal_c_rough_trajectory_config import AnymalCRoughTrajectoryCfg, AnymalCRoughTrajectoryCfgPPO
from .anymal_c.flat_trajectory.anymal_c_flat_trajectory_config import AnymalCFlatTrajectoryCfg, AnymalCFlatTrajectoryCfgPPO
from .anymal_c.mixed_terrains.anymal_c_rough_config import AnymalCRoughCfg, AnymalCRoughCfgPPO
from .anymal_c.flat.anymal_c_flat_config import AnymalCFlatCfg, AnymalCFlatCfgPPO
from .cassie.cassie import Cassie
from .cassie.cassie_config import CassieRoughCfg, CassieRoughCfgPPO
from .a1.a1_config import A1RoughCfg, A1RoughCfgPPO
from .anymal_b.anymal_b_config import AnymalBRoughCfg, AnymalBRoughCfgPPO

task_registry.register("anymal_c_rough", Anymal, AnymalCRoughCfg(), AnymalCRoughCfgPPO())
task_registry.register("anymal_c_flat", Anymal, AnymalCFlatCfg(), AnymalCFlatCfgPPO())
task_registry.register("anymal_c_rough_trajectory", AnymalTrajectory, AnymalCRoughTrajectoryCfg(), AnymalCRoughTrajectoryCfgPPO())
task_registry.register("anymal_c_flat_trajectory", AnymalTrajectory, AnymalCFlatTrajectoryCfg(), AnymalCFlatTrajectoryCfgPPO())
task_registry.register("cassie", Cassie, CassieRoughCfg(), CassieRoughCfgPPO())
task_registry.register("anymal_b", Anymal, AnymalBRoughCfg(), AnymalBRoughCfgPPO())
task_registry.register("a1", LeggedRobot, A1RoughCfg(), A1RoughCfgPPO())
