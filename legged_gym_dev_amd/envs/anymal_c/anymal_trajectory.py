"""ANYmal (actuator network) on the trajectory-tracking env (reference: envs/anymal_c/anymal_trajectory.py:44-81)."""
from legged_gym_dev_amd.envs.base.legged_robot_trajectory import LeggedRobotTrajectory


class AnymalTrajectory(LeggedRobotTrajectory):
    pass
