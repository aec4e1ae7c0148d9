"""Cassie (reference: envs/cassie/cassie.py:42-46): adds the ``no_fly`` reward term, which the HIP
post-step kernel implements as LG_REW_NO_FLY; nothing else differs from LeggedRobot."""
from legged_gym_dev_amd.envs.base.legged_robot import LeggedRobot


class Cassie(LeggedRobot):
    pass
