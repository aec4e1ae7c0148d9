"""legged_gym_dev_amd -- MI355X-native drop-in for the legged_gym hot path
(LeggedRobot.step()/post_physics_step() + the rsl_rl PPO rollout/update loop).

``LEGGED_GYM_ROOT_DIR`` / ``LEGGED_GYM_ENVS_DIR`` keep the names the reference exports
(legged_gym/__init__.py) so config strings such as
``"{LEGGED_GYM_ROOT_DIR}/resources/robots/anymal_c/urdf/anymal_c.urdf"`` still format.
"""
import os

LEGGED_GYM_ROOT_DIR = os.environ.get(
    "LEGGED_GYM_ROOT_DIR", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LEGGED_GYM_ENVS_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "envs")
