"""Unitree A1 on mixed terrain, PD position control (reference: envs/a1/a1_config.py:33-84; task "a1" uses the
plain LeggedRobot class, envs/__init__.py:57).

Declared as a tree (envs/base/base_config.py: cfg_class / S): each S(...) becomes the nested section class a
hand-written ``class <section>(Base.<section>)`` would be, so tasks still override by subclassing.
"""
from legged_gym_dev_amd.envs.base.base_config import S, cfg_class
from legged_gym_dev_amd.envs.base.legged_robot_config import LeggedRobotCfg, LeggedRobotCfgPPO


A1RoughCfg = cfg_class("A1RoughCfg", LeggedRobotCfg, dict(
    init_state=S(
        pos=[0.0, 0.0, 0.42],
        default_joint_angles={'FL_hip_joint': 0.1, 'RL_hip_joint': 0.1, 'FR_hip_joint': -0.1, 'RR_hip_joint': -0.1, 'FL_thigh_joint': 0.8, 'FR_thigh_joint': 0.8, 'RL_thigh_joint': 1.0, 'RR_thigh_joint': 1.0, 'FL_calf_joint': -1.5, 'FR_calf_joint': -1.5, 'RL_calf_joint': -1.5, 'RR_calf_joint': -1.5},
    ),
    control=S(
        control_type='P', stiffness={'joint': 20.0}, damping={'joint': 0.5}, action_scale=0.25, decimation=4,
    ),
    asset=S(
        file='{LEGGED_GYM_ROOT_DIR}/resources/robots/a1/urdf/a1.urdf', name='a1', foot_name='foot',
        penalize_contacts_on=['thigh', 'calf'], terminate_after_contacts_on=['base'], self_collisions=1,
    ),
    rewards=S(
        soft_dof_pos_limit=0.9, base_height_target=0.25,
        scales=S(
            torques=-0.0002, dof_pos_limits=-10.0,
        ),
    ),
), doc=None, module=__name__)

A1RoughCfgPPO = cfg_class("A1RoughCfgPPO", LeggedRobotCfgPPO, dict(
    algorithm=S(
        entropy_coef=0.01,
    ),
    runner=S(
        run_name='', experiment_name='rough_a1',
    ),
), doc=None, module=__name__)
