"""Cassie biped on mixed terrain (reference: envs/cassie/cassie_config.py:33-109).

Declared as a tree (envs/base/base_config.py: cfg_class / S): each S(...) becomes the nested section class a
hand-written ``class <section>(Base.<section>)`` would be, so tasks still override by subclassing.
"""
from legged_gym_dev_amd.envs.base.base_config import S, cfg_class
from legged_gym_dev_amd.envs.base.legged_robot_config import LeggedRobotCfg, LeggedRobotCfgPPO


CassieRoughCfg = cfg_class("CassieRoughCfg", LeggedRobotCfg, dict(
    env=S(
        num_envs=4096, num_observations=169, num_actions=12,
    ),
    terrain=S(
        measured_points_x=[-0.5, -0.4, -0.3, -0.2, -0.1, 0.0, 0.1, 0.2, 0.3, 0.4, 0.5],
        measured_points_y=[-0.5, -0.4, -0.3, -0.2, -0.1, 0.0, 0.1, 0.2, 0.3, 0.4, 0.5],
    ),
    init_state=S(
        pos=[0.0, 0.0, 1.0],
        default_joint_angles={'hip_abduction_left': 0.1, 'hip_rotation_left': 0.0, 'hip_flexion_left': 1.0, 'thigh_joint_left': -1.8, 'ankle_joint_left': 1.57, 'toe_joint_left': -1.57, 'hip_abduction_right': -0.1, 'hip_rotation_right': 0.0, 'hip_flexion_right': 1.0, 'thigh_joint_right': -1.8, 'ankle_joint_right': 1.57, 'toe_joint_right': -1.57},
    ),
    control=S(
        stiffness={'hip_abduction': 100.0, 'hip_rotation': 100.0, 'hip_flexion': 200.0, 'thigh_joint': 200.0, 'ankle_joint': 200.0, 'toe_joint': 40.0},
        damping={'hip_abduction': 3.0, 'hip_rotation': 3.0, 'hip_flexion': 6.0, 'thigh_joint': 6.0, 'ankle_joint': 6.0, 'toe_joint': 1.0},
        action_scale=0.5, decimation=4,
    ),
    asset=S(
        file='{LEGGED_GYM_ROOT_DIR}/resources/robots/cassie/urdf/cassie.urdf', name='cassie', foot_name='toe',
        terminate_after_contacts_on=['pelvis'], flip_visual_attachments=False, self_collisions=1,
    ),
    rewards=S(
        soft_dof_pos_limit=0.95, soft_dof_vel_limit=0.9, soft_torque_limit=0.9, max_contact_force=300.0,
        only_positive_rewards=False,
        scales=S(
            termination=-200.0, tracking_ang_vel=1.0, torques=-5e-06, dof_acc=-2e-07, lin_vel_z=-0.5,
            feet_air_time=5.0, dof_pos_limits=-1.0, no_fly=0.25, dof_vel=-0.0, ang_vel_xy=-0.0,
            feet_contact_forces=-0.0,
        ),
    ),
), doc=None, module=__name__)

CassieRoughCfgPPO = cfg_class("CassieRoughCfgPPO", LeggedRobotCfgPPO, dict(
    runner=S(
        run_name='', experiment_name='rough_cassie',
    ),
    algorithm=S(
        entropy_coef=0.01,
    ),
), doc=None, module=__name__)
