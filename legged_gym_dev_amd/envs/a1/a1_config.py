"""Unitree A1 on mixed terrain, PD position control (reference: envs/a1/a1_config.py:33-84; task "a1" uses the
plain LeggedRobot class, envs/__init__.py:57)."""
from legged_gym_dev_amd.envs.base.legged_robot_config import LeggedRobotCfg, LeggedRobotCfgPPO

_HIP, _THIGH_F, _THIGH_R, _CALF = 0.1, 0.8, 1.0, -1.5


class A1RoughCfg(LeggedRobotCfg):
    class init_state(LeggedRobotCfg.init_state):
        pos = [0.0, 0.0, 0.42]
        default_joint_angles = {           # target angles [rad] at zero action
            "FL_hip_joint": _HIP, "RL_hip_joint": _HIP, "FR_hip_joint": -_HIP, "RR_hip_joint": -_HIP,
            "FL_thigh_joint": _THIGH_F, "FR_thigh_joint": _THIGH_F, "RL_thigh_joint": _THIGH_R, "RR_thigh_joint": _THIGH_R,
            "FL_calf_joint": _CALF, "FR_calf_joint": _CALF, "RL_calf_joint": _CALF, "RR_calf_joint": _CALF,
        }

    class control(LeggedRobotCfg.control):
        control_type = "P"
        stiffness = {"joint": 20.0}
        damping = {"joint": 0.5}
        action_scale = 0.25
        decimation = 4

    class asset(LeggedRobotCfg.asset):
        file = "{LEGGED_GYM_ROOT_DIR}/resources/robots/a1/urdf/a1.urdf"
        name = "a1"
        foot_name = "foot"
        penalize_contacts_on = ["thigh", "calf"]
        terminate_after_contacts_on = ["base"]
        self_collisions = 1

    class rewards(LeggedRobotCfg.rewards):
        soft_dof_pos_limit = 0.9
        base_height_target = 0.25

        class scales(LeggedRobotCfg.rewards.scales):
            torques = -0.0002
            dof_pos_limits = -10.0


class A1RoughCfgPPO(LeggedRobotCfgPPO):
    class algorithm(LeggedRobotCfgPPO.algorithm):
        entropy_coef = 0.01

    class runner(LeggedRobotCfgPPO.runner):
        run_name = ""
        experiment_name = "rough_a1"
