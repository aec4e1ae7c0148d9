"""ANYmal-C tracking a reduced-order-model trajectory on mixed terrain (reference:
envs/anymal_c/mixed_terrains_trajectory/anymal_c_rough_trajectory_config.py:33-94)."""
from legged_gym_dev_amd.envs.base.base_config import S, cfg_class
from legged_gym_dev_amd.envs.base.legged_robot_trajectory_config import LeggedRobotTrajectoryCfg, LeggedRobotTrajectoryCfgPPO
from legged_gym_dev_amd.envs.anymal_c.mixed_terrains.anymal_c_rough_config import AnymalCRoughCfg

AnymalCRoughTrajectoryCfg = cfg_class("AnymalCRoughTrajectoryCfg", LeggedRobotTrajectoryCfg, dict(
    # num_observations: the fork's base class says 240 (legged_robot_trajectory_config.py:38, "Changes with RoM"), which is the width for a
    # 4-point trajectory window; with the committed generator (N = 10, :94) and the height scan of this terrain the observation
    # LeggedRobotTrajectory.compute_observations builds (:280-295) is 9 + 2*10 + 3*12 + 187 = 252 wide, and the reference's policy
    # construction fails on the mismatch.  Repaired here like the other fork defects.
    env=S(num_envs=4096, num_actions=12, num_observations=252),
    terrain=S(mesh_type='trimesh'),
    init_state=S(pos=[0.0, 0.0, 0.6], default_joint_angles=dict(AnymalCRoughCfg.init_state.default_joint_angles)),
    control=S(stiffness={'HAA': 80.0, 'HFE': 80.0, 'KFE': 80.0}, damping={'HAA': 2.0, 'HFE': 2.0, 'KFE': 2.0}, action_scale=0.5,
              decimation=4, use_actuator_network=True,
              actuator_net_file='{LEGGED_GYM_ROOT_DIR}/resources/actuator_nets/anydrive_v3_lstm.pt'),
    asset=S(file='{LEGGED_GYM_ROOT_DIR}/resources/robots/anymal_c/urdf/anymal_c.urdf', name='anymal_c', foot_name='FOOT',
            penalize_contacts_on=['SHANK', 'THIGH'], terminate_after_contacts_on=['base'], self_collisions=1),
    domain_rand=S(randomize_base_mass=True, added_mass_range=[-5.0, 5.0]),
    rewards=S(base_height_target=0.5, max_contact_force=500.0, only_positive_rewards=False, scales=S()),
), doc=None, module=__name__)

AnymalCRoughTrajectoryCfgPPO = cfg_class("AnymalCRoughTrajectoryCfgPPO", LeggedRobotTrajectoryCfgPPO, dict(
    runner=S(run_name='', experiment_name='rough_anymal_trajectory_c', load_run=-1),
), doc=None, module=__name__)
