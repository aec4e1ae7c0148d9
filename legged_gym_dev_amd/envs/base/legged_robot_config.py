"""Default environment / PPO configuration (drop-in for the reference's LeggedRobotCfg /
LeggedRobotCfgPPO, legged_gym/envs/base/legged_robot_config.py:34-279).

Values are the reference defaults.  Two fork defects are repaired here instead of being
patched by every caller: ``domain_rand.max_push_vel`` exists (alias of
``max_push_vel_xy``) and ``curriculum.use_curriculum / curriculum_steps`` are real attributes.

Declared as a tree (envs/base/base_config.py: cfg_class / S): each S(...) becomes the nested section class a
hand-written ``class <section>(Base.<section>)`` would be, so tasks still override by subclassing.
"""
from .base_config import BaseConfig, S, cfg_class


LeggedRobotCfg = cfg_class("LeggedRobotCfg", BaseConfig, dict(
    env=S(
        num_envs=4096, num_observations=235, num_privileged_obs=None, num_actions=12, env_spacing=3.0,
        send_timeouts=True, episode_length_s=20,
    ),
    terrain=S(
        mesh_type='trimesh', horizontal_scale=0.1, vertical_scale=0.005, border_size=25, curriculum=True,
        static_friction=1.0, dynamic_friction=1.0, restitution=0.0, measure_heights=True,
        measured_points_x=[-0.8, -0.7, -0.6, -0.5, -0.4, -0.3, -0.2, -0.1, 0.0, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8],
        measured_points_y=[-0.5, -0.4, -0.3, -0.2, -0.1, 0.0, 0.1, 0.2, 0.3, 0.4, 0.5], selected=False,
        terrain_kwargs=None, max_init_terrain_level=5, terrain_length=8.0, terrain_width=8.0, num_rows=10,
        num_cols=20, terrain_proportions=[0.1, 0.1, 0.35, 0.25, 0.2], slope_treshold=0.75,
    ),
    commands=S(
        num_commands=4, resampling_time=10.0, heading_command=True,
        ranges=S(
            lin_vel_x=[-0.0, 0.0], lin_vel_y=[-0.0, 0.0], ang_vel_yaw=[0, 0], heading=[-0.0, 0.0],
        ),
    ),
    init_state=S(
        pos=[0.0, 0.0, 1.0], rot=[0.0, 0.0, 0.0, 1.0], lin_vel=[0.0, 0.0, 0.0], ang_vel=[0.0, 0.0, 0.0],
        default_joint_angles={'joint_a': 0.0, 'joint_b': 0.0},
    ),
    control=S(
        control_type='P', stiffness={'joint_a': 10.0, 'joint_b': 15.0}, damping={'joint_a': 1.0, 'joint_b': 1.5},
        action_scale=0.5, decimation=4,
    ),
    asset=S(
        file='', name='legged_robot', foot_name='None', penalize_contacts_on=[], terminate_after_contacts_on=[],
        disable_gravity=False, collapse_fixed_joints=True, fix_base_link=False, default_dof_drive_mode=3,
        self_collisions=0, replace_cylinder_with_capsule=True, flip_visual_attachments=True, density=0.001,
        angular_damping=0.0, linear_damping=0.0, max_angular_velocity=1000.0, max_linear_velocity=1000.0,
        armature=0.0, thickness=0.01,
    ),
    domain_rand=S(
        randomize_friction=True, friction_range=[0.5, 1.25], randomize_base_mass=False,
        added_mass_range=[-1.0, 1.0], randomize_inv_base_mass=False, inv_mass_range=[-1.0, 1.0], push_robots=True,
        push_interval_s=15, max_push_vel_xy=1.0, max_push_vel=1.0,
        rigid_shape_properties=S(
            randomize_restitution=False, restitution_range=[0.0, 1.0], randomize_compliance=False,
            compliance_range=[0.0, 1.0], randomize_thickness=False, thickness_range=[0.0, 0.05],
        ),
        dof_properties=S(
            randomize_stiffness=False, added_stiffness_range=[-50.0, 50.0], randomize_damping=False,
            added_damping_range=[-2.0, 2.0],
        ),
    ),
    rewards=S(
        scales=S(
            termination=-0.0,
        ),
        only_positive_rewards=True, tracking_sigma=0.25, soft_dof_pos_limit=1.0, soft_dof_vel_limit=1.0,
        soft_torque_limit=1.0, base_height_target=1.0, max_contact_force=100.0,
    ),
    curriculum=S(
        use_curriculum=False, curriculum_steps=[100, 200], commands=[0.5, 0.75, 1],
        push=S(
            magnitude=[0.1, 0.5, 1], time=[3, 2, 1],
        ),
    ),
    normalization=S(
        obs_scales=S(
            lin_vel=2.0, ang_vel=0.25, dof_pos=1.0, dof_vel=0.05, height_measurements=5.0,
        ),
        clip_observations=100.0, clip_actions=100.0,
    ),
    noise=S(
        add_noise=True, noise_level=1.0,
        noise_scales=S(
            dof_pos=0.01, dof_vel=1.5, lin_vel=0.1, ang_vel=0.2, gravity=0.05, height_measurements=0.1,
        ),
    ),
    viewer=S(
        ref_env=0, pos=[10, 0, 6], lookat=[11.0, 5, 3.0],
    ),
    sim=S(
        dt=0.005, substeps=1, gravity=[0.0, 0.0, -9.81], up_axis=1,
        physx=S(
            num_threads=10, solver_type=1, num_position_iterations=4, num_velocity_iterations=0,
            contact_offset=0.01, rest_offset=0.0, bounce_threshold_velocity=0.5, max_depenetration_velocity=1.0,
            max_gpu_contact_pairs=8388608, default_buffer_size_multiplier=5, contact_collection=2,
        ),
    ),
), doc=None, module=__name__)

LeggedRobotCfgPPO = cfg_class("LeggedRobotCfgPPO", BaseConfig, dict(
    seed=1,
    runner_class_name='OnPolicyRunner',
    policy=S(
        init_noise_std=1.0, actor_hidden_dims=[512, 256, 128], critic_hidden_dims=[512, 256, 128],
        activation='elu',
    ),
    algorithm=S(
        value_loss_coef=1.0, use_clipped_value_loss=True, clip_param=0.2, entropy_coef=0.01, num_learning_epochs=5,
        num_mini_batches=4, learning_rate=0.001, schedule='adaptive', gamma=0.99, lam=0.95, desired_kl=0.01,
        max_grad_norm=1.0,
    ),
    runner=S(
        policy_class_name='ActorCritic', algorithm_class_name='PPO', num_steps_per_env=24, max_iterations=1500,
        save_interval=50, experiment_name='test', run_name='', resume=False, load_run=-1, checkpoint=-1,
        resume_path=None,
    ),
), doc=None, module=__name__)
