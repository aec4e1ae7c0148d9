"""ANYmal-C on a flat plane (reference: envs/anymal_c/flat/anymal_c_flat_config.py:33-74).

Declared as a tree (envs/base/base_config.py: cfg_class / S): each S(...) becomes the nested section class a
hand-written ``class <section>(Base.<section>)`` would be, so tasks still override by subclassing.
"""
from legged_gym_dev_amd.envs.base.base_config import S, cfg_class
from legged_gym_dev_amd.envs.anymal_c.mixed_terrains.anymal_c_rough_config import AnymalCRoughCfg, AnymalCRoughCfgPPO


AnymalCFlatCfg = cfg_class("AnymalCFlatCfg", AnymalCRoughCfg, dict(
    env=S(
        num_observations=48,
    ),
    terrain=S(
        mesh_type='plane', measure_heights=False,
    ),
    asset=S(
        self_collisions=0,
    ),
    rewards=S(
        max_contact_force=350.0,
        scales=S(
            orientation=-5.0, torques=-2.5e-05, feet_air_time=2.0,
        ),
    ),
    commands=S(
        heading_command=False, resampling_time=4.0,
        ranges=S(
            ang_vel_yaw=[-1.5, 1.5],
        ),
    ),
    domain_rand=S(
        friction_range=[0.0, 1.5],
    ),
), doc=None, module=__name__)

AnymalCFlatCfgPPO = cfg_class("AnymalCFlatCfgPPO", AnymalCRoughCfgPPO, dict(
    policy=S(
        actor_hidden_dims=[128, 64, 32], critic_hidden_dims=[128, 64, 32], activation='elu',
    ),
    algorithm=S(
        entropy_coef=0.01,
    ),
    runner=S(
        run_name='', experiment_name='flat_anymal_c', load_run=-1, max_iterations=300,
    ),
), doc=None, module=__name__)
