"""ANYmal-C tracking a reduced-order-model trajectory on a plane: the task the fork's authors train (reference:
envs/anymal_c/flat_trajectory/anymal_c_flat_trajectory_config.py:33-66; observation = 9 + 10 trajectory points x 2 + 36 = 65).

The fork's reward table for this task has no tracking term (only orientation / torques / feet_air_time / termination); the
table the authors launch with comes from their hydra file (deep_tube_learning/configs/rl/default.yaml:29-39) -- set
``cfg.rewards.scales.tracking_rom`` etc. to train something meaningful (tools/train_sanity.py --task anymal_c_flat_trajectory
does)."""
from legged_gym_dev_amd.envs.base.base_config import S, cfg_class
from legged_gym_dev_amd.envs.anymal_c.mixed_terrains_trajectory.anymal_c_rough_trajectory_config import (
    AnymalCRoughTrajectoryCfg, AnymalCRoughTrajectoryCfgPPO)

AnymalCFlatTrajectoryCfg = cfg_class("AnymalCFlatTrajectoryCfg", AnymalCRoughTrajectoryCfg, dict(
    env=S(num_observations=65),
    terrain=S(mesh_type='plane', measure_heights=False),
    asset=S(self_collisions=0),
    rewards=S(max_contact_force=350.0, scales=S(orientation=-5.0, torques=-2.5e-05, feet_air_time=0.5)),
    domain_rand=S(friction_range=[0.0, 1.5]),
), doc=None, module=__name__)

AnymalCFlatTrajectoryCfgPPO = cfg_class("AnymalCFlatTrajectoryCfgPPO", AnymalCRoughTrajectoryCfgPPO, dict(
    policy=S(actor_hidden_dims=[128, 64, 32], critic_hidden_dims=[128, 64, 32], activation='elu'),
    algorithm=S(entropy_coef=0.01),
    runner=S(run_name='', experiment_name='flat_anymal_c_trajectory', load_run=-1, max_iterations=300),
), doc=None, module=__name__)
