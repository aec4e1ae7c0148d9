"""ANYmal-B: the ANYmal-C rough-terrain task on the B model (reference: envs/anymal_b/anymal_b_config.py:33-45;
task "anymal_b" uses the Anymal class, envs/__init__.py:56).

Declared as a tree (envs/base/base_config.py: cfg_class / S): each S(...) becomes the nested section class a
hand-written ``class <section>(Base.<section>)`` would be, so tasks still override by subclassing.
"""
from legged_gym_dev_amd.envs.base.base_config import S, cfg_class
from legged_gym_dev_amd.envs.anymal_c.mixed_terrains.anymal_c_rough_config import AnymalCRoughCfg, AnymalCRoughCfgPPO


AnymalBRoughCfg = cfg_class("AnymalBRoughCfg", AnymalCRoughCfg, dict(
    asset=S(
        file='{LEGGED_GYM_ROOT_DIR}/resources/robots/anymal_b/urdf/anymal_b.urdf', name='anymal_b',
        foot_name='FOOT',
    ),
), doc=None, module=__name__)

AnymalBRoughCfgPPO = cfg_class("AnymalBRoughCfgPPO", AnymalCRoughCfgPPO, dict(
    runner=S(
        run_name='', experiment_name='rough_anymal_b', load_run=-1,
    ),
), doc=None, module=__name__)
