"""ANYmal-B: the ANYmal-C rough-terrain task on the B model (reference: envs/anymal_b/anymal_b_config.py:33-45;
task "anymal_b" uses the Anymal class, envs/__init__.py:56)."""
from legged_gym_dev_amd.envs.anymal_c.mixed_terrains.anymal_c_rough_config import AnymalCRoughCfg, AnymalCRoughCfgPPO


class AnymalBRoughCfg(AnymalCRoughCfg):
    class asset(AnymalCRoughCfg.asset):
        file = "{LEGGED_GYM_ROOT_DIR}/resources/robots/anymal_b/urdf/anymal_b.urdf"
        name = "anymal_b"
        foot_name = "FOOT"


class AnymalBRoughCfgPPO(AnymalCRoughCfgPPO):
    class runner(AnymalCRoughCfgPPO.runner):
        run_name = ""
        experiment_name = "rough_anymal_b"
        load_run = -1
