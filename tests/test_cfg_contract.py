"""No silently ignored configuration field: every leaf of every registered task's env configuration is listed in
legged_gym_dev_amd/envs/base/cfg_contract.py as consumed / inert / fixed / unmodelled; fixed values are refused at construction, the
unmodelled ones warn, and what the table calls consumed by the setup is really read (recording proxy).
Reference: the asset options handed to the simulator, legged_gym/envs/base/legged_robot.py:692-705,745; defaults
legged_robot_config.py:104-124."""
import warnings

import numpy as np
import pytest

from legged_gym_dev_amd.envs import task_registry
from legged_gym_dev_amd.envs.base import cfg_contract as cc
from legged_gym_dev_amd.envs.base.env_setup import EnvSetup, sim_dt_float
from legged_gym_dev_amd.model.robot_model import compile_model, resolve_model
from legged_gym_dev_amd.utils.helpers import class_to_dict
from tests import harness

TASKS = sorted(task_registry.task_classes.keys()) if hasattr(task_registry, "task_classes") else [
    "anymal_c_rough", "anymal_c_flat", "anymal_c_rough_trajectory", "anymal_c_flat_trajectory", "cassie", "anymal_b", "a1"]


@pytest.mark.parametrize("task", TASKS)
def test_every_leaf_of_every_registered_cfg_is_in_the_contract(task):
    env_cfg, _ = task_registry.get_cfgs(task)
    paths = [p for p, _ in cc.leaves(class_to_dict(env_cfg))]
    assert len(paths) > 100
    missing = [p for p in paths if cc.lookup(p) is None]
    assert not missing, f"not in cfg_contract.CONTRACT: {missing}"
    kinds = {cc.lookup(p)[1] for p in paths}
    assert kinds <= {cc.CONSUMED, cc.INERT, cc.FIXED, cc.UNMODELLED}
    for row in cc.CONTRACT:                                   # fixed rows carry their accepted values, unmodelled ones a predicate
        assert len(row) == (4 if row[1] in (cc.FIXED, cc.UNMODELLED) else 3), row[0]


def test_an_unlisted_leaf_is_refused():
    cfg = harness.make_cfg("anymal_c_flat")
    cfg.asset.some_new_option = 3
    with pytest.raises(AttributeError, match="asset.some_new_option"):
        cc.enforce(cfg)


@pytest.mark.parametrize("path,value", [
    ("asset.fix_base_link", True), ("asset.default_dof_drive_mode", 1), ("asset.angular_damping", 0.1), ("asset.linear_damping", 0.5),
    ("sim.physx.solver_type", 0), ("sim.physx.num_velocity_iterations", 1), ("sim.up_axis", 0), ("commands.num_commands", 3),
    ("sim.physx.contact_collection", 0), ("terrain.dynamic_friction", 0.5)])
def test_values_the_build_does_not_implement_are_refused_at_construction(path, value):
    cfg = harness.make_cfg("anymal_c_flat")
    obj = cfg
    *parents, leaf = path.split(".")
    for p in parents:
        obj = getattr(obj, p)
    setattr(obj, leaf, value)
    cm = compile_model(resolve_model("", "anymal_c"))
    with pytest.raises(NotImplementedError, match=leaf):
        EnvSetup(cfg, cm, sim_dt_float(cfg.sim.dt))


def test_unmodelled_values_warn_once_and_are_accepted():
    cfg = harness.make_cfg("anymal_c_flat")
    assert cfg.asset.self_collisions == 0                     # anymal_c_flat_config.py:42 enables self-collision
    cc._warned.clear()
    cm = compile_model(resolve_model("", "anymal_c"))
    with pytest.warns(UserWarning, match="self_collisions"):
        EnvSetup(cfg, cm, sim_dt_float(cfg.sim.dt))
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        EnvSetup(cfg, cm, sim_dt_float(cfg.sim.dt))           # the second construction is silent
    rough = harness.make_cfg("anymal_c_rough") if "anymal_c_rough" in TASKS else None
    if rough is not None:
        assert rough.asset.self_collisions == 1               # disabled there: nothing to warn about
        cc._warned.clear()
        with warnings.catch_warnings():
            warnings.simplefilter("error")
            cc.enforce(rough)


def test_asset_options_reach_the_device_constants():
    cfg = harness.make_cfg("anymal_c_flat")
    cfg.asset.armature, cfg.asset.max_linear_velocity, cfg.asset.max_angular_velocity = 0.02, 30.0, 40.0
    cfg.asset.thickness, cfg.sim.physx.rest_offset = 0.015, 0.002
    cm = compile_model(resolve_model("", "anymal_c"))
    c, _, _keep = EnvSetup(cfg, cm, sim_dt_float(cfg.sim.dt)).to_structs()
    assert abs(c.armature - 0.02) < 1e-7 and c.max_linear_velocity == 30.0 and c.max_angular_velocity == 40.0
    assert abs(c.rest_offset - 0.017) < 1e-7
    assert abs(c.gravity[2] + 9.81) < 1e-6
    cfg.asset.disable_gravity = True
    c, _, _keep = EnvSetup(cfg, cm, sim_dt_float(cfg.sim.dt)).to_structs()
    assert c.gravity[0] == 0.0 and c.gravity[1] == 0.0 and c.gravity[2] == 0.0


class _Rec:
    """Attribute-access recorder around a cfg tree."""

    def __init__(self, obj, path, log):
        object.__setattr__(self, "_o", obj)
        object.__setattr__(self, "_p", path)
        object.__setattr__(self, "_log", log)

    def __getattr__(self, k):
        v = getattr(self._o, k)
        p = f"{self._p}.{k}" if self._p else k
        if hasattr(v, "__dict__") and not callable(v) and not isinstance(v, (np.ndarray, dict, list)):
            return _Rec(v, p, self._log)
        self._log.add(p)
        return v

    def __setattr__(self, k, v):
        setattr(self._o, k, v)

    def __dir__(self):
        return dir(self._o)


@pytest.mark.parametrize("task", ["anymal_c_flat", "cassie"])
def test_what_the_table_calls_consumed_by_the_setup_is_read_and_inert_leaves_are_not(task, monkeypatch):
    cfg = harness.make_cfg(task)
    cm = compile_model(resolve_model("", "anymal_c" if task != "cassie" else "cassie"))
    terrain = None
    if cfg.terrain.mesh_type in ("heightfield", "trimesh"):
        from legged_gym_dev_amd.utils.terrain import Terrain
        cfg.terrain.num_rows, cfg.terrain.num_cols = 2, 2
        terrain = Terrain(cfg.terrain, cfg.env.num_envs)
    log = set()
    monkeypatch.setattr(cc, "enforce", lambda cfg: None)      # enforce() walks the whole tree (tested above); record the setup's own reads
    EnvSetup(_Rec(cfg, "", log), cm, sim_dt_float(cfg.sim.dt), terrain=terrain).to_structs()
    read_inert = sorted(p for p in log if cc.lookup(p) is not None and cc.lookup(p)[1] == cc.INERT)
    assert not read_inert, f"listed as inert but read by the setup: {read_inert}"
    setup_rows = [p for p, _ in cc.leaves(class_to_dict(cfg)) if cc.lookup(p)[1] == cc.CONSUMED and cc.lookup(p)[2].startswith("setup")]
    conditional = {"noise.noise_scales.height_measurements"} if not cfg.terrain.measure_heights else set()     # read with the height scan only
    not_read = [p for p in setup_rows if p not in log and p not in conditional and not p.startswith("curriculum.")
                and "trajectory" not in cc.lookup(p)[2]]
    assert not not_read, f"listed as consumed by the setup but never read: {not_read}"
