"""CPU: host-side logic of this round -- bench.py's rank launcher, the guards on unimplemented randomisations, the oracle's
subset reset, and the runner's refusal of an env built for another rank."""
import os
import sys
import types

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tests import harness  # noqa: E402


def test_bench_rank_environments():
    import bench
    envs = bench.rank_environments(4, 29517, base={"PATH": "/bin"})
    assert [e["RANK"] for e in envs] == ["0", "1", "2", "3"] == [e["LOCAL_RANK"] for e in envs]
    assert all(e["WORLD_SIZE"] == "4" and e["MASTER_ADDR"] == "127.0.0.1" and e["MASTER_PORT"] == "29517" for e in envs)
    assert all(e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and e["PATH"] == "/bin" for e in envs)


def test_bench_launcher_propagates_failure(monkeypatch, tmp_path):
    """launch_ranks starts one child per rank and returns non-zero as soon as one of them fails (no GPU involved: the
    children are a stand-in script)."""
    import bench
    script = tmp_path / "child.py"
    script.write_text("import os, sys, time\n"
                      "r = int(os.environ['RANK'])\n"
                      "assert os.environ['WORLD_SIZE'] == '3'\n"
                      "if r == 1: sys.exit(7)\n"
                      "time.sleep(30)\n")
    monkeypatch.setattr(bench, "__file__", str(script))
    monkeypatch.setenv("LG_BENCH_SHARE_GPU", "1")
    import time
    t0 = time.time()
    assert bench.launch_ranks(3, []) == 7
    assert time.time() - t0 < 20          # the sleeping ranks were stopped, not waited for


def test_hopper_only_dof_randomisation_flags_are_inert():
    """domain_rand.dof_properties.* belong to the hopper env; LeggedRobot._process_dof_props (legged_robot.py:301-328) never reads
    them, so switching them on changes nothing here either (they used to be refused)."""
    from legged_gym_dev_amd.envs.base.env_setup import EnvSetup, sim_dt_float
    from legged_gym_dev_amd.model.robot_model import compile_model, resolve_model
    cfg = harness.make_cfg("anymal_c_flat")
    cfg.domain_rand.dof_properties.randomize_stiffness = cfg.domain_rand.dof_properties.randomize_damping = True
    cm = compile_model(resolve_model("", "anymal_c"))
    s = EnvSetup(cfg, cm, sim_dt_float(cfg.sim.dt), seed=1)
    assert (s.p_gains == 80.0).all()


def test_oracle_subset_reset_equals_in_step_reset(oracle_built):
    """reset_idx(env_ids) called directly (lgo_reset_ids) does to those envs what the same reset does inside a step."""
    from legged_gym_dev_amd.envs.base.env_setup import EnvSetup, sim_dt_float
    from legged_gym_dev_amd.model.robot_model import compile_model, resolve_model
    cfg = harness.make_cfg("anymal_c_flat")
    cm = compile_model(resolve_model("", "anymal_c"))

    def mk():
        return oracle_built.OracleEnv(EnvSetup(cfg, cm, sim_dt_float(cfg.sim.dt), seed=4))
    a, b = mk(), mk()
    try:
        rng = np.random.default_rng(0)
        for e in (a, b):
            e.set_step_counter(0)
            e.inject(0)
            e.call("reset_all")
        for t in range(3):
            act = rng.uniform(-1, 1, (64, 12)).astype(np.float32)
            a.step(act)
            b.step(act)
        ids = np.array([3, 17, 40], np.int32)
        sums = a.get("episode_sums")
        a.call("reset_ids", ids.ctypes.data, 3)
        # b: force the same envs to time out inside a teacher-forced post step at the same counter
        assert int(a.get("n_reset")[0]) == 3
        assert (a.get("episode_length")[ids] == 0).all()
        assert (a.get("dof_state")[ids][..., 1] == 0).all()
        rest = np.setdiff1d(np.arange(64), ids)
        np.testing.assert_array_equal(a.get("root_states")[rest], b.get("root_states")[rest])
        k = [i for i in range(sums.shape[0]) if np.abs(sums[i]).sum() > 0]
        want = sums[k][:, ids].mean(1) / np.float32(cfg.env.episode_length_s)
        np.testing.assert_allclose(a.get("extras_episode")[k], want, rtol=1e-5)
        assert (a.get("episode_sums")[:, ids] == 0).all()
        # the draws are the env's reset slots at the current counter: a second identical context gives the same state
        b.call("reset_ids", ids.ctypes.data, 3)
        np.testing.assert_array_equal(a.get("root_states"), b.get("root_states"))
        np.testing.assert_array_equal(a.get("dof_state"), b.get("dof_state"))
    finally:
        a.close()
        b.close()


def test_runner_refuses_env_of_another_rank():
    import torch
    from legged_gym_dev_amd.rl.runner import OnPolicyRunner

    class Comm:
        rank, world_size = 1, 2
    env = types.SimpleNamespace(rank=0, world_size=1, device="cuda:0")
    cfg = {"runner": {"num_steps_per_env": 24, "save_interval": 50}, "algorithm": {}, "policy": {}}
    with pytest.raises(RuntimeError, match="rank 0 of 1"):
        OnPolicyRunner(env, cfg, None, device="cuda:0", comm=Comm())
    del torch
