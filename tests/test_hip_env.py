"""GPU parity tests of the environment step: every call goes through the C-ABI of
liblegged_hip.so.  (1) fixtures produced by the reference's own Python, (2) the CPU oracle on the
same seeded inputs, (3) size-independent properties at the BASELINE size (4096 envs)."""
import numpy as np
import pytest

from tests import harness

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", harness.FIXTURES)
def test_hip_replays_reference_steps(name):
    z, meta = harness.load_fixture(name)
    setup, _ = harness.make_setup(name, z, meta)
    hs = z["const_height_samples"] if "const_height_samples" in z.files else None
    env = harness.HipHandle(setup, hs)
    try:
        harness.replay_fixture(env, z, meta)
    finally:
        env.close()


def _pair(name, oracle_built, n=None):
    z, meta = harness.load_fixture(name)
    cfg = harness.make_cfg(name)
    if n:
        cfg.env.num_envs = n
        meta = dict(meta, num_envs=n)
    from legged_gym_dev_amd.envs.base.env_setup import EnvSetup, sim_dt_float
    from legged_gym_dev_amd.model.robot_model import compile_model, resolve_model
    cm = compile_model(resolve_model("", meta["robot"]))
    terrain = harness.FixtureTerrain(z, meta, cfg) if "const_height_samples" in z.files else None
    hs = z["const_height_samples"] if terrain else None

    def mk():
        return EnvSetup(cfg, cm, sim_dt_float(cfg.sim.dt), terrain=terrain, seed=7)
    return harness.HipHandle(mk(), hs), oracle_built.OracleEnv(mk(), hs), z, meta


def _seed_state(envs, rng, n, A, z0, origins=None, spread=0.02):
    root = np.zeros((n, 13), np.float32)
    root[:, 2] = z0 + rng.uniform(-spread, 0.15, n)
    if origins is not None:
        root[:, :3] += origins
        root[:, :2] += rng.uniform(-2, 2, (n, 2))
    ang = rng.uniform(-0.3, 0.3, (n, 3))
    qw = np.sqrt(np.maximum(1 - (ang ** 2).sum(1) / 4, 0))
    root[:, 3:6], root[:, 6] = ang / 2, qw
    root[:, 3:7] /= np.linalg.norm(root[:, 3:7], axis=1, keepdims=True)
    root[:, 7:13] = rng.uniform(-0.5, 0.5, (n, 6))
    dof = np.zeros((n, A, 2), np.float32)
    dof[..., 0] = envs[0].setup.default_dof_pos + rng.uniform(-0.2, 0.2, (n, A))
    dof[..., 1] = rng.uniform(-1, 1, (n, A))
    fr = rng.uniform(0.2, 1.2, n).astype(np.float32)
    dm = rng.uniform(-5, 5, n).astype(np.float32)
    for e in envs:
        e.set("root_states", root)
        e.set("dof_state", dof)
        e.set("friction", fr)
        e.set("base_mass_delta", dm)


@pytest.mark.parametrize("name,z0", [("anymal_c_flat", 0.50), ("anymal_c_rough", 0.55), ("cassie", 0.85),
                                     ("anymal_c_allrewards", 0.30), ("a1", 0.28), ("anymal_b", 0.50)])
def test_physics_substep_matches_oracle(name, z0, oracle_built):
    """One sim_dt of ABA + contact: HIP lane-parallel kernel vs the scalar oracle.
    Tolerance: 2e-4 abs/rel on state, 0.5 N + 2e-3 rel on contact forces (fp32, different
    summation order at the base)."""
    hip, ora, z, meta = _pair(name, oracle_built, n=256)
    try:
        rng = np.random.default_rng(3)
        n, A = 256, meta["num_dofs"]
        origins = z["const_env_origins_init"][rng.integers(0, len(z["const_env_origins_init"]), n)] if meta["custom_origins"] else None
        _seed_state([hip, ora], rng, n, A, z0, origins)
        tau = rng.uniform(-20, 20, (n, A)).astype(np.float32)
        n_contact = 0
        for step in range(6):
            for e in (hip, ora):
                e.set("torques", tau)
                e.call("simulate")
            cf_h, cf_o = hip.get("contact_forces"), ora.get("contact_forces")
            n_contact += int((np.abs(cf_o).sum(-1) > 0).sum())
            np.testing.assert_allclose(hip.get("root_states"), ora.get("root_states"), rtol=2e-4, atol=2e-4)
            # joint velocities of the 0.15 kg Cassie toe links see accelerations > 1e3 rad/s^2: 2e-3
            np.testing.assert_allclose(hip.get("dof_state"), ora.get("dof_state"), rtol=2e-3, atol=2e-3)
            np.testing.assert_allclose(cf_h, cf_o, rtol=2e-3, atol=0.5)
            # re-synchronise so fp32 drift does not accumulate across steps (chaotic contacts)
            hip.set("root_states", ora.get("root_states"))
            hip.set("dof_state", ora.get("dof_state"))
        assert n_contact > 100, "test state never touched the ground"
    finally:
        hip.close()
        ora.close()


@pytest.mark.parametrize("name", ["anymal_c_flat", "anymal_c_rough", "cassie", "a1", "anymal_b"])
def test_full_step_philox_matches_oracle(name, oracle_built):
    """lg_step with the built-in Philox streams (no injection) vs the oracle on the same seed:
    masks / counters bit-exact, fp32 state within tolerance per policy step."""
    hip, ora, z, meta = _pair(name, oracle_built, n=128)
    try:
        rng = np.random.default_rng(5)
        n, A = 128, meta["num_dofs"]
        for e in (hip, ora):
            if meta["custom_origins"]:
                e.set("env_origins", z["const_env_origins_init"][np.arange(n) % len(z["const_env_origins_init"])])
                e.set("terrain_levels", z["const_terrain_levels_init"][np.arange(n) % len(z["const_terrain_levels_init"])])
                e.set("terrain_types", z["const_terrain_types"][np.arange(n) % len(z["const_terrain_types"])])
            e.set_step_counter(0)
            e.inject(0)
            e.call("reset_all")
        np.testing.assert_allclose(hip.get("root_states"), ora.get("root_states"), rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(hip.get("dof_state"), ora.get("dof_state"), rtol=1e-6, atol=1e-6)
        ep = rng.integers(0, 1000, n)
        ep[:8] = [199, 499, 999, 1000, 1001, 399, 0, 1]
        for e in (hip, ora):
            e.set("episode_length", ep)
            e.set_step_counter(748)                     # a push (751) falls inside the window
        for t in range(4):
            act = rng.uniform(-1, 1, (n, A)).astype(np.float32)
            hip.step(act)
            ora.step(act)
            np.testing.assert_array_equal(hip.get("reset"), ora.get("reset"), err_msg=f"step {t} reset")
            np.testing.assert_array_equal(hip.get("time_out"), ora.get("time_out"))
            np.testing.assert_array_equal(hip.get("episode_length"), ora.get("episode_length"))
            np.testing.assert_array_equal(hip.get("terrain_levels"), ora.get("terrain_levels"))
            assert int(hip.get("n_reset")[0]) == int(ora.get("n_reset")[0])
            for key, tol in (("obs", 2e-3), ("rew", 2e-3), ("root_states", 1e-3), ("dof_state", 2e-3),
                             ("commands", 1e-6), ("torques", 5e-3)):
                np.testing.assert_allclose(hip.get(key), ora.get(key), rtol=tol, atol=tol, err_msg=f"step {t} {key}")
            # keep the two trajectories glued together (contacts amplify fp32 noise)
            for key in ("root_states", "dof_state", "lstm_h", "lstm_c", "last_dof_vel", "last_root_vel",
                        "feet_air_time", "episode_sums"):
                hip.set(key, ora.get(key))
    finally:
        hip.close()
        ora.close()


def test_rollout_properties_at_baseline_size():
    """BASELINE.json configs[1] size (anymal_c_flat, 4096 envs): size-independent properties of
    100 policy steps with random actions -- finite state, unit quaternions, reset => episode
    length 0, time_out => reset, zero-velocity joints after reset, clip bounds respected,
    weight supported on average, determinism of two identically seeded contexts."""
    import torch
    cfg = harness.make_cfg("anymal_c_flat")
    cfg.env.num_envs = 4096
    from legged_gym_dev_amd.envs.base.env_setup import EnvSetup, sim_dt_float
    from legged_gym_dev_amd.model.robot_model import compile_model, resolve_model
    cm = compile_model(resolve_model("", "anymal_c"))
    outs = []
    for rep in range(2):
        env = harness.HipHandle(EnvSetup(cfg, cm, sim_dt_float(cfg.sim.dt), seed=11))
        try:
            g = torch.Generator(device="cuda").manual_seed(0)
            env.set_step_counter(0)
            env.call("reset_all")
            for t in range(100):
                a = torch.randn(4096, 12, device="cuda", generator=g) * 0.5
                env.core.step(a)
                if t % 25 == 24 or t == 0:
                    root, dof = env.get("root_states"), env.get("dof_state")
                    rst, to = env.get("reset").astype(bool), env.get("time_out").astype(bool)
                    ep = env.get("episode_length")
                    assert np.isfinite(root).all() and np.isfinite(dof).all() and np.isfinite(env.get("obs")).all()
                    np.testing.assert_allclose(np.linalg.norm(root[:, 3:7], axis=1), 1.0, atol=1e-4)
                    assert (ep[rst] == 0).all() and (rst | ~to).all()
                    assert (dof[rst][..., 1] == 0).all()
                    assert np.abs(env.get("obs")).max() <= 100.0
            fz = env.get("contact_forces")[:, :, 2].sum(1)
            assert 0.3 * 511 < fz.mean() < 3.0 * 511, fz.mean()
            outs.append((env.get("root_states"), env.get("obs")))
        finally:
            env.close()
    np.testing.assert_array_equal(outs[0][0], outs[1][0])
    np.testing.assert_array_equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("name", ["anymal_c_flat", "anymal_c_pd_V", "anymal_c_rough", "cassie", "a1", "anymal_b"])
def test_fused_control_loop_equals_launch_per_substep(name):
    """lg_step's single-launch control loop (k_substeps: clip + decimation x {torque law, physics} with the
    state resident on chip) against the operator-level sequence lg_set_actions / lg_compute_torques /
    lg_simulate / lg_post_physics_step on a second context with the same seed (133 envs: the last block of
    each kernel is ragged).  Same source, two kernels: hipcc contracts multiply-adds differently in each, so
    fp32 buffers agree to rounding (rtol = atol = 2e-4 on >= 99.8 % of the elements after decimation substeps of contact dynamics, the
    state is re-glued every policy step), clipped actions and masks bit-exactly."""
    z, meta = harness.load_fixture(name)
    n = 133
    cfg = harness.make_cfg(name)
    cfg.env.num_envs = n
    meta = dict(meta, num_envs=n)
    from legged_gym_dev_amd.envs.base.env_setup import EnvSetup, sim_dt_float
    from legged_gym_dev_amd.model.robot_model import compile_model, resolve_model
    cm = compile_model(resolve_model("", meta["robot"]))
    terrain = harness.FixtureTerrain(z, meta, cfg) if "const_height_samples" in z.files else None
    hs = z["const_height_samples"] if terrain else None

    def mk():
        return harness.HipHandle(EnvSetup(cfg, cm, sim_dt_float(cfg.sim.dt), terrain=terrain, seed=7), hs)
    a, b = mk(), mk()
    try:
        rng = np.random.default_rng(9)
        A = meta["num_dofs"]
        for e in (a, b):
            e.set_step_counter(0)
            e.inject(0)
            e.call("reset_all")
        dec = int(cfg.control.decimation)
        if meta["custom_origins"]:
            for e in (a, b):
                e.set("env_origins", z["const_env_origins_init"][np.arange(n) % len(z["const_env_origins_init"])])
                e.set("terrain_levels", z["const_terrain_levels_init"][np.arange(n) % len(z["const_terrain_levels_init"])])
                e.set("terrain_types", z["const_terrain_types"][np.arange(n) % len(z["const_terrain_types"])])
                e.call("reset_all")
        for t in range(6):
            act = (rng.uniform(-3, 3, (n, A)) * (200.0 if t == 5 else 1.0)).astype(np.float32)   # the last step exercises the action clip
            a.step(act)                                            # fused
            b.set_actions(act)
            for _ in range(dec):
                b.call("compute_torques")
                b.call("simulate")
            b.call("post_physics_step")
            for key in ("actions", "reset", "time_out", "episode_length"):
                np.testing.assert_array_equal(a.get(key), b.get(key), err_msg=f"step {t} {key}")
            for key in ("torques", "dof_state", "root_states", "obs", "rew", "lstm_h", "lstm_c", "episode_sums", "feet_air_time"):
                x, y = a.get(key).astype(np.float64), b.get(key).astype(np.float64)
                bad = np.abs(x - y) > 2e-4 + 2e-4 * np.abs(y)
                # contact dynamics amplify rounding: a stray element may exceed the band, never by much
                assert bad.mean() <= 2e-3 and np.abs(x - y).max() <= 5e-2 * max(1.0, np.abs(y).max()), f"step {t} {key}: {bad.sum()} of {bad.size}"
            fa, fb = a.get("contact_forces"), b.get("contact_forces")
            np.testing.assert_allclose(fa, fb, rtol=2e-3, atol=2e-3 * max(1.0, float(np.abs(fb).max())), err_msg=f"step {t} contact_forces")
            for key in ("root_states", "dof_state", "lstm_h", "lstm_c", "last_dof_vel", "last_root_vel", "feet_air_time",
                        "episode_sums", "last_actions", "commands"):
                b.set(key, a.get(key))
    finally:
        a.close()
        b.close()


@pytest.mark.parametrize("name", ["a1", "cassie"])
def test_joint_limit_constraints_match_oracle(name, oracle_built):
    """Joint-limit constraints of the physics (URDF lower/upper): free flight, joints driven into their stops --
    HIP lanes vs the oracle step by step (state re-glued each step), and the stop is never passed by more than
    2e-3 rad.  Tolerance 5e-4 abs/rel on joint state (fp32, different summation order at the base)."""
    hip, ora, z, meta = _pair(name, oracle_built, n=64)
    try:
        n, A = 64, meta["num_dofs"]
        cm = hip.setup.cm if hasattr(hip.setup, "cm") else None
        from legged_gym_dev_amd.model.robot_model import compile_model, resolve_model
        cm = compile_model(resolve_model("", meta["robot"]))
        lo, hi = cm["q_lower"].astype(np.float64), cm["q_upper"].astype(np.float64)
        J = int(cm["joints_per_leg"])
        rng = np.random.default_rng(21)
        root = np.zeros((n, 13), np.float32)
        root[:, 2] = 50.0
        root[:, 6] = 1.0
        dof = np.zeros((n, A, 2), np.float32)
        dof[..., 0] = 0.5 * (lo + hi)
        tau = np.zeros((n, A), np.float32)
        tmax = 3.0 if name == "a1" else 20.0
        for i in range(n):                                   # each env: a random subset of joints pushed into a random stop
            sel = rng.random(A) < 0.35
            up = rng.random(A) < 0.5
            dof[i, sel & up, 0] = (hi - 0.01)[sel & up]
            dof[i, sel & ~up, 0] = (lo + 0.01)[sel & ~up]
            tau[i, sel & up] = tmax
            tau[i, sel & ~up] = -tmax
        dof[:4, :, 0] = hi + 0.04                            # and a few that start beyond their stops
        tau[:4] = 0.0
        for e in (hip, ora):
            e.set("root_states", root)
            e.set("dof_state", dof)
            e.set("torques", tau)
        worst = 0.0
        for t in range(12):
            hip.call("simulate")
            ora.call("simulate")
            dh, do = hip.get("dof_state").astype(np.float64), ora.get("dof_state").astype(np.float64)
            np.testing.assert_allclose(dh, do, rtol=5e-4, atol=5e-4, err_msg=f"step {t}")
            np.testing.assert_allclose(hip.get("root_states"), ora.get("root_states"), rtol=5e-4, atol=5e-4, err_msg=f"step {t} root")
            worst = max(worst, float(np.max(dh[4:, :, 0] - hi)), float(np.max(lo - dh[4:, :, 0])))
            hip.set("dof_state", do)
            hip.set("root_states", ora.get("root_states"))
        assert worst <= 2e-3, worst
        assert np.all(dh[:4, :, 0] < hi + 0.04)              # the ones that started outside are on their way back
    finally:
        hip.close()
        ora.close()


def test_env_shards_equal_the_unsharded_run():
    """Multi-GPU by construction: rank r of G owns envs [r N/G, (r+1) N/G) and keys every random
    stream with the GLOBAL env id.  Two shards of 96 envs (env_offset 0 and 96, total 192), stepped with the matching
    halves of the actions, reproduce the unsharded 192-env run bit for bit: resets, commands, pushes, observation noise."""
    cfg = harness.make_cfg("anymal_c_flat")
    from legged_gym_dev_amd.envs.base.env_setup import EnvSetup, sim_dt_float
    from legged_gym_dev_amd.model.robot_model import compile_model, resolve_model
    cm = compile_model(resolve_model("", "anymal_c"))
    import copy

    def mk(n, off, tot):
        c = copy.deepcopy(cfg)
        c.env.num_envs = n
        return harness.HipHandle(EnvSetup(c, cm, sim_dt_float(c.sim.dt), env_offset=off, total_envs=tot, seed=13))
    full, lo, hi = mk(192, 0, 192), mk(96, 0, 192), mk(96, 96, 192)
    try:
        rng = np.random.default_rng(4)
        for e in (full, lo, hi):
            e.set_step_counter(0)
            e.inject(0)
            e.call("reset_all")
        # per-env constants are drawn by the host for the global range and sliced per shard: mirror that here
        for key in ("friction", "base_mass_delta", "env_origins"):
            v = full.get(key)
            lo.set(key, v[:96]); hi.set(key, v[96:])
        for e in (lo, hi, full):
            e.call("reset_all")
        ep = rng.integers(0, 1001, 192)
        full.set("episode_length", ep); lo.set("episode_length", ep[:96]); hi.set("episode_length", ep[96:])
        for e in (full, lo, hi):
            e.set_step_counter(745)                      # the push at 751 falls inside the window
        for t in range(8):
            act = rng.uniform(-2, 2, (192, 12)).astype(np.float32)
            full.step(act); lo.step(act[:96]); hi.step(act[96:])
            for key in ("obs", "rew", "reset", "time_out", "commands", "root_states", "dof_state", "episode_length", "lstm_h"):
                want = full.get(key)
                got = np.concatenate([lo.get(key), hi.get(key)], axis=0 if key != "lstm_h" else 1)
                if key == "lstm_h":
                    want = want.reshape(2, 192, -1); got = np.concatenate([lo.get(key).reshape(2, 96, -1), hi.get(key).reshape(2, 96, -1)], axis=1)
                np.testing.assert_array_equal(got, want, err_msg=f"step {t} {key}")
    finally:
        for e in (full, lo, hi):
            e.close()
