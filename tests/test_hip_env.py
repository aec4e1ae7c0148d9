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
                                     ("anymal_c_allrewards", 0.30), ("a1", 0.28), ("anymal_b", 0.50), ("anymal_c_randomised", 0.50)])
def test_physics_substep_matches_oracle(name, z0, oracle_built):
    """One sim_dt of ABA + contact: HIP lane-parallel kernel vs the scalar oracle.
    Tolerance: 2e-4 abs/rel on state, 0.5 N + 2e-3 rel on contact forces (fp32, different
    summation order at the base).  anymal_c_randomised: per-env restitution / compliance / thickness (lg_buffers.material) live in
    the contact law."""
    hip, ora, z, meta = _pair(name, oracle_built, n=256)
    try:
        rng = np.random.default_rng(3)
        n, A = 256, meta["num_dofs"]
        if name == "anymal_c_randomised":
            assert hip.core.cfg_struct.material_rand == 1
            mat = np.zeros((n, 4), np.float32)
            mat[:, 0], mat[:, 1], mat[:, 2] = rng.uniform(0, 1, n), rng.uniform(0, 2e-6, n), rng.uniform(0, 0.03, n)
            mat[::7] = 0.0
            for e in (hip, ora):
                e.set("material", mat)
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
            # Joint state: 2e-4 like the root.  What may leave that band -- asserted as stated -- is a joint RATE of an
            # environment that is in ground contact during this substep (the projected-Jacobi sweeps clamp impulses to the
            # friction cone; a lane-order or 1-ulp reciprocal difference that lands a tangential impulse on the other side of
            # the clamp moves the rates of that leg by ~1e-3 rad/s): never a position, never an env in free flight, never
            # beyond 2e-3, fewer than 0.1 % of the entries.  tools/diag_physics_band.py lists them (14 of 331 776 rates over
            # three seeds x six robots: all rates, all in contact, largest 1.8e-3).
            dh, do = hip.get("dof_state").astype(np.float64), ora.get("dof_state").astype(np.float64)
            err = np.abs(dh - do)
            band = err > 2e-4 + 2e-4 * np.abs(do)
            assert not band[..., 0].any(), f"step {step}: joint positions outside 2e-4"
            out = band[..., 1]
            in_contact = np.abs(cf_o).reshape(n, -1).sum(1) > 0.0      # any contact row that ended the substep with an impulse
            assert out.mean() < 1e-3, int(out.sum())
            assert in_contact[np.nonzero(out)[0]].all(), "a joint rate of an env without ground contact left the 2e-4 band"
            assert (err[..., 1] <= 2e-3 + 2e-3 * np.abs(do[..., 1])).all(), err[..., 1].max()
            np.testing.assert_allclose(cf_h, cf_o, rtol=2e-3, atol=0.5)
            # re-synchronise so fp32 drift does not accumulate across steps (chaotic contacts)
            hip.set("root_states", ora.get("root_states"))
            hip.set("dof_state", ora.get("dof_state"))
        assert n_contact > 100, "test state never touched the ground"
    finally:
        hip.close()
        ora.close()


@pytest.mark.parametrize("name,min_bodies", [("anymal_c_flat", 7), ("anymal_c_rough", 7), ("cassie", 3)])
def test_fallen_robots_match_oracle(name, min_bodies, oracle_built):
    """Robots lying on the ground in every orientation (on their side, on their back): thighs, shanks, feet and base spheres all
    touch, three to eight contacts per leg.  That is the path that sets a control-loop launch's duration (profiles/r04_substeps_spread.txt)
    and the one the round-4 kernel changed most: active contacts dealt between the two lanes of a pair, the first ones in registers
    and the rest through the pair's LDS column, W from the inward pass alone -- against the oracle's literal loops.  Glued after every
    substep; the contact-force band of test_physics_substep_matches_oracle, and the legs must really be loaded with contacts."""
    hip, ora, z, meta = _pair(name, oracle_built, n=256)
    try:
        rng = np.random.default_rng(17)
        n, A = 256, meta["num_dofs"]
        origins = z["const_env_origins_init"][rng.integers(0, len(z["const_env_origins_init"]), n)] if meta["custom_origins"] else None
        _seed_state([hip, ora], rng, n, A, 0.10, origins)
        root = ora.get("root_states").copy()
        q = rng.normal(size=(n, 4)).astype(np.float32)                     # any orientation
        root[:, 3:7] = q / np.linalg.norm(q, axis=1, keepdims=True)
        root[:, 2] += 0.12                                                 # base centre a sphere radius or two above the surface
        root[:, 7:13] *= 0.2
        for e in (hip, ora):
            e.set("root_states", root)
        tau = rng.uniform(-10, 10, (n, A)).astype(np.float32)
        most = 0
        for step in range(8):
            for e in (hip, ora):
                e.set("torques", tau)
                e.call("simulate")
            cf_h, cf_o = hip.get("contact_forces"), ora.get("contact_forces")
            most = max(most, int((np.abs(cf_o).sum(-1) > 0).sum(1).max()))
            assert np.isfinite(hip.get("root_states")).all() and np.isfinite(hip.get("dof_state")).all()
            np.testing.assert_allclose(hip.get("root_states"), ora.get("root_states"), rtol=5e-4, atol=5e-4)
            dh, do = hip.get("dof_state").astype(np.float64), ora.get("dof_state").astype(np.float64)
            err = np.abs(dh - do)
            assert (err[..., 0] <= 5e-4 + 5e-4 * np.abs(do[..., 0])).all(), err[..., 0].max()
            assert (err[..., 1] <= 5e-3 + 5e-3 * np.abs(do[..., 1])).mean() > 0.998, float(err[..., 1].max())
            np.testing.assert_allclose(cf_h, cf_o, rtol=5e-3, atol=1.0)
            hip.set("root_states", ora.get("root_states"))
            hip.set("dof_state", ora.get("dof_state"))
        # (the biped's collision model has fewer bodies: three at once is everything it can put on the ground)
        assert most >= min_bodies, f"no env with many bodies in contact ({most}): the test never left the one-contact-per-leg path"
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
    """lg_step's single-launch control loop (k_substeps: clip + decimation x {torque law, physics} with the state resident
    on chip) against the operator-level sequence lg_set_actions / decimation x {lg_compute_torques, lg_simulate} /
    lg_post_physics_step on a second context with the same seed (133 envs: the last block is ragged).  The operator-level
    entries run the same kernel with one stage switched off, so one instance of the torque code and of physics_lane serves
    both paths and fp32 state crossing HBM between launches is exact: every buffer must be EQUAL bit for bit, over 6 policy
    steps with contacts, resets and the action clip, without re-gluing the two contexts."""
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
            for key in ("actions", "reset", "time_out", "episode_length", "torques", "dof_state", "root_states", "obs", "rew",
                        "lstm_h", "lstm_c", "episode_sums", "feet_air_time", "contact_forces", "commands", "last_dof_vel"):
                np.testing.assert_array_equal(a.get(key), b.get(key), err_msg=f"step {t} {key}")
    finally:
        a.close()
        b.close()


@pytest.mark.parametrize("shape", ["two_waves", "two_per_cu"])
@pytest.mark.parametrize("name", ["anymal_c_flat", "anymal_c_rough", "cassie"])
def test_control_loop_block_shapes_are_bit_identical(name, shape):
    """The control loop on blocks of 4 waves (two physics waves, 64/L envs) against blocks of 2 waves (one physics wave, 32/L envs:
    LG_SUBSTEPS_NW=2): which workgroup an environment lands in and how many waves share its barriers must not change a bit of
    its state -- 133 envs (ragged last block in both shapes), 6 policy steps with contacts and resets.  "two_per_cu": the same
    4-wave kernel compiled for two waves per SIMD (256 registers, some of its state in scratch), which launches of more workgroups
    than CUs take (above 4096 quadruped envs per GPU): register allocation must not change a bit either."""
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
    lib = a.core.lib
    try:
        rng = np.random.default_rng(11)
        A = meta["num_dofs"]
        for e in (a, b):
            e.set_step_counter(0)
            e.inject(0)
            e.call("reset_all")
        for t in range(6):
            act = rng.uniform(-3, 3, (n, A)).astype(np.float32)
            lib.lg_debug_set_substeps_nw(4)
            lib.lg_debug_set_substeps_occ(1)
            a.step(act)
            if shape == "two_waves":
                lib.lg_debug_set_substeps_nw(2)
            else:
                lib.lg_debug_set_substeps_occ(2)
            b.step(act)
            for key in ("reset", "episode_length", "torques", "dof_state", "root_states", "obs", "rew", "lstm_h", "lstm_c",
                        "contact_forces", "feet_air_time"):
                np.testing.assert_array_equal(a.get(key), b.get(key), err_msg=f"step {t} {key}")
    finally:
        lib.lg_debug_set_substeps_nw(0)                            # back to the default: by topology
        lib.lg_debug_set_substeps_occ(0)                           # ... and by grid size
        a.close()
        b.close()


@pytest.mark.parametrize("name", ["anymal_c_flat", "anymal_c_rough", "cassie"])
def test_pair_lane_physics_equals_one_lane_per_leg(name):
    """The two lane maps of the physics -- two lanes per (env, leg) splitting every spatial quantity by rows (lg_physics_pair.h,
    the default) and one lane per (env, leg) (lg_physics.h) -- are the same algorithm with differently associated fp32 sums.
    From bit-identical states with feet, shanks and (Cassie) joint stops in play, one lg_simulate under each map: joint state,
    root state and contact forces agree to 2e-5 relative to the buffer's scale (contacts amplify beyond one substep, which is why
    the comparison restarts from an identical state at each checkpoint)."""
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
    A = meta["num_dofs"]
    most_contacts = 0
    for warm in (2, 5, 9):
        a, b = (harness.HipHandle(EnvSetup(cfg, cm, sim_dt_float(cfg.sim.dt), terrain=terrain, seed=7), hs) for _ in range(2))
        lib = a.core.lib
        try:
            rng = np.random.default_rng(3)
            lib.lg_debug_set_phys_pair(1)
            for e in (a, b):
                e.set_step_counter(0)
                e.inject(0)
                e.call("reset_all")
            for t in range(warm):
                act = rng.uniform(-2, 2, (n, A)).astype(np.float32)
                a.step(act)
                b.step(act)
            for key in ("dof_state", "root_states", "torques"):
                np.testing.assert_array_equal(a.get(key), b.get(key), err_msg=f"warm-up {warm} {key}")
            a.call("simulate")
            lib.lg_debug_set_phys_pair(0)
            b.call("simulate")
            lib.lg_debug_set_phys_pair(1)
            most_contacts = max(most_contacts, int((np.abs(a.get("contact_forces")).sum(-1) > 1.0).sum()))
            for key in ("dof_state", "root_states", "contact_forces"):
                x, y = a.get(key).astype(np.float64), b.get(key).astype(np.float64)
                scale = max(1.0, float(np.abs(y).max()))
                bad = np.abs(x - y) > 2e-5 * scale + 2e-5 * np.abs(y)
                assert bad.mean() < 1e-3, (name, warm, key, float(np.abs(x - y).max()), scale, float(bad.mean()))
        finally:
            lib.lg_debug_set_phys_pair(1)
            a.close()
            b.close()
    assert most_contacts >= n // 4, (name, most_contacts)             # the comparison was not of free flight only


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


def _glue(hip, ora, keys=("root_states", "dof_state", "lstm_h", "lstm_c", "last_dof_vel", "last_root_vel", "feet_air_time",
                          "episode_sums", "commands", "last_actions")):
    for key in keys:
        hip.set(key, ora.get(key))


@pytest.mark.parametrize("name", ["anymal_c_flat", "anymal_c_rough", "cassie"])
def test_reset_ids_matches_oracle(name, oracle_built):
    """lg_reset_ids = LeggedRobot.reset_idx(env_ids) for a caller-given subset (legged_robot.py:147-187, the indexed state
    writes of :428,452): state, commands, buffers, actuator-net state, terrain level / origin and the extras of the subset
    equal the oracle's reset_idx on the same ids and the same Philox slots; envs outside the subset are untouched."""
    import torch
    hip, ora, z, meta = _pair(name, oracle_built, n=96)
    try:
        rng = np.random.default_rng(12)
        n, A = 96, meta["num_dofs"]
        for e in (hip, ora):
            if meta["custom_origins"]:
                e.set("env_origins", z["const_env_origins_init"][np.arange(n) % len(z["const_env_origins_init"])])
                e.set("terrain_levels", z["const_terrain_levels_init"][np.arange(n) % len(z["const_terrain_levels_init"])])
                e.set("terrain_types", z["const_terrain_types"][np.arange(n) % len(z["const_terrain_types"])])
            e.set_step_counter(0)
            e.inject(0)
            e.call("reset_all")
        for t in range(3):
            act = rng.uniform(-1, 1, (n, A)).astype(np.float32)
            hip.step(act)
            ora.step(act)
            _glue(hip, ora)
        if meta["custom_origins"]:                          # walk some robots far enough for the curriculum to move them
            root = ora.get("root_states")
            root[::5, 0] += 5.0
            for e in (hip, ora):
                e.set("root_states", root)
        ids = np.array(sorted(rng.choice(n, 17, replace=False)), np.int32)
        before = {k: hip.get(k) for k in ("root_states", "dof_state", "commands", "episode_length", "episode_sums")}
        ids_dev = torch.as_tensor(ids, device="cuda")
        import ctypes as C
        hip.core.call("reset_ids", C.c_void_p(ids_dev.data_ptr()), len(ids))
        ora.call("reset_ids", ids.ctypes.data, len(ids))
        for key in ("reset", "episode_length", "terrain_levels", "n_reset"):
            np.testing.assert_array_equal(hip.get(key), ora.get(key), err_msg=key)
        assert int(hip.get("n_reset")[0]) == 17
        for key in ("root_states", "dof_state", "commands", "last_actions", "last_dof_vel", "feet_air_time", "env_origins",
                    "episode_sums", "lstm_h", "lstm_c"):
            np.testing.assert_allclose(hip.get(key), ora.get(key), rtol=1e-6, atol=1e-6, err_msg=key)
        np.testing.assert_allclose(hip.get("extras_episode"), ora.get("extras_episode"), rtol=1e-4, atol=1e-6)
        np.testing.assert_array_equal(hip.get("extras_time_outs"), ora.get("extras_time_outs"))
        rest = np.setdiff1d(np.arange(n), ids)
        for key in ("root_states", "dof_state", "commands", "episode_length"):
            np.testing.assert_array_equal(hip.get(key)[rest], before[key][rest], err_msg=f"{key} outside the subset")
        np.testing.assert_array_equal(hip.get("episode_sums")[:, rest], before["episode_sums"][:, rest])
        assert (hip.get("episode_length")[ids] == 0).all() and (hip.get("dof_state")[ids][..., 1] == 0).all()
        hip.core.call("reset_ids", None, 0)                 # empty list: early return (legged_robot.py:156-157)
    finally:
        hip.close()
        ora.close()


def test_fault_guard_is_counted(oracle_built):
    """A runaway base velocity is CLAMPED at cfg.asset.max_linear_velocity / max_angular_velocity and the env carries on, as under
    PhysX (legged_robot.py:701-702; counted in n_vel_clamp / vel_clamp_total); the fault guard is for non-finite state only (the env
    keeps its pose, is brought to rest, is terminated by the post-step and counted in n_fault / fault_total).  HIP == oracle."""
    hip, ora, z, meta = _pair("anymal_c_flat", oracle_built, n=64)
    try:
        for e in (hip, ora):
            e.set_step_counter(0)
            e.inject(0)
            e.call("reset_all")
        act = np.zeros((64, 12), np.float32)
        for e in (hip, ora):
            e.step(act)
        assert int(hip.get("n_fault")[0]) == 0 and int(hip.get("fault_total")[0]) == 0 and int(hip.get("vel_clamp_total")[0]) == 0
        root = ora.get("root_states")
        root[5, 2] = 200.0                                  # high above the ground: nothing but the clamp acts on it
        root[5, 7:10] = 2.0e4                               # absurd base velocity
        root[9, 10] = np.nan
        for e in (hip, ora):
            e.set("root_states", root)
            e.step(act)
        lim = float(hip.setup.cfg.asset.max_linear_velocity)
        for e in (hip, ora):
            assert int(e.get("n_fault")[0]) == 1 and int(e.get("fault_total")[0]) == 1, e
            assert int(e.get("n_vel_clamp")[0]) >= 1 and int(e.get("vel_clamp_total")[0]) == int(e.get("n_vel_clamp")[0])
            rst = e.get("reset").astype(bool)
            assert rst[9] and not rst[5]                    # the non-finite env is reset, the fast one flies on
            r = e.get("root_states")
            assert np.isfinite(r).all() and np.isfinite(e.get("obs")[[5, 9]][:, 9:]).all()
            assert abs(np.linalg.norm(r[5, 7:10]) - lim) < 1e-2 * lim
        # position and linear velocity of the clamped env agree; its attitude / spin are rounding residue of 2e4 m/s velocity products
        np.testing.assert_allclose(hip.get("root_states")[5, [0, 1, 2, 7, 8, 9]], ora.get("root_states")[5, [0, 1, 2, 7, 8, 9]], rtol=1e-3, atol=1e-2)
        for e in (hip, ora):
            e.step(act)
        assert int(hip.get("n_fault")[0]) == 0 and int(hip.get("fault_total")[0]) == 1 and int(ora.get("fault_total")[0]) == 1
    finally:
        hip.close()
        ora.close()


@pytest.mark.gpu
@pytest.mark.parametrize("max_ang", [1000.0, 6.0])
def test_saturated_actions_never_trip_the_fault_guard(max_ang, oracle_built):
    """4096 flat ANYmal-C envs driven with |a| = 100 (the clip of LeggedRobot.step, legged_robot.py:86-87; the
    actuator-net path has no torque clip, anymal.py:71-81) for 200 policy steps.  The state stays finite, no env is stopped by the
    physics fault guard (the reference's PhysX clamps body velocities at asset.max_angular_velocity / max_linear_velocity and carries
    on, legged_robot.py:701-702), and the reset / time-out masks equal the oracle's on every step (the two trajectories are glued
    after every step: contacts amplify fp32 rounding).  With the asset's 1000 rad/s the clamp is a backstop; the second case lowers
    max_angular_velocity to 6 rad/s so that it acts on thousands of substeps, and HIP and oracle clamp the same ones."""
    import torch
    n, steps = 4096, 200
    cfg = harness.make_cfg("anymal_c_flat")
    cfg.env.num_envs = n
    cfg.asset.max_angular_velocity = max_ang
    from legged_gym_dev_amd.envs.base.env_setup import EnvSetup, sim_dt_float
    from legged_gym_dev_amd.model.robot_model import compile_model, resolve_model
    cm = compile_model(resolve_model("", "anymal_c"))

    def mk():
        return EnvSetup(cfg, cm, sim_dt_float(cfg.sim.dt), seed=3)
    hip, ora = harness.HipHandle(mk()), oracle_built.OracleEnv(mk())
    try:
        for e in (hip, ora):
            e.set_step_counter(0)
            e.inject(0)
            e.call("reset_all")
        rng = np.random.default_rng(11)
        A = 12
        resets = mism = 0
        wmax = 0.0
        glue = ["root_states", "dof_state", "lstm_h", "lstm_c", "last_dof_vel", "last_root_vel", "feet_air_time", "episode_sums",
                "last_actions", "last_contacts", "commands"]
        for t in range(steps):
            act = (100.0 * rng.choice([-1.0, 1.0], (n, A))).astype(np.float32)
            if t % 3 == 2:
                act = rng.uniform(-100, 100, (n, A)).astype(np.float32)
            hip.step(act)
            ora.step(act)
            r = hip.get("root_states")
            assert np.isfinite(r).all() and np.isfinite(hip.get("dof_state")).all() and np.isfinite(hip.get("obs")).all(), t
            wmax = max(wmax, float(np.linalg.norm(r[:, 10:13], axis=1).max()))
            assert int(hip.get("n_fault")[0]) == 0 and int(ora.get("n_fault")[0]) == 0, t
            mism += int((hip.get("reset") != ora.get("reset")).sum())
            np.testing.assert_array_equal(hip.get("time_out"), ora.get("time_out"), err_msg=f"step {t} time_out")
            np.testing.assert_array_equal(hip.get("episode_length")[hip.get("reset") == ora.get("reset")],
                                          ora.get("episode_length")[hip.get("reset") == ora.get("reset")])
            resets += int(ora.get("n_reset")[0])
            for key in glue:
                hip.set(key, ora.get(key))
            hip.set("reset", ora.get("reset")); hip.set("episode_length", ora.get("episode_length"))
        ch, co = int(hip.get("vel_clamp_total")[0]), int(ora.get("vel_clamp_total")[0])
        print(f"max_ang {max_ang}: resets {resets}, max base spin {wmax:.1f} rad/s, clamps hip {ch} oracle {co}, reset-mask mismatches {mism}")
        assert int(hip.get("fault_total")[0]) == 0 and int(ora.get("fault_total")[0]) == 0
        assert resets > n                                     # the robots do thrash: every env falls at least once on average
        # the termination mask is a threshold on a contact force (> 1 N on the base, legged_robot.py:139-145): of the 819 200 env-steps a
        # handful may sit within fp32 rounding of it
        assert mism <= 8, mism
        assert wmax <= max_ang * 1.001
        if max_ang < 100.0:
            assert ch > 1000 and abs(ch - co) <= max(8, ch // 200), (ch, co)
        else:
            assert ch == co
    finally:
        hip.close()
        ora.close()


def _product_env(task, n, rank=0, world=1, terrain=(4, 8), cmd=None):
    """The product env class through task_registry.make_env (per-env constants drawn for the GLOBAL env range, sliced per rank)."""
    import copy
    from legged_gym_dev_amd.envs import task_registry
    from legged_gym_dev_amd.utils import get_args
    args = get_args(["--task", task, "--num_envs", str(n), "--headless"])
    args.sim_device = args.rl_device = "cuda:0"
    env_cfg, _ = task_registry.get_cfgs(task)
    env_cfg = copy.deepcopy(env_cfg)
    env_cfg.env.num_envs = n
    if terrain and env_cfg.terrain.mesh_type in ("heightfield", "trimesh"):
        env_cfg.terrain.num_rows, env_cfg.terrain.num_cols = terrain
        env_cfg.terrain.border_size = 5
        env_cfg.terrain.max_init_terrain_level = terrain[0] - 1
    if cmd is not None:        # the fork's rough configs command zero velocity: the curriculum's move-down rule needs |cmd| > 0
        env_cfg.commands.ranges.lin_vel_x = env_cfg.commands.ranges.lin_vel_y = [-cmd, cmd]
    env, _ = task_registry.make_env(name=task, args=args, env_cfg=env_cfg, rank=rank, world_size=world)
    return env


def test_rough_terrain_shards_equal_the_unsharded_run():
    """BASELINE configs[3] in small: anymal_c_rough with the terrain curriculum, 256 envs as one context against 2 shards
    of 128 built the way a rank builds them (LeggedRobot(rank, world_size)).  terrain_types follow the reference rule on the
    GLOBAL index, floor(i / (N_total / num_cols)) (legged_robot.py:801-803); levels, origins, resets, curriculum moves and the
    height scans of the shards equal the unsharded run bit for bit over 40 policy steps with time-outs and falls."""
    import torch
    full = _product_env("anymal_c_rough", 256, cmd=1.0)
    sh = [_product_env("anymal_c_rough", 128, rank=r, world=2, cmd=1.0) for r in range(2)]
    try:
        t = full.core.t
        want_types = torch.div(torch.arange(256), 256 / 8, rounding_mode="floor").long()
        assert torch.equal(t["terrain_types"].cpu(), want_types)

        def cat(name):
            return torch.cat([s.core.t[name] for s in sh], 0)
        for name in ("terrain_types", "terrain_levels", "env_origins", "friction", "base_mass_delta"):
            assert torch.equal(cat(name), t[name]), name
        for e in [full] + sh:
            e.reset()
        g = torch.Generator(device="cuda").manual_seed(3)
        ep = torch.randint(0, 1001, (256,), device="cuda", generator=g)
        ep[:4] = torch.tensor([1000, 1001, 999, 500], device="cuda")
        full.episode_length_buf = ep
        sh[0].episode_length_buf, sh[1].episode_length_buf = ep[:128], ep[128:]
        moved = 0
        lv0 = t["terrain_levels"].clone()
        n_reset = 0
        for step in range(40):
            a = torch.randn(256, 12, device="cuda", generator=g) * (2.0 if step % 7 == 3 else 0.6)
            full.step(a)
            sh[0].step(a[:128].contiguous())
            sh[1].step(a[128:].contiguous())
            for name in ("obs", "rew", "reset", "time_out", "terrain_levels", "env_origins", "measured_heights", "root_states",
                         "dof_state", "commands", "episode_length"):
                assert torch.equal(cat(name), t[name]), f"step {step} {name}"
            n_reset += int(t["n_reset"][0])
            assert int(t["n_reset"][0]) == int(sh[0].core.t["n_reset"][0]) + int(sh[1].core.t["n_reset"][0])
        moved = int((t["terrain_levels"] != lv0).sum())
        assert n_reset >= 10 and moved > 0, (n_reset, moved)
        assert int(t["fault_total"][0]) == 0
    finally:
        full.close()
        for s in sh:
            s.close()


def test_baseline_config3_eight_shards_equal_unsharded_32768():
    """BASELINE.json configs[3] at FULL size on one GPU: anymal_c_rough, 32 768 envs on the reference's 10 x 20 tile terrain,
    built once as a single context and once as ranks 0..7 of 8 x 4096 the way a rank builds its shard
    (LeggedRobot(rank=r, world_size=8): per-env constants drawn for the global range and sliced, Philox keyed by the global
    env id).  terrain_types = floor(i / (N_total / num_cols)) on the GLOBAL index (legged_robot.py:801-803); levels, origins,
    friction and base mass equal slice by slice; 12 policy steps bit-equal for obs, rew, reset, time_out, terrain_levels,
    measured_heights and the state, with time-outs, falls and curriculum moves (envs displaced past half a tile move up and
    wrap to a random level at the top row); the physics fault guard never fires."""
    import torch
    W, n = 8, 4096
    N = W * n
    full = _product_env("anymal_c_rough", N, terrain=None)
    sh = []
    try:
        t = full.core.t
        assert full.terrain.heightsamples.shape == (1300, 2100) and full.cfg.terrain.num_cols == 20
        want_types = torch.div(torch.arange(N), N / 20, rounding_mode="floor").long()
        assert torch.equal(t["terrain_types"].cpu(), want_types)
        lv_init = t["terrain_levels"].clone()
        full.reset()
        g = torch.Generator(device="cuda").manual_seed(11)
        ep = torch.randint(0, 1001, (N,), device="cuda", generator=g)
        mover = (torch.arange(N, device="cuda") % 97) == 0              # walked 6 m: level up at its time-out
        top = (torch.arange(N, device="cuda") % 194) == 0               # ... half of them from the top row: wrap to a random level
        ep[mover] = 1001
        shift = torch.zeros(N, 13, device="cuda")
        shift[mover, 0] = 6.0
        full.episode_length_buf = ep
        t["root_states"].add_(shift)
        t["terrain_levels"][top] = 9
        lv0 = t["terrain_levels"].clone()
        acts = [torch.randn(N, 12, device="cuda", generator=g) * (2.0 if s % 5 == 3 else 0.6) for s in range(12)]
        names = ("obs", "rew", "reset", "time_out", "terrain_levels", "env_origins", "measured_heights", "root_states", "dof_state",
                 "commands", "episode_length", "contact_forces")
        want = []
        n_reset = 0
        for a in acts:
            full.step(a)
            want.append({k: t[k].clone() for k in names})
            n_reset += int(t["n_reset"][0])
        moved = int((t["terrain_levels"] != lv0).sum())
        wrapped = int((top & (t["terrain_levels"] != 9)).sum())
        assert n_reset >= 300 and moved >= 300 and wrapped > 100, (n_reset, moved, wrapped)
        assert int(t["fault_total"][0]) == 0
        consts = {k: t[k].clone() for k in ("terrain_types", "env_origins", "friction", "base_mass_delta")}
        for r in range(W):
            s = _product_env("anymal_c_rough", n, rank=r, world=W, terrain=None)
            sh.append(s)
            lo, hi = r * n, (r + 1) * n
            st = s.core.t
            assert int(s.setup.env_offset) == lo and int(s.setup.total_envs) == N
            assert torch.equal(st["terrain_types"], consts["terrain_types"][lo:hi]), f"rank {r} terrain_types"
            assert torch.equal(st["terrain_levels"], lv_init[lo:hi]), f"rank {r} initial levels"
            for k in ("friction", "base_mass_delta"):
                assert torch.equal(st[k], consts[k][lo:hi]), f"rank {r} {k}"
            s.reset()
            s.episode_length_buf = ep[lo:hi]
            st["root_states"].add_(shift[lo:hi])
            st["terrain_levels"][top[lo:hi]] = 9
            resets = 0
            for step, a in enumerate(acts):
                s.step(a[lo:hi].contiguous())
                for k in names:
                    assert torch.equal(st[k], want[step][k][lo:hi]), f"rank {r} step {step} {k}"
                resets += int(st["n_reset"][0])
            assert int(st["fault_total"][0]) == 0
            s.close()
            sh.pop()
    finally:
        full.close()
        for s in sh:
            s.close()


@pytest.mark.parametrize("task,z_lo,z_hi", [("anymal_c_rough", -3.0, 6.0), ("cassie", -3.0, 6.0)])
def test_rollout_properties_at_baseline_size_on_terrain(task, z_lo, z_hi):
    """BASELINE.json configs[2] / configs[4] sizes: 4096 envs on the reference's full 10 x 20 tile terrain (1300 x 2100
    height samples).  Size-independent properties of 60 policy steps with random actions: finite state, unit quaternions,
    reset => episode length 0 and zero joint rates, time_out => reset, observations within the clip, height scan within the
    terrain's range, robots stay above the lowest terrain point, the physics fault guard never fires, and two identically
    seeded contexts agree bit for bit."""
    import torch
    outs = []
    for rep in range(2):
        env = _product_env(task, 4096, terrain=None)
        try:
            t = env.core.t
            hs = env.terrain.heightsamples.astype(np.float64) * env.cfg.terrain.vertical_scale
            assert env.terrain.heightsamples.shape == (1300, 2100)
            env.reset()
            g = torch.Generator(device="cuda").manual_seed(0)
            for step in range(60):
                a = torch.randn(4096, env.num_actions, device="cuda", generator=g) * 0.5
                env.step(a)
                if step % 20 == 19 or step == 0:
                    root, dof = t["root_states"].cpu().numpy(), t["dof_state"].cpu().numpy()
                    rst, to = t["reset"].cpu().numpy().astype(bool), t["time_out"].cpu().numpy().astype(bool)
                    assert np.isfinite(root).all() and np.isfinite(dof).all() and bool(torch.isfinite(t["obs"]).all())
                    np.testing.assert_allclose(np.linalg.norm(root[:, 3:7], axis=1), 1.0, atol=1e-4)
                    assert (t["episode_length"].cpu().numpy()[rst] == 0).all() and (rst | ~to).all()
                    assert (dof[rst][..., 1] == 0).all()
                    assert float(t["obs"].abs().max()) <= 100.0
                    mh = t["measured_heights"].cpu().numpy()
                    assert hs.min() - 1e-6 <= mh.min() and mh.max() <= hs.max() + 1e-6
                    assert root[:, 2].min() > hs.min() - 0.5 and root[:, 2].max() < hs.max() + 3.0
            assert int(t["fault_total"][0]) == 0, int(t["fault_total"][0])
            lv = t["terrain_levels"].cpu().numpy()
            assert lv.min() >= 0 and lv.max() < env.cfg.terrain.num_rows
            outs.append((t["root_states"].cpu().numpy().copy(), t["obs"].cpu().numpy().copy(), lv.copy()))
        finally:
            env.close()
    for x, y in zip(outs[0], outs[1]):
        np.testing.assert_array_equal(x, y)


# ------------------------------------------------------------------------------------------------ trajectory-tracking variant
TRAJ = "anymal_c_flat_trajectory"


@pytest.mark.parametrize("name", [TRAJ, "anymal_c_flat_trajectory_curriculum", "anymal_c_rough_trajectory"])
def test_hip_replays_reference_trajectory_steps(name):
    """the HIP post-step with lg_cfg.traj enabled against the recorded steps of the reference's own
    LeggedRobotTrajectory / AnymalTrajectory / TrajectoryGenerator (tests/golden/anymal_c_flat_trajectory.npz): generator
    resamples (in the callback and on the reset loop's re-check), ROM steps, interpolated trajectory, per-env push timers,
    tracking_rom / differential_error through the generic term table, the 65-wide observation, resets with the random
    trajectory start offset.  ..._curriculum: the authors' staged curriculum (default.yaml:77-109) changes stage inside recorded
    steps 1 and 3 -- lg_set_curriculum_stage(in_callback=1) rewrites reward scales, tracking sigma, ROM input bounds, hold-time
    sampler and start-offset range so that the change step's callback still sees the old stage and its resets the new one.
    anymal_c_rough_trajectory: the registered rough-terrain task, 252 observations with the height scan."""
    z, meta = harness.load_fixture(name)
    setup, _ = harness.make_setup(name, z, meta)
    env = harness.HipHandle(setup, z["const_height_samples"] if "const_height_samples" in z.files else None)
    try:
        harness.replay_trajectory_fixture(env, z, meta)
    finally:
        env.close()


def test_trajectory_env_full_step_philox_matches_oracle(oracle_built):
    """lg_step of the trajectory env with its own Philox streams vs the oracle on the same seed over 30 policy steps (clocks
    scattered so that generator resamples, ROM steps, pushes, time-outs and resets all occur): masks, counters and the
    generator's discrete state bit-exact, fp32 state within tolerance (re-glued every step)."""
    z, meta = harness.load_fixture(TRAJ)
    cfg = harness.make_cfg(TRAJ)
    n = cfg.env.num_envs = 128
    from legged_gym_dev_amd.envs.base.env_setup import EnvSetup, sim_dt_float
    from legged_gym_dev_amd.model.robot_model import compile_model, resolve_model
    cm = compile_model(resolve_model("", "anymal_c"))

    def mk():
        return EnvSetup(cfg, cm, sim_dt_float(cfg.sim.dt), seed=17, extra_terms=harness.extra_terms_for(cfg))
    hip, ora = harness.HipHandle(mk()), oracle_built.OracleEnv(mk())
    try:
        rng = np.random.default_rng(6)
        tg = np.zeros((n, capi_TG_STRIDE()), np.float32)
        o = _tgf()
        tg[:, o["ramp_v_end"][0]:o["ramp_v_end"][0] + 2] = rng.uniform(-0.35, 0.35, (n, 2))
        for e in (hip, ora):
            e.set("tg_state", tg)
            e.set("push_timer", rng.uniform(0.0, 0.5, n).astype(np.float32) if e is hip else hip.get("push_timer"))
            e.set_step_counter(0)
            e.inject(0)
            e.call("reset_all")
        for key in ("tg_state", "tg_traj", "root_states", "dof_state", "prev_error"):
            np.testing.assert_allclose(hip.get(key), ora.get(key), rtol=1e-6, atol=1e-6, err_msg=f"after reset_all: {key}")
        ep = rng.integers(0, 1000, n)
        ep[:4] = [1000, 1001, 999, 3]
        for e in (hip, ora):
            e.set("episode_length", ep)
        seen = {"reset": 0, "pushed": 0, "resampled": 0, "rom": 0}
        for t in range(30):
            act = rng.uniform(-1, 1, (n, 12)).astype(np.float32)
            before = ora.get("tg_state").copy()
            pt_before = ora.get("push_timer").copy()
            hip.step(act)
            ora.step(act)
            for key in ("reset", "time_out", "episode_length"):
                np.testing.assert_array_equal(hip.get(key), ora.get(key), err_msg=f"step {t} {key}")
            assert int(hip.get("n_reset")[0]) == int(ora.get("n_reset")[0])
            gh, go = hip.get("tg_state"), ora.get("tg_state")
            for name in ("k", "stationary"):
                np.testing.assert_array_equal(gh[:, o[name][0]], go[:, o[name][0]], err_msg=f"step {t} generator {name}")
            np.testing.assert_allclose(gh, go, rtol=2e-5, atol=2e-5, err_msg=f"step {t} generator state")
            for key, tol in (("trajectory", 2e-5), ("tg_traj", 2e-5), ("prev_error", 2e-4), ("push_timer", 1e-6)):
                np.testing.assert_allclose(hip.get(key), ora.get(key), rtol=tol, atol=tol, err_msg=f"step {t} {key}")
            # what passed through four substeps of contact dynamics (its own per-substep parity test is
            # test_physics_substep_matches_oracle): 2e-3, a stray joint in a stick/slip transition up to 5e-2
            for key in ("obs", "rew", "root_states", "dof_state", "torques"):
                x, y = hip.get(key).astype(np.float64), ora.get(key).astype(np.float64)
                d = np.abs(x - y)
                bad = d > 2e-3 + 2e-3 * np.abs(y)
                assert bad.mean() < 2e-3 and d.max() < 5e-2 * max(1.0, np.abs(y).max()), f"step {t} {key}: {int(bad.sum())} of {bad.size}, max {d.max():.3g}"
            rst = ora.get("reset").astype(bool)
            seen["reset"] += int(rst.sum())
            seen["pushed"] += int((ora.get("push_timer") > pt_before).sum())
            seen["resampled"] += int(((go[:, o["t_final"][0]] != before[:, o["t_final"][0]]) & ~rst).sum())
            seen["rom"] += int(((go[:, o["k"][0]] != before[:, o["k"][0]]) & ~rst).sum())
            for key in ("root_states", "dof_state", "lstm_h", "lstm_c", "last_dof_vel", "last_root_vel", "feet_air_time",
                        "episode_sums", "tg_state", "tg_traj", "trajectory", "prev_error", "push_timer"):
                hip.set(key, ora.get(key))
        assert seen["reset"] > 5 and seen["pushed"] > 20 and seen["resampled"] > 10 and seen["rom"] > 500, seen
    finally:
        hip.close()
        ora.close()


def test_staged_curriculum_through_the_product_env(oracle_built):
    """cfg.curriculum.use_curriculum on the product classes (task_registry.make_env), base env and trajectory env: stage 0 is
    applied at construction, the stage moves on the steps the reference's rule names (common_step_counter % curriculum_steps[state]
    == 0), the Python-side attributes the reference rewrites (command_ranges, push_time, max_push_vel, reward_scales,
    tracking_sigma, rom.v_max, max_rom_distance, traj_gen.t_sampler) follow, and the constants in force on the device
    (lg_get_stage) are the stage's.  Base env with pushes ON (the reference's own push raises with the curriculum's list-valued
    max_push_vel, oracle/gen_fixtures.py): the scaled push period and magnitude drive real pushes here, HIP == oracle."""
    import copy
    import ctypes as C
    import torch
    from legged_gym_dev_amd import capi
    from legged_gym_dev_amd.envs import task_registry
    from legged_gym_dev_amd.utils import get_args

    def make(task, edit):
        args = get_args(["--task", task, "--num_envs", "128", "--headless"])
        args.sim_device = args.rl_device = "cuda:0"
        env_cfg, _ = task_registry.get_cfgs(task)
        env_cfg = copy.deepcopy(env_cfg)
        env_cfg.env.num_envs = 128
        edit(env_cfg)
        return task_registry.make_env(name=task, args=args, env_cfg=env_cfg)[0]

    def stage_of(env):
        st = capi.lg_stage()
        assert env.core.lib.lg_get_stage(env.core.ctx, C.byref(st)) == 0
        return st

    # ---- base env
    def edit_base(c):
        c.curriculum.use_curriculum, c.curriculum.curriculum_steps, c.curriculum.commands = True, [3, 6], [0.5, 0.75, 1]
        c.curriculum.push.magnitude, c.curriculum.push.time = [0.1, 0.5, 1], [3, 2, 1]
        c.commands.ranges.lin_vel_x, c.commands.ranges.lin_vel_y = [-1.0, 1.0], [-1.0, 1.0]
        c.commands.resampling_time = 0.04                      # every 2 policy steps
        c.domain_rand.push_interval_s = 0.039                  # nominal period ceil(0.039 / dt) = 2 steps -> 6 / 4 / 2 with the stage multipliers
    env = make("anymal_c_flat", edit_base)
    try:
        assert env.curriculum_state == 0 and env.push_time == 6.0 and abs(env.max_push_vel - 0.1) < 1e-12
        assert env.command_ranges["lin_vel_x"] == [-0.5, 0.5] and stage_of(env).push_time == 6.0
        env.reset_idx(torch.arange(128, device="cuda:0"))
        states, pushed, cmd_max = [], [], []
        g = torch.Generator(device="cuda").manual_seed(2)
        for k in range(12):
            v0 = env.root_states[:, 7:9].clone()
            env.step(torch.randn(128, 12, device="cuda", generator=g) * 0.3)
            states.append(env.curriculum_state)
            cmd_max.append(float(env.commands[:, :2].abs().max()))
            st = stage_of(env)
            assert st.push_time == env.push_time and abs(st.max_push_vel - env.max_push_vel) < 1e-6
            assert abs(st.cmd_hi[0] - env.command_ranges["lin_vel_x"][1]) < 1e-6
        # counter 1..12: stage 1 at counter 3, stage 2 at counter 6
        assert states == [0, 0, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2], states
        assert env.push_time == 2.0 and env.max_push_vel == 1.0 and env.command_ranges["lin_vel_x"] == [-1.0, 1.0]
        assert max(cmd_max[:2]) <= 0.5 + 1e-6 and max(cmd_max[:5]) <= 0.75 + 1e-6 and max(cmd_max) > 0.75
    finally:
        env.close()

    # ---- trajectory env
    def edit_traj(c):
        c.curriculum.use_curriculum, c.curriculum.curriculum_steps = True, [3, 6]
        c.rewards.scales.tracking_rom, c.rewards.scales.differential_error = 6.0, -1.5
        c.domain_rand.randomize_rom_distance, c.domain_rand.max_rom_dist = True, [0.4, 0.2]
        c.curriculum.max_rom_distance = [0.5, 0.75, 1.0]
    env = make("anymal_c_flat_trajectory", edit_traj)
    try:
        dt = env.dt
        assert env.curriculum_state == 0 and abs(env.tracking_sigma - 0.25) < 1e-12 and float(env.rom.v_max[0]) == np.float32(0.35 * 0.5)
        assert env.traj_gen.t_sampler.t_low == 3 and env.traj_gen.t_sampler.t_high == 6
        g = torch.Generator(device="cuda").manual_seed(3)
        for k in range(7):
            env.step(torch.randn(128, 12, device="cuda", generator=g) * 0.3)
        assert env.curriculum_state == 2
        assert abs(env.tracking_sigma - 0.25 * 0.6) < 1e-12 and float(env.rom.v_max[0]) == np.float32(0.35)
        assert abs(env.reward_scales["tracking_rom"] - 6.0 * 0.6 * dt) < 1e-12
        assert abs(env.reward_scales["termination"] - (-0.5) * 0.6 * dt) < 1e-12
        st = stage_of(env)
        row = env.setup.xterm_names.index("tracking_rom")
        assert abs(st.xterm_p0[row] - 0.25 * 0.6) < 1e-7 and abs(st.xterm_scale[row] - 6.0 * 0.6 * dt) < 1e-7
        assert abs(st.traj_t_low - 1.0) < 1e-7 and abs(st.traj_v_max[0] - 0.35) < 1e-7 and abs(st.traj_max_rom_dist[0] - 0.4) < 1e-7
        assert abs(st.rew_scale[capi.REWARD_NAMES.index("termination")] - (-0.5) * 0.6 * dt) < 1e-8
        assert bool(torch.isfinite(env.obs_buf).all()) and int(env.fault_total[0]) == 0
    finally:
        env.close()

    # ---- pushes under the stage's period and magnitude, HIP == oracle on the same Philox streams
    from legged_gym_dev_amd.envs.base.env_setup import CurriculumClock, EnvSetup, sim_dt_float
    from legged_gym_dev_amd.model.robot_model import compile_model, resolve_model
    cfg = harness.make_cfg("anymal_c_flat")
    n = cfg.env.num_envs = 64
    edit_base(cfg)
    cm = compile_model(resolve_model("", "anymal_c"))

    def mk():
        return EnvSetup(cfg, cm, sim_dt_float(cfg.sim.dt), seed=9)
    hip, ora = harness.HipHandle(mk()), oracle_built.OracleEnv(mk())
    try:
        clocks = [CurriculumClock(hip.setup), CurriculumClock(ora.setup)]
        for e in (hip, ora):
            harness.set_stage(e, e.setup, 0, in_callback=False)
            e.set_step_counter(0)
            e.inject(0)
            e.call("reset_all")
        rng = np.random.default_rng(4)
        pushed_at = []
        for k in range(1, 13):
            for e, clk in zip((hip, ora), clocks):
                if clk.tick(k):
                    harness.set_stage(e, e.setup, clk.state, in_callback=True)
            act = rng.uniform(-0.3, 0.3, (n, 12)).astype(np.float32)
            hip.step(act)
            ora.step(act)
            np.testing.assert_array_equal(hip.get("reset"), ora.get("reset"))
            np.testing.assert_allclose(hip.get("commands"), ora.get("commands"), rtol=1e-6, atol=1e-6, err_msg=f"step {k} commands")
            rv_h, rv_o = hip.get("root_states")[:, 7:9], ora.get("root_states")[:, 7:9]
            live = ~ora.get("reset").astype(bool)
            np.testing.assert_allclose(rv_h[live], rv_o[live], rtol=2e-3, atol=2e-3, err_msg=f"step {k} base velocity")
            for key in ("root_states", "dof_state", "lstm_h", "lstm_c", "last_dof_vel", "last_root_vel", "feet_air_time", "episode_sums"):
                hip.set(key, ora.get(key))
        # the push of step k is drawn from +-max_push_vel of the stage in force when the step began; periods 6, then 4, then 2:
        # counter 6 is a multiple of the old period 4? no (6 % 4 = 2) -- pushes at 4 (stage 1, period 4), then 8, 10, 12 (period 2)
        assert clocks[0].state == 2
    finally:
        hip.close()
        ora.close()


def capi_TG_STRIDE():
    from legged_gym_dev_amd import capi
    return capi.TG_STRIDE


def _tgf():
    from legged_gym_dev_amd import capi
    return capi.TG_FIELDS


def test_declared_extra_term_equals_the_builtin_it_mirrors(oracle_built):
    """The extension point for reward terms (the reference's _reward_<name> methods, legged_robot.py:605-629): a term declared
    as data -- exp(-|cmd_xy - v_xy|^2 / sigma) under the name tracking_xy, plus a weighted square of the projected gravity --
    must produce the values of the builtin tracking_lin_vel / orientation terms, get its own episode sum and extras entry,
    and take its alphabetical place in the reward sum.  HIP == oracle, and extra == builtin within fp32 rounding."""
    from legged_gym_dev_amd.envs.base.env_setup import EnvSetup, sim_dt_float
    from legged_gym_dev_amd.envs.base.reward_terms import ExpNegWeightedSqErr, WeightedSq
    from legged_gym_dev_amd.model.robot_model import compile_model, resolve_model
    cfg = harness.make_cfg("anymal_c_allrewards")
    n = cfg.env.num_envs = 96
    cfg.rewards.scales.tracking_xy = 1.0          # same scale as tracking_lin_vel
    cfg.rewards.scales.gravity_xy = -0.5          # same scale as orientation
    terms = {"tracking_xy": ExpNegWeightedSqErr("commands", "base_lin_vel", weights=[1.0, 1.0], sigma=cfg.rewards.tracking_sigma),
             "gravity_xy": WeightedSq("projected_gravity", weights=[1.0, 1.0])}
    cm = compile_model(resolve_model("", "anymal_c"))

    def mk():
        return EnvSetup(cfg, cm, sim_dt_float(cfg.sim.dt), seed=3, extra_terms=terms)
    setup = mk()
    assert setup.xterm_names == ["tracking_xy", "gravity_xy"]
    names = sorted(k for k in setup.reward_scales if k != "termination")
    assert [setup.term_row[k] for k in names] == setup.term_order and names.index("tracking_xy") == names.index("tracking_lin_vel") + 1
    hip, ora = harness.HipHandle(setup), oracle_built.OracleEnv(mk())
    try:
        rng = np.random.default_rng(1)
        for e in (hip, ora):
            e.set_step_counter(0)
            e.inject(0)
            e.call("reset_all")
        for t in range(5):
            act = rng.uniform(-1, 1, (n, 12)).astype(np.float32)
            hip.step(act)
            ora.step(act)
            np.testing.assert_allclose(hip.get("rew"), ora.get("rew"), rtol=2e-3, atol=2e-3)
            np.testing.assert_allclose(hip.get("episode_sums"), ora.get("episode_sums"), rtol=2e-3, atol=2e-3)
            es = hip.get("episode_sums")
            row = setup.term_row
            live = ~hip.get("reset").astype(bool)
            np.testing.assert_allclose(es[row["tracking_xy"]][live], es[row["tracking_lin_vel"]][live], rtol=1e-5, atol=1e-7)
            np.testing.assert_allclose(es[row["gravity_xy"]][live], es[row["orientation"]][live], rtol=1e-5, atol=1e-7)
            assert np.abs(es[row["tracking_xy"]]).sum() > 0
            for key in ("root_states", "dof_state", "last_dof_vel", "last_root_vel", "feet_air_time", "episode_sums", "commands"):
                hip.set(key, ora.get(key))
        np.testing.assert_allclose(hip.get("extras_episode"), ora.get("extras_episode"), rtol=1e-3, atol=1e-6)
    finally:
        hip.close()
        ora.close()


def test_trajectory_task_trains_through_the_registry(tmp_path):
    """anymal_c_flat_trajectory as a user runs it: task_registry.make_env builds AnymalTrajectory (65 observations, reference
    attribute names), the runner trains two iterations with the authors' tracking reward, a caller resets a subset."""
    import copy
    import torch
    from legged_gym_dev_amd.envs import task_registry
    from legged_gym_dev_amd.utils import get_args
    from legged_gym_dev_amd.utils.helpers import class_to_dict
    from legged_gym_dev_amd.rl.runner import OnPolicyRunner
    args = get_args(["--task", TRAJ, "--num_envs", "256", "--headless"])
    args.sim_device = args.rl_device = "cuda:0"
    env_cfg, train_cfg = task_registry.get_cfgs(TRAJ)
    env_cfg, train_cfg = copy.deepcopy(env_cfg), copy.deepcopy(train_cfg)
    env_cfg.rewards.scales.tracking_rom = 6.0
    env_cfg.rewards.scales.differential_error = -0.5
    env, _ = task_registry.make_env(name=TRAJ, args=args, env_cfg=env_cfg)
    try:
        assert type(env).__name__ == "AnymalTrajectory" and env.num_obs == 65 and not hasattr(env, "commands")
        assert env.trajectory.shape == (256, 10, 2) and env.prev_error.shape == (256, 2) and env.time_until_next_push.shape == (256, 1)
        assert float(env.time_until_next_push.min()) >= 0.5 and float(env.time_until_next_push.max()) <= 10.0
        assert set(env.episode_sums) == {"differential_error", "feet_air_time", "orientation", "termination", "torques", "tracking_rom"}
        runner = OnPolicyRunner(env, class_to_dict(train_cfg), None, device="cuda:0")
        runner.learn(2, init_at_random_ep_len=True)
        torch.cuda.synchronize()
        assert bool(torch.isfinite(runner.ppo.t["params"]).all()) and bool(torch.isfinite(env.obs_buf).all())
        # the trajectory block of the observation is the interpolated window relative to the robot (noise scale 0 there)
        want = (env.trajectory - env.root_states[:, None, :2]).reshape(256, 20)
        live = ~env.reset_buf                      # reset envs observe the pre-reset window against the new pose (reference quirk)
        torch.testing.assert_close(env.obs_buf[live, 9:29], want[live].clamp(-100, 100), rtol=1e-5, atol=1e-6)
        assert float(env.traj_gen.t.min()) > -1.01 and float(env.traj_gen.weights.sum(1).sub(1).abs().max()) < 1e-5
        assert "rew_tracking_rom" in env.extras["episode"] and int(env.fault_total[0]) == 0
        env.reset_idx(torch.tensor([3, 5, 200], device="cuda:0"))
        torch.cuda.synchronize()
        assert (env.episode_length_buf[[3, 5, 200]] == 0).all() and float(env.traj_gen.k[[3, 5, 200]].abs().max()) == 0.0
        runner.ppo.close()
    finally:
        env.close()


class _HipBackend:
    """What tests/test_physics_oracle.py calls `oracle_built`, backed by the HIP library instead: the invariants below then check the
    product physics against physics itself (a float64 mass-matrix formulation, conservation laws, statics), with no oracle in between."""

    @staticmethod
    def OracleEnv(setup, height_samples=None):
        return harness.HipHandle(setup, height_samples)


@pytest.mark.parametrize("robot", ["anymal_c", "cassie", "a1"])
def test_hip_free_dynamics_match_mass_matrix_formulation(robot):
    """One substep of lg_simulate in free flight == M(q) nu_dot + h(q, nu) = tau built from link Jacobians in float64 (an independent
    formulation: no articulated-body recursion), for the quadruped and the biped topology kernels."""
    from tests import test_physics_oracle as tpo
    tpo.test_free_dynamics_match_mass_matrix_formulation(robot, _HipBackend)


def test_hip_momentum_conserved_without_gravity():
    from tests import test_physics_oracle as tpo
    tpo.test_momentum_conserved_without_gravity(_HipBackend)


def test_hip_momentum_conserved_with_joints_at_their_velocity_limit():
    """Round 4's solver fix through lg_simulate: knees saturated at their velocity limit under 60 N m do not spin the base up."""
    from tests import test_physics_oracle as tpo
    tpo.test_momentum_conserved_with_joints_at_their_velocity_limit(_HipBackend)


@pytest.mark.parametrize("robot", ["anymal_c", "cassie"])
def test_hip_armature_adds_to_the_joint_space_inertia(robot):
    """cfg.asset.armature through lg_simulate against the float64 mass-matrix formulation with armature on the joint diagonal."""
    from tests import test_physics_oracle as tpo
    tpo.test_armature_adds_to_the_joint_space_inertia(robot, _HipBackend)


@pytest.mark.parametrize("robot,height", [("anymal_c", 0.56), ("cassie", 0.95), ("a1", 0.36)])
def test_hip_static_stance_supports_weight(robot, height):
    """PD-held stance on the plane through lg_compute_torques + lg_simulate: the vertical contact forces add up to the robot's weight,
    every foot carries load, the base stays put."""
    from tests import test_physics_oracle as tpo
    tpo.test_static_stance_supports_weight(robot, height, _HipBackend)


@pytest.mark.parametrize("robot", ["a1", "cassie"])
def test_hip_joint_limits_hold_against_torque(robot):
    from tests import test_physics_oracle as tpo
    tpo.test_joint_limits_hold_against_torque(robot, _HipBackend)


def test_hip_shape_material_parameters_act_on_the_contacts():
    """Restitution / compliance / thickness of the randomised rigid-shape properties (legged_robot.py:284-299) as the HIP contact model carries them."""
    from tests import test_physics_oracle as tpo
    tpo.test_shape_material_parameters_act_on_the_contacts(_HipBackend)


@pytest.mark.parametrize("task", ["anymal_c_flat", "anymal_c_rough", "cassie", "anymal_c_flat_trajectory"])
def test_product_env_matches_oracle_at_baseline_size(task, oracle_built):
    """BASELINE.json configs[1], [2], [4] at their FULL size: the product env (task_registry.make_env: 4096 envs, for the rough tasks the
    reference's 10 x 20 tile terrain of 1300 x 2100 height samples) against the oracle built from the same EnvSetup and height field,
    with the product's per-env constants copied over.  reset() and then 6 policy steps on the built-in Philox streams with episode
    clocks scattered over time-outs, command resamples and the push period: reset / time-out masks, episode lengths, terrain levels and
    the reset count bit-exact on all 4096 envs, fp32 state / observations / rewards / height scans within the per-step tolerance (the
    two trajectories are glued together after every step: contacts amplify fp32 rounding)."""
    import torch
    n = 4096
    env = _product_env(task, n, terrain=None)
    ora = oracle_built.OracleEnv(env.setup, env.terrain.heightsamples if env.terrain is not None else None)
    hip = harness.HipHandle.__new__(harness.HipHandle)           # the adapter around the product env's own context
    hip.torch, hip.core, hip.setup, hip._act = torch, env.core, env.setup, None
    try:
        for key in ora.buf:                                       # whatever the product env's constructor put into its context
            if key in hip.core.t:
                ora.set(key, hip.get(key))
        rng = np.random.default_rng(7)
        A = env.num_actions
        for e in (hip, ora):
            e.set_step_counter(0)
            e.inject(0)
            e.set_init_done(1)
            e.call("reset_all")
        np.testing.assert_allclose(hip.get("root_states"), ora.get("root_states"), rtol=1e-6, atol=1e-6)
        np.testing.assert_array_equal(hip.get("terrain_levels"), ora.get("terrain_levels"))
        ep = rng.integers(0, int(env.max_episode_length), n)
        ep[:8] = [199, 499, int(env.max_episode_length) - 1, int(env.max_episode_length), int(env.max_episode_length) + 1, 399, 0, 1]
        push = int(env.setup.push_time) if env.setup.push_time else 0
        for e in (hip, ora):
            e.set("episode_length", ep)
            e.set_step_counter(max(push - 3, 0))                  # a push falls inside the window where the task pushes
        resets = 0
        for t in range(6):
            act = rng.uniform(-1, 1, (n, A)).astype(np.float32)
            hip.step(act)
            ora.step(act)
            np.testing.assert_array_equal(hip.get("reset"), ora.get("reset"), err_msg=f"step {t} reset")
            np.testing.assert_array_equal(hip.get("time_out"), ora.get("time_out"), err_msg=f"step {t} time_out")
            np.testing.assert_array_equal(hip.get("episode_length"), ora.get("episode_length"))
            np.testing.assert_array_equal(hip.get("terrain_levels"), ora.get("terrain_levels"), err_msg=f"step {t} terrain_levels")
            assert int(hip.get("n_reset")[0]) == int(ora.get("n_reset")[0])
            resets += int(ora.get("n_reset")[0])
            keys = [("obs", 2e-3), ("rew", 2e-3), ("root_states", 1e-3), ("dof_state", 2e-3), ("commands", 1e-6), ("torques", 5e-3)]
            if env.setup.num_height_points:
                keys.append(("measured_heights", 1e-6))
            if env.setup.traj:                                     # the trajectory generator's state and the tracked window
                keys += [("trajectory", 1e-5), ("tg_traj", 1e-5), ("prev_error", 1e-4), ("push_timer", 1e-6)]
            for key, tol in keys:
                a, b = hip.get(key), ora.get(key)
                bad = ~np.isclose(a, b, rtol=tol, atol=tol)
                # a contact that opens or closes one substep apart moves a joint rate by more than the band: a handful of envs per step
                assert bad.reshape(n, -1).any(1).sum() <= (0 if key in ("commands", "measured_heights", "push_timer") else 8), (t, key, int(bad.sum()))
            glue = ["root_states", "dof_state", "lstm_h", "lstm_c", "last_dof_vel", "last_root_vel", "feet_air_time", "episode_sums",
                    "last_actions", "last_contacts", "commands"]
            if env.setup.traj:
                glue += ["tg_state", "tg_traj", "trajectory", "prev_error", "push_timer"]
            for key in glue:
                hip.set(key, ora.get(key))
        assert resets > 0 and int(hip.get("fault_total")[0]) == 0
    finally:
        ora.close()
        env.close()
