"""Shared helpers for the parity tests: rebuild the fixture's configuration with THIS repo's
config classes, replay a golden fixture (tests/golden/*.npz, made by oracle/gen_fixtures.py from
the reference's own Python) through an env handle (CPU oracle or HIP), and compare."""
import json
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from legged_gym_dev_amd import capi  # noqa: E402
from legged_gym_dev_amd.envs.base.env_setup import EnvSetup, sim_dt_float  # noqa: E402
from legged_gym_dev_amd.model.robot_model import compile_model, resolve_model  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
FIXTURES = ["anymal_c_flat", "anymal_c_rough", "cassie", "anymal_c_allrewards", "anymal_c_pd_V", "anymal_c_pd_T", "a1", "anymal_b",
            "anymal_c_flat_curriculum", "anymal_c_randomised", "anymal_c_yawcmd"]


def load_fixture(name):
    z = np.load(os.path.join(GOLDEN, f"{name}.npz"))
    meta = json.loads(bytes(z["meta_json"]).decode())
    return z, meta


def _small_terrain(cfg):
    cfg.terrain.num_rows, cfg.terrain.num_cols, cfg.terrain.border_size = 3, 5, 5
    cfg.terrain.max_init_terrain_level = 2


def make_cfg(name):
    """Mirror of the per-case overrides in oracle/gen_fixtures.py main(), on our config classes."""
    from legged_gym_dev_amd.envs.anymal_c.flat.anymal_c_flat_config import AnymalCFlatCfg
    from legged_gym_dev_amd.envs.anymal_c.mixed_terrains.anymal_c_rough_config import AnymalCRoughCfg
    from legged_gym_dev_amd.envs.cassie.cassie_config import CassieRoughCfg
    if name == "anymal_c_flat":
        cfg = AnymalCFlatCfg()
        cfg.env.num_envs = 64
    elif name == "anymal_c_flat_curriculum":              # staged command curriculum, stage changes inside recorded steps 1 and 3
        cfg = AnymalCFlatCfg()
        cfg.env.num_envs = 64
        cfg.commands.ranges.lin_vel_x, cfg.commands.ranges.lin_vel_y = [-1.0, 1.0], [-0.5, 1.5]
        cfg.curriculum.use_curriculum, cfg.curriculum.curriculum_steps, cfg.curriculum.commands = True, [2252, 2254], [0.5, 0.75, 1]
        cfg.domain_rand.max_push_vel = [cfg.domain_rand.max_push_vel_xy]      # the list form the reference's curriculum needs
        cfg.domain_rand.push_robots = False                                    # (and with which its _push_robots raises)
    elif name == "anymal_c_randomised":                   # every randomisation of the property callbacks on
        cfg = AnymalCFlatCfg()
        cfg.env.num_envs = 32
        cfg.domain_rand.randomize_base_mass, cfg.domain_rand.added_mass_range = True, [-5.0, 5.0]
        cfg.domain_rand.randomize_inv_base_mass = True
        rsp = cfg.domain_rand.rigid_shape_properties
        rsp.randomize_restitution = rsp.randomize_compliance = rsp.randomize_thickness = True
    elif name == "anymal_c_yawcmd":                       # yaw-rate commands, no observation noise, decimation 2
        cfg = AnymalCFlatCfg()
        cfg.env.num_envs = 32
        cfg.commands.heading_command = False
        cfg.commands.ranges.lin_vel_x, cfg.commands.ranges.lin_vel_y, cfg.commands.ranges.ang_vel_yaw = [-1.0, 1.0], [-0.5, 0.5], [-1.5, 1.5]
        cfg.commands.resampling_time = 0.1
        cfg.control.decimation = 2
        cfg.noise.add_noise = False
        cfg.rewards.scales.tracking_lin_vel, cfg.rewards.scales.tracking_ang_vel = 1.0, 0.5
    elif name == "anymal_c_rough":
        cfg = AnymalCRoughCfg()
        cfg.env.num_envs = 64
        _small_terrain(cfg)
    elif name == "cassie":
        cfg = CassieRoughCfg()
        cfg.env.num_envs = 64
        _small_terrain(cfg)
    elif name == "anymal_c_allrewards":
        cfg = AnymalCFlatCfg()
        cfg.env.num_envs = 64
        cfg.control.use_actuator_network = False
        cfg.commands.heading_command = True
        cfg.commands.ranges.lin_vel_x = [-1.0, 1.0]
        cfg.commands.ranges.lin_vel_y = [-1.0, 1.0]
        cfg.commands.ranges.heading = [-3.14, 3.14]
        cfg.rewards.only_positive_rewards = False
        cfg.rewards.soft_dof_vel_limit = 0.5
        cfg.rewards.soft_torque_limit = 0.5
        for k, v in dict(termination=-3.0, tracking_lin_vel=1.0, tracking_ang_vel=0.5, lin_vel_z=-2.0,
                         ang_vel_xy=-0.05, orientation=-0.5, torques=-1e-5, dof_vel=-1e-3, dof_acc=-2.5e-7,
                         base_height=-1.0, feet_air_time=1.0, collision=-1.0, stumble=-0.5, action_rate=-0.01,
                         stand_still=-0.1, dof_pos_limits=-1.0, dof_vel_limits=-0.3, torque_limits=-0.2,
                         feet_contact_forces=-0.01).items():
            setattr(cfg.rewards.scales, k, v)
        cfg.terrain.measure_heights = True
        cfg.env.num_observations = 235
    elif name == "a1":
        from legged_gym_dev_amd.envs.a1.a1_config import A1RoughCfg
        cfg = A1RoughCfg()
        cfg.env.num_envs = 32
        _small_terrain(cfg)
    elif name == "anymal_b":
        from legged_gym_dev_amd.envs.anymal_b.anymal_b_config import AnymalBRoughCfg
        cfg = AnymalBRoughCfg()
        cfg.env.num_envs = 32
        _small_terrain(cfg)
    elif name in ("anymal_c_flat_trajectory", "anymal_c_flat_trajectory_curriculum", "anymal_c_rough_trajectory"):
        # the repairs + reward table of oracle/gen_fixtures_trajectory.py::patch_cfg, on this repo's config class
        if name == "anymal_c_rough_trajectory":
            from legged_gym_dev_amd.envs.anymal_c.mixed_terrains_trajectory.anymal_c_rough_trajectory_config import AnymalCRoughTrajectoryCfg
            cfg = AnymalCRoughTrajectoryCfg()
            _small_terrain(cfg)
            cfg.terrain.curriculum = False          # with it on the reference's first reset raises (gen_fixtures_trajectory.py)
        else:
            from legged_gym_dev_amd.envs.anymal_c.flat_trajectory.anymal_c_flat_trajectory_config import AnymalCFlatTrajectoryCfg
            cfg = AnymalCFlatTrajectoryCfg()
        cfg.env.num_envs = 64
        if name.endswith("_curriculum"):            # gen_fixtures_trajectory.py::CURRICULUM (the other rows are the config's own)
            cur = cfg.curriculum
            cur.use_curriculum, cur.curriculum_steps = True, [2, 4]
            cur.max_rom_distance, cur.zero_rom_distance_likelihood = [0.5, 0.75, 1.0], [1.0, 2.0, 3.0]
        cfg.domain_rand.randomize_rom_distance = True
        cfg.domain_rand.max_rom_dist = [0.3, 0.2]
        cfg.domain_rand.zero_rom_distance_likelihood = 0.25
        for k, v in dict(termination=-0.5, tracking_rom=6.0, differential_error=-1.5, ang_vel_xy=-0.05, orientation=-1.0,
                         torques=-1e-5, dof_acc=-2.5e-7, collision=-1.0, action_rate=-0.1, feet_air_time=0.5).items():
            setattr(cfg.rewards.scales, k, v)
    elif name.startswith("anymal_c_pd_"):
        cfg = AnymalCFlatCfg()
        cfg.env.num_envs = 32
        cfg.control.use_actuator_network = False
        cfg.control.control_type = name[-1]
    else:
        raise KeyError(name)
    return cfg


class FixtureTerrain:
    """Terrain stand-in carrying the fixture's height samples / tile origins."""

    def __init__(self, z, meta, cfg):
        hs = z["const_height_samples"]
        self.heightsamples = self.height_field_raw = hs
        self.tot_rows, self.tot_cols = hs.shape
        self.env_length = meta["terrain_env_length"]
        self.env_origins = z["const_terrain_origins"]
        self.cfg = cfg.terrain


def extra_terms_for(cfg):
    """The extra reward terms the registered env class of this cfg declares (without constructing the env)."""
    if hasattr(cfg, "trajectory_generator"):
        from legged_gym_dev_amd.envs.base.legged_robot_trajectory import LeggedRobotTrajectory
        return LeggedRobotTrajectory.extra_reward_terms(types.SimpleNamespace(cfg=cfg))
    return {}


def make_setup(name, z, meta):
    cfg = make_cfg(name)
    cm = compile_model(resolve_model("", meta["robot"]))
    terrain = FixtureTerrain(z, meta, cfg) if "const_height_samples" in z.files else None
    return EnvSetup(cfg, cm, sim_dt_float(cfg.sim.dt), terrain=terrain, seed=1, extra_terms=extra_terms_for(cfg)), cfg


def check_setup_against_fixture(setup, z, meta):
    """Host logic pinned against what the reference derived (feet indices, gains, limits, ...)."""
    assert setup.feet_indices == z["const_feet_indices"].tolist()
    assert setup.penalised_contact_indices == z["const_penalised_contact_indices"].tolist()
    assert setup.termination_contact_indices == z["const_termination_contact_indices"].tolist()
    np.testing.assert_array_equal(setup.default_dof_pos, z["const_default_dof_pos"])
    np.testing.assert_array_equal(setup.p_gains, z["const_p_gains"])
    np.testing.assert_array_equal(setup.d_gains, z["const_d_gains"])
    np.testing.assert_array_equal(setup.dof_pos_limits, z["const_dof_pos_limits"])
    np.testing.assert_array_equal(setup.dof_vel_limits, z["const_dof_vel_limits"])
    np.testing.assert_array_equal(setup.torque_limits, z["const_torque_limits"])
    np.testing.assert_array_equal(setup.noise_scale_vec, z["const_noise_scale_vec"])
    assert list(setup.reward_scales.keys()) == meta["reward_names"]
    np.testing.assert_array_equal(np.array([setup.reward_scales[k] for k in meta["reward_names"]]),
                                  z["const_reward_scales"])
    assert setup.dt == meta["dt"]
    assert setup.max_episode_length == meta["max_episode_length"]
    if "push_time" in meta:          # (with the staged curriculum on, _parse_cfg has already applied stage 0: legged_robot.py:828-829)
        sv0 = setup.stage_values(0 if setup.use_curriculum else None)
        assert sv0["push_time"] == meta["push_time"] and sv0["max_push_vel"] == meta["max_push_vel"]
        assert setup.use_curriculum == bool(meta.get("use_curriculum", False))
    if "resample_steps" in meta:
        assert int(setup.cfg.commands.resampling_time / setup.dt) == meta["resample_steps"]
    if setup.measure_heights:
        np.testing.assert_array_equal(setup.height_points, z["const_height_points"][:, :2])
    if setup.traj is not None:                              # what _init_rom / _init_trajectory_generator / _init_buffers derived
        tj = setup.traj
        # (with the staged curriculum on, the constructor has already applied stage 0: legged_robot_trajectory.py:78-79)
        sv = setup.stage_values(0 if setup.use_curriculum else None)
        np.testing.assert_array_equal(np.float32(sv["v_min"]), z["const_rom_v_min"])
        np.testing.assert_array_equal(np.float32(sv["v_max"]), z["const_rom_v_max"])
        np.testing.assert_array_equal(np.tile(np.float32(tj["obs_scale"]), (tj["N"], 1)), z["const_trajectory_scale"])
        np.testing.assert_array_equal(np.float32(sv["max_rom_dist"]), z["const_max_rom_distance"])
        assert (tj["N"], tj["dN"], tj["rom_dt"]) == (meta["traj_N"], meta["traj_dN"], meta["rom_dt"])
        assert (sv["t_low"], sv["t_high"], tj["freq_low"], tj["freq_high"]) == (meta["t_low"], meta["t_high"], meta["freq_low"], meta["freq_high"])
        assert tj["prob_stationary"] == meta["prob_stationary"] and tj["zero_rom_dist_llh"] == meta["zero_rom_dist_llh"]
        assert tj["push_t"] == meta["time_between_pushes"] and tj["max_push_vel_xy"] == meta["max_push_vel_xy"]
        xt = setup.extra_terms
        np.testing.assert_array_equal(np.float32(xt["tracking_rom"].w), z["const_reward_weighting"])
        assert sv["tracking_sigma"] == meta["tracking_sigma"]
        assert (xt["differential_error"].neg, xt["differential_error"].pos) == (meta["neg_slope"], meta["pos_slope"])
        assert capi.tslots(setup.num_dof)["noise"] + meta["num_obs"] == meta["slots"]["K"]
        assert {k: v for k, v in capi.tslots(setup.num_dof).items()} == {k: v for k, v in meta["slots"].items() if k != "K"}


def load_state(env, z, prefix, meta):
    """Install a fixture snapshot (init_ or sK_post_) into the env handle's buffers."""
    names = meta["reward_names"]
    env.set("root_states", z[prefix + "root_states"])
    env.set("dof_state", z[prefix + "dof_state"])
    env.set("commands", z[prefix + "commands"])
    env.set("last_actions", z[prefix + "last_actions"])
    env.set("last_dof_vel", z[prefix + "last_dof_vel"])
    env.set("last_root_vel", z[prefix + "last_root_vel"])
    env.set("feet_air_time", z[prefix + "feet_air_time"])
    env.set("last_contacts", z[prefix + "last_contacts"].astype(np.uint8))
    env.set("episode_length", z[prefix + "episode_length_buf"])
    es = np.zeros((capi.NUM_TERMS, meta["num_envs"]), np.float32)
    for k, n in enumerate(names):
        es[env.setup.term_row[n]] = z[prefix + "episode_sums"][:, k]
    env.set("episode_sums", es)
    env.set("env_origins", z[prefix + "env_origins"])
    if prefix + "terrain_levels" in z.files:
        env.set("terrain_levels", z[prefix + "terrain_levels"])
        env.set("terrain_types", z["const_terrain_types"])
    if meta["use_lstm"]:
        env.set("lstm_h", z[prefix + "lstm_h"])
        env.set("lstm_c", z[prefix + "lstm_c"])


TOL = dict(rtol=2e-5, atol=2e-5)


def _tg_rows(z, prefix, n):
    """Generator state of a fixture snapshot as rows of lg_buffers.tg_state."""
    rows = np.zeros((n, capi.TG_STRIDE), np.float32)
    key = {"weights": "tg_weights", "t_final": "tg_t_final", "t": "tg_t", "k": "tg_k", "const": "tg_const", "extreme": "tg_extreme",
           "ramp_t_start": "tg_ramp_t_start", "ramp_v_start": "tg_ramp_v_start", "ramp_v_end": "tg_ramp_v_end",
           "sin_mag": "tg_sin_mag", "sin_freq": "tg_sin_freq", "sin_off": "tg_sin_off", "sin_mean": "tg_sin_mean",
           "stationary": "tg_stationary", "v": "tg_v"}
    for name, (off, w) in capi.TG_FIELDS.items():
        rows[:, off:off + w] = np.asarray(z[prefix + key[name]], np.float32).reshape(n, w)
    return rows


def check_stage_against_fixture(setup, state, z, prefix, names):
    """EnvSetup.stage_values(state) against what the reference's update_command_curriculum left in the env."""
    v = setup.stage_values(state if setup.use_curriculum else None)
    assert int(z[prefix + "curriculum_state"]) == state
    np.testing.assert_allclose(np.array([v["reward_scales"][n] for n in names]), z[prefix + "stage_reward_scales"], rtol=1e-12)
    assert v["tracking_sigma"] == float(z[prefix + "stage_tracking_sigma"])
    np.testing.assert_array_equal(np.float32(v["v_min"]), z[prefix + "stage_v_min"])
    np.testing.assert_array_equal(np.float32(v["v_max"]), z[prefix + "stage_v_max"])
    assert (v["t_low"], v["t_high"]) == (float(z[prefix + "stage_t_low"]), float(z[prefix + "stage_t_high"]))
    np.testing.assert_array_equal(np.float32(v["max_rom_dist"]), z[prefix + "stage_max_rom_distance"])
    # reset_traj keeps reading zero_rom_dist_llh, which update_command_curriculum never writes (legged_robot_trajectory.py:73,251,531)
    assert setup.traj["zero_rom_dist_llh"] == float(z[prefix + "stage_zero_rom_dist_llh"])


def set_stage(env, setup, state, in_callback):
    import ctypes as C
    env.call("set_curriculum_stage", C.byref(setup.stage_struct(state)), int(in_callback))


def replay_trajectory_fixture(env, z, meta):
    """Teacher-forced replay of a trajectory-env fixture (the reference's LeggedRobotTrajectory / AnymalTrajectory
    with its torch TrajectoryGenerator, oracle/gen_fixtures_trajectory.py).  Bit-exact: reset / time_out masks, episode
    lengths, last_contacts, reset count, the generator's integer-like state (ROM step counter k, stationary flag, which envs
    were pushed / resampled -- visible through t_final and the push timers); fp32 within TOL.
    anymal_c_flat_trajectory_curriculum: the staged curriculum changes stage inside recorded steps 1 and 3 (the host stage
    machine of the product, CurriculumClock, decides when; lg_set_curriculum_stage carries the constants).
    anymal_c_rough_trajectory: 252 observations with the height scan, custom origins."""
    from legged_gym_dev_amd.envs.base.env_setup import CurriculumClock
    N, A = meta["num_envs"], meta["num_dofs"]
    names = meta["reward_names"]
    ridx = [env.setup.term_row[n] for n in names]
    setup = env.setup
    rough = "const_height_samples" in z.files
    clock = CurriculumClock(setup)
    assert clock.enabled == bool(meta.get("use_curriculum", False))
    if clock.enabled:                                       # the update at construction (legged_robot_trajectory.py:78-79)
        set_stage(env, setup, 0, in_callback=False)
    if "init_curriculum_state" in z.files:
        check_stage_against_fixture(setup, 0, z, "init_", names)

    def install(prefix):
        for key in ("root_states", "dof_state", "last_actions", "last_dof_vel", "last_root_vel", "feet_air_time", "env_origins",
                    "prev_error", "trajectory", "lstm_h", "lstm_c"):
            env.set(key, z[prefix + key])
        env.set("last_contacts", z[prefix + "last_contacts"].astype(np.uint8))
        env.set("episode_length", z[prefix + "episode_length_buf"])
        env.set("push_timer", z[prefix + "time_until_next_push"])
        env.set("tg_state", _tg_rows(z, prefix, N))
        env.set("tg_traj", z[prefix + "tg_traj"])
        es = np.zeros((capi.NUM_TERMS, N), np.float32)
        for k, n in enumerate(names):
            es[env.setup.term_row[n]] = z[prefix + "episode_sums"][:, k]
        env.set("episode_sums", es)
        if rough:
            env.set("terrain_levels", z[prefix + "terrain_levels"])
            env.set("terrain_types", z["const_terrain_types"])
    install("init_")
    counter = int(z["init_common_step_counter"])
    env.set_step_counter(counter)
    env.set_init_done(1)
    env.inject(1)
    dec = z["s0_sub_dof"].shape[0]
    seen = {"reset": 0, "pushed": 0, "resampled": 0, "rom_steps": 0, "stage_changes": 0}
    for t in range(meta["n_steps"]):
        p = f"s{t}_"
        counter += 1
        if clock.tick(counter):                             # legged_robot_trajectory.py:414-417, decided on the host as there
            set_stage(env, setup, clock.state, in_callback=True)
            seen["stage_changes"] += 1
        env.set("episode_length", z[p + "pre_episode_length_buf"])
        env.set_actions(z[p + "actions"])
        for k in range(dec):
            env.call("compute_torques")
            np.testing.assert_allclose(env.get("torques"), z[p + "sub_torques"][k], rtol=1e-4, atol=2e-4, err_msg=f"{p}substep{k} torques")
            env.set("dof_state", z[p + "sub_dof"][k])
        env.set("root_states", z[p + "new_root"])
        env.set("contact_forces", z[p + "contact_forces"])
        env.set("inject_uniforms", np.nan_to_num(z[p + "uniforms"], nan=0.5))
        if p + "inj_level" in z.files:
            env.set("inject_levels", np.maximum(z[p + "inj_level"], 0))
        k_before, tf_before = env.get("tg_state")[:, capi.TG_FIELDS["k"][0]].copy(), env.get("tg_state")[:, capi.TG_FIELDS["t_final"][0]].copy()
        env.call("post_physics_step")
        env.sync()
        np.testing.assert_array_equal(env.get("reset").astype(bool), z[p + "reset"].astype(bool), err_msg=p + "reset")
        np.testing.assert_array_equal(env.get("time_out").astype(bool), z[p + "time_out"], err_msg=p + "time_out")
        np.testing.assert_array_equal(env.get("episode_length"), z[p + "post_episode_length_buf"], err_msg=p + "ep_len")
        np.testing.assert_array_equal(env.get("last_contacts").astype(bool), z[p + "post_last_contacts"], err_msg=p + "last_contacts")
        assert int(env.get("n_reset")[0]) == int(z[p + "n_reset"]), p + "n_reset"
        np.testing.assert_array_equal(env.get("extras_time_outs").astype(bool), z[p + "extras_time_outs"], err_msg=p + "extras time_outs")
        want = _tg_rows(z, p + "post_", N)
        got = env.get("tg_state")
        for name in ("k", "stationary"):                     # discrete generator state: exact
            o = capi.TG_FIELDS[name][0]
            np.testing.assert_array_equal(got[:, o], want[:, o], err_msg=p + "generator " + name)
        np.testing.assert_allclose(got, want, err_msg=p + "generator state", **TOL)
        for key, ref in (("obs", z[p + "obs"]), ("rew", z[p + "rew"]), ("root_states", z[p + "post_root_states"]),
                         ("dof_state", z[p + "post_dof_state"]), ("last_actions", z[p + "post_last_actions"]),
                         ("last_dof_vel", z[p + "post_last_dof_vel"]), ("last_root_vel", z[p + "post_last_root_vel"]),
                         ("feet_air_time", z[p + "post_feet_air_time"]), ("tg_traj", z[p + "post_tg_traj"]),
                         ("trajectory", z[p + "post_trajectory"]), ("prev_error", z[p + "post_prev_error"]),
                         ("push_timer", z[p + "post_time_until_next_push"])):
            np.testing.assert_allclose(env.get(key), np.asarray(ref).reshape(env.get(key).shape), err_msg=p + key, **TOL)
        es = env.get("episode_sums")[ridx].T
        np.testing.assert_allclose(es, z[p + "post_episode_sums"], err_msg=p + "episode_sums", **TOL)
        if int(z[p + "n_reset"]) > 0:
            np.testing.assert_allclose(env.get("extras_episode")[ridx], z[p + "extras_episode"], rtol=1e-4, atol=1e-5,
                                       err_msg=p + "extras episode means")
        if rough:
            np.testing.assert_allclose(env.get("measured_heights"), z[p + "measured_heights"], err_msg=p + "heights", **TOL)
            np.testing.assert_array_equal(env.get("terrain_levels"), z[p + "post_terrain_levels"], err_msg=p + "levels")
            np.testing.assert_allclose(env.get("env_origins"), z[p + "post_env_origins"], err_msg=p + "env_origins", **TOL)
        if p + "post_curriculum_state" in z.files:
            check_stage_against_fixture(setup, clock.state, z, p + "post_", names)
        np.testing.assert_allclose(env.get("lstm_h"), z[p + "post_lstm_h"], rtol=1e-4, atol=1e-5, err_msg=p + "lstm_h")
        np.testing.assert_allclose(env.get("lstm_c"), z[p + "post_lstm_c"], rtol=1e-4, atol=1e-5, err_msg=p + "lstm_c")
        rst = z[p + "reset"].astype(bool)
        seen["reset"] += int(rst.sum())
        seen["pushed"] += int(z[p + "n_pushed"])
        seen["resampled"] += int(((want[:, capi.TG_FIELDS["t_final"][0]] != tf_before) & ~rst).sum())
        seen["rom_steps"] += int(((want[:, capi.TG_FIELDS["k"][0]] != k_before) & ~rst).sum())
    # the fixture exercised every event of the variant
    assert seen["reset"] > 20 and seen["pushed"] > 20 and seen["resampled"] >= 4 and seen["rom_steps"] > 40, seen
    assert seen["stage_changes"] == (2 if clock.enabled else 0), seen


def replay_fixture(env, z, meta, torque_tol=None, report=None):
    """Teacher-forced replay of every recorded step; asserts parity with the reference outputs.

    Bit-exact: reset / time_out masks, episode lengths, terrain levels, last_contacts, reset count.
    fp32 within TOL (rtol=atol=2e-5; torques from the actuator net 1e-4 abs): everything else."""
    from legged_gym_dev_amd.envs.base.env_setup import CurriculumClock
    N, A = meta["num_envs"], meta["num_dofs"]
    names = meta["reward_names"]
    ridx = [env.setup.term_row[n] for n in names]
    ttol = torque_tol or dict(rtol=1e-4, atol=2e-4)
    load_state(env, z, "init_", meta)
    counter = int(z["init_common_step_counter"])
    env.set_step_counter(counter)
    env.set_init_done(1)
    env.inject(1)
    setup = env.setup
    clock = CurriculumClock(setup)                         # the product's stage machine (legged_robot.py:360-363)
    assert clock.enabled == bool(meta.get("use_curriculum", False))

    def check_stage(prefix):
        if prefix + "curriculum_state" not in z.files:
            return
        v = setup.stage_values(clock.state if clock.enabled else None)
        assert int(z[prefix + "curriculum_state"]) == clock.state
        assert v["push_time"] == float(z[prefix + "stage_push_time"]) and v["max_push_vel"] == float(z[prefix + "stage_max_push_vel"])
        np.testing.assert_array_equal(np.array([v["command_ranges"][k] for k in ("lin_vel_x", "lin_vel_y", "ang_vel_yaw", "heading")]),
                                      z[prefix + "stage_command_ranges"])
    if clock.enabled:                                      # the update inside _parse_cfg (legged_robot.py:828-829)
        set_stage(env, setup, 0, in_callback=False)
    check_stage("init_")
    dec = z["s0_sub_dof"].shape[0]
    changes = 0
    for t in range(meta["n_steps"]):
        p = f"s{t}_"
        counter += 1
        if clock.tick(counter):
            set_stage(env, setup, clock.state, in_callback=True)
            changes += 1
        env.set("episode_length", z[p + "pre_episode_length_buf"])   # generator edits it before step 2
        env.set_actions(z[p + "actions"])
        for k in range(dec):
            env.call("compute_torques")
            np.testing.assert_allclose(env.get("torques"), z[p + "sub_torques"][k], err_msg=f"{p}substep{k} torques", **ttol)
            env.set("dof_state", z[p + "sub_dof"][k])           # teacher forcing (physics not in fixture)
        env.set("root_states", z[p + "new_root"])
        env.set("contact_forces", z[p + "contact_forces"])
        U = np.nan_to_num(z[p + "uniforms"], nan=0.5)
        env.set("inject_uniforms", U)
        env.set("inject_levels", np.maximum(z[p + "inj_level"], 0))
        env.call("post_physics_step")
        env.sync()
        # ---- integer / mask outputs: bit exact
        np.testing.assert_array_equal(env.get("reset").astype(bool), z[p + "reset"].astype(bool), err_msg=p + "reset")
        np.testing.assert_array_equal(env.get("time_out").astype(bool), z[p + "time_out"], err_msg=p + "time_out")
        np.testing.assert_array_equal(env.get("episode_length"), z[p + "post_episode_length_buf"], err_msg=p + "ep_len")
        np.testing.assert_array_equal(env.get("last_contacts").astype(bool), z[p + "post_last_contacts"], err_msg=p + "last_contacts")
        assert int(env.get("n_reset")[0]) == int(z[p + "n_reset"]), p + "n_reset"
        if p + "post_terrain_levels" in z.files:
            np.testing.assert_array_equal(env.get("terrain_levels"), z[p + "post_terrain_levels"], err_msg=p + "levels")
        np.testing.assert_array_equal(env.get("extras_time_outs").astype(bool), z[p + "extras_time_outs"],
                                      err_msg=p + "extras time_outs (stale-mask quirk)")
        # ---- fp32 outputs
        for key, ref in (("obs", z[p + "obs"]), ("rew", z[p + "rew"]), ("root_states", z[p + "post_root_states"]),
                         ("dof_state", z[p + "post_dof_state"]), ("commands", z[p + "post_commands"]),
                         ("last_actions", z[p + "post_last_actions"]), ("last_dof_vel", z[p + "post_last_dof_vel"]),
                         ("last_root_vel", z[p + "post_last_root_vel"]), ("feet_air_time", z[p + "post_feet_air_time"]),
                         ("env_origins", z[p + "post_env_origins"])):
            np.testing.assert_allclose(env.get(key), ref, err_msg=p + key, **TOL)
        if z[p + "measured_heights"].shape[1]:
            np.testing.assert_allclose(env.get("measured_heights"), z[p + "measured_heights"], err_msg=p + "heights", **TOL)
        es = env.get("episode_sums")[ridx].T if ridx else np.zeros((N, 0), np.float32)
        np.testing.assert_allclose(es, z[p + "post_episode_sums"], err_msg=p + "episode_sums", **TOL)
        if int(z[p + "n_reset"]) > 0 and ridx:
            np.testing.assert_allclose(env.get("extras_episode")[ridx], z[p + "extras_episode"], rtol=1e-4, atol=1e-5,
                                       err_msg=p + "extras episode means")
            if meta["curriculum"]:
                np.testing.assert_allclose(env.get("extras_terrain_level")[0], z[p + "extras_terrain_level"], rtol=1e-5)
        if meta["use_lstm"]:
            np.testing.assert_allclose(env.get("lstm_h"), z[p + "post_lstm_h"], rtol=1e-4, atol=1e-5, err_msg=p + "lstm_h")
            np.testing.assert_allclose(env.get("lstm_c"), z[p + "post_lstm_c"], rtol=1e-4, atol=1e-5, err_msg=p + "lstm_c")
        check_stage(p + "post_")
        if report is not None:
            report.append((t, float(np.abs(env.get("obs") - z[p + "obs"]).max())))
    assert changes == (2 if clock.enabled else 0)


class HipHandle:
    """Adapter giving the HIP context (legged_gym_dev_amd.lib.HipEnvCore) the get/set/call
    interface of oracle_lib.OracleEnv; every call goes through the C-ABI of liblegged_hip.so."""

    def __init__(self, setup, height_samples=None, device="cuda:0"):
        import torch
        from legged_gym_dev_amd.lib import HipEnvCore
        self.torch = torch
        self.core = HipEnvCore(setup, height_samples, device)
        self.setup = setup
        self._act = None

    def get(self, name):
        self.torch.cuda.synchronize()
        return self.core.t[name].cpu().numpy()

    def set(self, name, value):
        t = self.core.t[name]
        v = self.torch.as_tensor(np.ascontiguousarray(np.asarray(value)).reshape(tuple(t.shape)))
        t.copy_(v.to(t.dtype))

    def call(self, fn, *args):
        self.core.call(fn, *args)

    def set_actions(self, actions):
        import ctypes as C
        self._act = self.torch.as_tensor(np.ascontiguousarray(actions, np.float32)).to(self.core.device)
        self.core.call("set_actions", C.c_void_p(self._act.data_ptr()))

    def step(self, actions):
        self._act = self.torch.as_tensor(np.ascontiguousarray(actions, np.float32)).to(self.core.device)
        self.core.step(self._act)

    def set_step_counter(self, v):
        self.core.lib.lg_set_step_counter(self.core.ctx, int(v))

    def set_init_done(self, v):
        self.core.lib.lg_set_init_done(self.core.ctx, int(v))

    def inject(self, enable):
        self.core.lib.lg_inject_uniforms(self.core.ctx, int(enable))

    def sync(self):
        self.torch.cuda.synchronize()

    def close(self):
        self.core.close()
