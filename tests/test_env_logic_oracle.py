"""CPU oracle (oracle/lgo_env.cpp) vs golden vectors produced by the reference's own Python
(legged_robot.py post_physics_step / _compute_torques, anymal.py, cassie.py) -- this is what PINS the
oracle.  Also pins the host-side setup logic (EnvSetup) against the constants the reference derived."""
import numpy as np
import pytest

from tests import harness


@pytest.mark.parametrize("name", harness.FIXTURES)
def test_setup_matches_reference_constants(name):
    z, meta = harness.load_fixture(name)
    setup, _ = harness.make_setup(name, z, meta)
    harness.check_setup_against_fixture(setup, z, meta)


@pytest.mark.parametrize("name", harness.FIXTURES)
def test_oracle_replays_reference_steps(name, oracle_built):
    z, meta = harness.load_fixture(name)
    setup, _ = harness.make_setup(name, z, meta)
    hs = z["const_height_samples"] if "const_height_samples" in z.files else None
    env = oracle_built.OracleEnv(setup, hs)
    try:
        harness.replay_fixture(env, z, meta)
    finally:
        env.close()


def test_actuator_lstm_golden(oracle_built):
    """Actuator network rows vs torch.nn.LSTM goldens (tests/golden/actuator_lstm.npz)."""
    import os
    g = np.load(os.path.join(harness.GOLDEN, "actuator_lstm.npz"))
    z, meta = harness.load_fixture("anymal_c_flat")
    setup, cfg = harness.make_setup("anymal_c_flat", z, meta)
    env = oracle_built.OracleEnv(setup)
    try:
        N, A = 64, 12
        env.set("lstm_h", g["h0"])
        env.set("lstm_c", g["c0"])
        dq = setup.default_dof_pos
        for k in range(4):
            x = g[f"x{k}"].reshape(N, A, 2)
            # choose q, qd, actions so that the network input equals the golden x
            dof = np.zeros((N, A, 2), np.float32)
            dof[..., 0] = dq[None, :] - x[..., 0]
            dof[..., 1] = x[..., 1]
            env.set("dof_state", dof)
            env.set_actions(np.zeros((N, A), np.float32))
            env.call("compute_torques")
            np.testing.assert_allclose(env.get("torques").reshape(-1), g[f"y{k}"], rtol=1e-4, atol=2e-4)
            if k == 1:
                h, c = env.get("lstm_h"), env.get("lstm_c")
                h[:, :96] = 0
                c[:, :96] = 0
                env.set("lstm_h", h)
                env.set("lstm_c", c)
            np.testing.assert_allclose(env.get("lstm_h"), g[f"h{k + 1}"], rtol=1e-4, atol=1e-5)
            np.testing.assert_allclose(env.get("lstm_c"), g[f"c{k + 1}"], rtol=1e-4, atol=1e-5)
    finally:
        env.close()


TRAJ = "anymal_c_flat_trajectory"
TRAJ_FIXTURES = [TRAJ, "anymal_c_flat_trajectory_curriculum", "anymal_c_rough_trajectory"]


@pytest.mark.parametrize("name", TRAJ_FIXTURES)
def test_trajectory_setup_matches_reference_constants(name):
    """Host setup of the trajectory-tracking variant against what the reference's LeggedRobotTrajectory
    derived: index sets, gains, noise vector (trajectory block unscaled), reward order incl. the two extra terms, ROM bounds,
    generator parameters, slot layout."""
    z, meta = harness.load_fixture(name)
    setup, _ = harness.make_setup(name, z, meta)
    harness.check_setup_against_fixture(setup, z, meta)
    assert setup.xterm_names == ["tracking_rom", "differential_error"]


@pytest.mark.parametrize("name", TRAJ_FIXTURES)
def test_oracle_replays_reference_trajectory_steps(name, oracle_built):
    """The oracle's trajectory env (lgo_traj.cpp + the traj branches of lgo_env.cpp) against the recorded steps of the reference's
    own LeggedRobotTrajectory / AnymalTrajectory / TrajectoryGenerator: this is what pins it.  The curriculum fixture changes
    stage inside steps 1 and 3 (reward scales, tracking sigma, ROM input bounds, hold-time sampler, start-offset range: the
    callback of the change step still resamples with the old stage, its resets with the new one); the rough-terrain fixture
    carries the height scan and the 252-wide observation."""
    z, meta = harness.load_fixture(name)
    setup, _ = harness.make_setup(name, z, meta)
    hs = z["const_height_samples"] if "const_height_samples" in z.files else None
    env = oracle_built.OracleEnv(setup, hs)
    try:
        harness.replay_trajectory_fixture(env, z, meta)
    finally:
        env.close()
