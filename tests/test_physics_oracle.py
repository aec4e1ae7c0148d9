"""Physics oracle (oracle/lgo_physics.cpp) validated by invariants.

PhysX is closed and absent, so parity of the physics step with the reference is UNPINNED
(SURVEY.md §8(c)); what pins our own specification is physics itself:
  * free dynamics == float64 mass-matrix dynamics  M(q) nu_dot + h(q, nu) = tau  built from link
    Jacobians in the world frame (an independent formulation: no articulated-body recursion),
  * momentum conservation without gravity,
  * static stance: sum of vertical contact forces == total weight, base stays put.
"""
import numpy as np
import pytest

from tests import harness
from legged_gym_dev_amd.envs.base.env_setup import EnvSetup, sim_dt_float
from legged_gym_dev_amd.model.robot_model import compile_model, resolve_model


def _cfg(robot, n=8, gravity=True):
    from legged_gym_dev_amd.envs.anymal_c.flat.anymal_c_flat_config import AnymalCFlatCfg
    from legged_gym_dev_amd.envs.cassie.cassie_config import CassieRoughCfg
    if robot == "anymal_c":
        cfg = AnymalCFlatCfg()
        cfg.control.use_actuator_network = False
    else:
        if robot == "a1":
            from legged_gym_dev_amd.envs.a1.a1_config import A1RoughCfg
            cfg = A1RoughCfg()
        elif robot == "anymal_b":
            from legged_gym_dev_amd.envs.anymal_b.anymal_b_config import AnymalBRoughCfg
            cfg = AnymalBRoughCfg()
            cfg.control.use_actuator_network = False
        else:
            cfg = CassieRoughCfg()
        cfg.terrain.mesh_type = "plane"
        cfg.terrain.measure_heights = False
        cfg.terrain.curriculum = False
        cfg.env.num_observations = 48
    cfg.env.num_envs = n
    cfg.control.control_type = "T"
    cfg.control.action_scale = 1.0
    if not gravity:
        cfg.sim.gravity = [0.0, 0.0, 0.0]
    return cfg


def _make(robot, oracle_built, n=8, gravity=True, cfg=None):
    cfg = cfg or _cfg(robot, n, gravity)
    cm = compile_model(resolve_model("", robot))
    setup = EnvSetup(cfg, cm, sim_dt_float(cfg.sim.dt))
    return oracle_built.OracleEnv(setup), cm, cfg


def _quat_mat(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _rot(a, th):
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def _kin(cm, x, Rb, q):
    """World pose of every dynamics link + joint axes/origins."""
    A, J = cm["num_dofs"], cm["joints_per_leg"]
    R, p, ax, po = [Rb], [x], [], []
    for d in range(A):
        pl = 0 if d % J == 0 else d
        Rj = R[pl] @ cm["R_pj"][d].astype(np.float64)
        pj = p[pl] + R[pl] @ cm["p_pj"][d].astype(np.float64)
        a = cm["axis"][d].astype(np.float64)
        ax.append(Rj @ a)
        po.append(pj)
        R.append(Rj @ _rot(a, q[d]))
        p.append(pj)
    return R, p, ax, po


def _jacobians(cm, x, Rb, q):
    A, J = cm["num_dofs"], cm["joints_per_leg"]
    R, p, ax, po = _kin(cm, x, Rb, q)
    out = []
    for l in range(A + 1):
        c = p[l] + R[l] @ cm["com"][l].astype(np.float64)
        Jv, Jw = np.zeros((3, 6 + A)), np.zeros((3, 6 + A))
        Jv[:, :3] = np.eye(3)
        r = c - x
        Jv[:, 3:6] = -np.array([[0, -r[2], r[1]], [r[2], 0, -r[0]], [-r[1], r[0], 0]])
        Jw[:, 3:6] = np.eye(3)
        if l > 0:
            leg = (l - 1) // J
            for d in range(leg * J, l):
                Jv[:, 6 + d] = np.cross(ax[d], c - po[d])
                Jw[:, 6 + d] = ax[d]
        Iw = R[l] @ cm["inertia"][l].astype(np.float64) @ R[l].T
        out.append((float(cm["mass"][l]), Jv, Jw, Iw))
    return out


def _advance(x, Rb, q, nu, eps):
    w = nu[3:6]
    th = np.linalg.norm(w) * eps
    Rn = (_rot(w / np.linalg.norm(w), th) if th != 0 else np.eye(3)) @ Rb
    return x + eps * nu[:3], Rn, q + eps * nu[6:]


def mass_matrix_dynamics(cm, x, quat, q, v_w, w_w, qd, tau, g, armature=0.0):
    """nu_dot for nu = [v_w, w_w, qd] from M nu_dot + h = tau_gen (float64).  armature: the asset option (legged_robot.py:703), a
    reflected rotor inertia added to the joint-space inertia of every joint."""
    A = cm["num_dofs"]
    Rb = _quat_mat(quat)
    nu = np.concatenate([v_w, w_w, qd])
    J0 = _jacobians(cm, x, Rb, q)
    eps = 1e-6
    Jp = _jacobians(cm, *_advance(x, Rb, q, nu, eps))
    Jm = _jacobians(cm, *_advance(x, Rb, q, nu, -eps))
    M = np.zeros((6 + A, 6 + A))
    h = np.zeros(6 + A)
    for (m, Jv, Jw, Iw), (_, Jvp, Jwp, _), (_, Jvm, Jwm, _) in zip(J0, Jp, Jm):
        M += m * Jv.T @ Jv + Jw.T @ Iw @ Jw
        dJv = (Jvp - Jvm) / (2 * eps) @ nu
        dJw = (Jwp - Jwm) / (2 * eps) @ nu
        w = Jw @ nu
        h += Jv.T @ (m * dJv - m * g) + Jw.T @ (Iw @ dJw + np.cross(w, Iw @ w))
    M[6:, 6:] += armature * np.eye(A)
    tg = np.concatenate([np.zeros(6), tau])
    return np.linalg.solve(M, tg - h), M


def _random_state(env, cm, rng, n, z=5.0):
    A = cm["num_dofs"]
    root = np.zeros((n, 13), np.float32)
    root[:, 2] = z
    qn = rng.normal(size=(n, 4))
    root[:, 3:7] = qn / np.linalg.norm(qn, axis=1, keepdims=True)
    root[:, 7:13] = rng.uniform(-1.5, 1.5, (n, 6))
    dof = np.zeros((n, A, 2), np.float32)
    dof[..., 0] = env.setup.default_dof_pos + rng.uniform(-0.5, 0.5, (n, A))
    lo, hi = cm["q_lower"], cm["q_upper"]                      # free-flight tests: stay clear of the joint-limit constraints
    lim = hi > lo
    dof[..., 0] = np.where(lim, np.clip(dof[..., 0], lo + 0.05, hi - 0.05), dof[..., 0])
    dof[..., 1] = rng.uniform(-4, 4, (n, A))
    env.set("root_states", root)
    env.set("dof_state", dof)
    return root, dof


@pytest.mark.parametrize("robot", ["anymal_c", "cassie", "a1", "anymal_b"])
def test_free_dynamics_match_mass_matrix_formulation(robot, oracle_built):
    env, cm, cfg = _make(robot, oracle_built)
    try:
        rng = np.random.default_rng(0)
        n, A = 8, cm["num_dofs"]
        root, dof = _random_state(env, cm, rng, n)
        tmax = 4.0 if robot == "a1" else 30.0            # A1's 0.15 kg calf would run into its 21 rad/s velocity limit within one dt
        tau = rng.uniform(-tmax, tmax, (n, A)).astype(np.float32)
        env.set("torques", tau)
        env.call("simulate")
        r1, d1 = env.get("root_states"), env.get("dof_state")
        dt = env.setup.sim_dt
        g = np.array(cfg.sim.gravity, np.float64)
        for i in range(n):
            nud, _ = mass_matrix_dynamics(cm, root[i, :3].astype(np.float64), root[i, 3:7].astype(np.float64),
                                          dof[i, :, 0].astype(np.float64), root[i, 7:10].astype(np.float64),
                                          root[i, 10:13].astype(np.float64), dof[i, :, 1].astype(np.float64),
                                          tau[i].astype(np.float64), g)
            acc_lin = (r1[i, 7:10] - root[i, 7:10]) / dt
            acc_ang = (r1[i, 10:13] - root[i, 10:13]) / dt
            qdd = (d1[i, :, 1] - dof[i, :, 1]) / dt
            scale = max(1.0, np.abs(nud).max())
            np.testing.assert_allclose(acc_lin, nud[:3], atol=2e-3 * scale)
            np.testing.assert_allclose(acc_ang, nud[3:6], atol=2e-3 * scale)
            np.testing.assert_allclose(qdd, nud[6:], atol=2e-3 * scale)
            assert np.all(env.get("contact_forces")[i] == 0)
    finally:
        env.close()


@pytest.mark.parametrize("robot", ["anymal_c", "cassie"])
def test_armature_adds_to_the_joint_space_inertia(robot, oracle_built):
    """cfg.asset.armature (legged_robot.py:703, default 0) against the float64 mass-matrix formulation with armature on the joint
    diagonal: one substep in free flight, and the joint accelerations really are smaller than without it."""
    arm = 0.05
    cfg = _cfg(robot)
    cfg.asset.armature = arm
    env, cm, cfg = _make(robot, oracle_built, cfg=cfg)
    ref, _, _ = _make(robot, oracle_built)
    try:
        rng = np.random.default_rng(5)
        n, A = 8, cm["num_dofs"]
        root, dof = _random_state(env, cm, rng, n)
        ref.set("root_states", root)
        ref.set("dof_state", dof)
        tau = rng.uniform(-30, 30, (n, A)).astype(np.float32)
        for e in (env, ref):
            e.set("torques", tau)
            e.call("simulate")
        d1, d0 = env.get("dof_state"), ref.get("dof_state")
        r1 = env.get("root_states")
        dt = env.setup.sim_dt
        g = np.array(cfg.sim.gravity, np.float64)
        for i in range(n):
            nud, _ = mass_matrix_dynamics(cm, root[i, :3].astype(np.float64), root[i, 3:7].astype(np.float64), dof[i, :, 0].astype(np.float64),
                                          root[i, 7:10].astype(np.float64), root[i, 10:13].astype(np.float64), dof[i, :, 1].astype(np.float64),
                                          tau[i].astype(np.float64), g, armature=arm)
            scale = max(1.0, np.abs(nud).max())
            np.testing.assert_allclose((d1[i, :, 1] - dof[i, :, 1]) / dt, nud[6:], atol=2e-3 * scale)
            np.testing.assert_allclose((r1[i, 10:13] - root[i, 10:13]) / dt, nud[3:6], atol=2e-3 * scale)
        acc1, acc0 = np.abs(d1[..., 1] - dof[..., 1]), np.abs(d0[..., 1] - dof[..., 1])
        assert acc1.mean() < 0.8 * acc0.mean(), (acc1.mean(), acc0.mean())
    finally:
        env.close()
        ref.close()


def _momentum(cm, root, dof):
    Rb = _quat_mat(root[3:7].astype(np.float64))
    nu = np.concatenate([root[7:10], root[10:13], dof[:, 1]]).astype(np.float64)
    P, L = np.zeros(3), np.zeros(3)
    R, p, _, _ = _kin(cm, root[:3].astype(np.float64), Rb, dof[:, 0].astype(np.float64))
    for l, (m, Jv, Jw, Iw) in enumerate(_jacobians(cm, root[:3].astype(np.float64), Rb, dof[:, 0].astype(np.float64))):
        c = p[l] + R[l] @ cm["com"][l].astype(np.float64)
        v, w = Jv @ nu, Jw @ nu
        P += m * v
        L += np.cross(c, m * v) + Iw @ w
    return P, L


def test_momentum_conserved_without_gravity(oracle_built):
    env, cm, cfg = _make("anymal_c", oracle_built, n=4, gravity=False)
    try:
        rng = np.random.default_rng(1)
        root, dof = _random_state(env, cm, rng, 4)
        dof[..., 1] *= 0.25                     # stay clear of the 20 rad/s joint-velocity clamp
        env.set("dof_state", dof)
        env.set("torques", rng.uniform(-1, 1, (4, 12)).astype(np.float32))     # internal torques only
        P0 = [_momentum(cm, root[i], dof[i]) for i in range(4)]
        for _ in range(40):
            env.call("simulate")
        r, d = env.get("root_states"), env.get("dof_state")
        for i in range(4):
            P1, L1 = _momentum(cm, r[i], d[i])
            np.testing.assert_allclose(P1, P0[i][0], atol=0.02 * 52)       # 2 cm/s of the 52 kg robot
            np.testing.assert_allclose(L1, P0[i][1], atol=0.05 * max(1.0, np.abs(P0[i][1]).max()))
    finally:
        env.close()


def test_momentum_conserved_with_joints_at_their_velocity_limit(oracle_built):
    """The regression test of round 4's solver fix (profiles/r04_diag_faults.txt).  Free flight, no gravity, the knee of every leg driven
    by a constant 60 N m into its 20 rad/s velocity limit and held there for 40 substeps: internal torques cannot change the robot's
    momentum.  With the limit as a row of the sweeps (a joint-space impulse, equal and opposite on child and parent; one active row per
    chain is solved exactly) linear and angular momentum stay put and the base barely turns.  The clamp at integration it replaced took
    the child's excess rate away and left the base its reaction: 0.3 N m s per joint and substep, i.e. a drift of tens of kg m^2 / s and
    a base spin of several rad/s over the same 40 substeps.
    With SEVERAL joints of one chain saturated at once the rows are relaxed Jacobi rows (1 / active rows, 4 iterations) and what they leave
    goes to the integration clamp: the drift is smaller than before but not zero -- there only boundedness is asserted."""
    for knees_only in (True, False):
        env, cm, cfg = _make("anymal_c", oracle_built, n=4, gravity=False)
        try:
            rng = np.random.default_rng(2)
            n, A = 4, 12
            root, dof = _random_state(env, cm, rng, n)
            root[:, 7:13] = 0.0
            dof[..., 1] = 0.0
            env.set("root_states", root)
            env.set("dof_state", dof)
            tau = (60.0 * rng.choice([-1.0, 1.0], (n, A))).astype(np.float32)
            if knees_only:
                tau[:, [0, 1, 3, 4, 6, 7, 9, 10]] = 0.0
            env.set("torques", tau)
            P0 = [_momentum(cm, root[i], dof[i]) for i in range(n)]
            at_limit = 0
            for k in range(40):
                env.call("simulate")
                at_limit = max(at_limit, int((np.abs(env.get("dof_state")[..., 1]) > 19.9).sum()))
            r, d = env.get("root_states"), env.get("dof_state")
            assert np.isfinite(r).all() and np.isfinite(d).all()
            if knees_only:
                assert at_limit == 16, at_limit                                   # all four knees of all four robots sit at the limit
                assert np.abs(r[:, 10:13]).max() < 1.0, np.abs(r[:, 10:13]).max()  # the base barely turns
                for i in range(n):
                    P1, L1 = _momentum(cm, r[i], d[i])
                    np.testing.assert_allclose(P1, P0[i][0], atol=0.5)                 # kg m / s (52 kg robot: < 1 cm / s)
                    np.testing.assert_allclose(L1, P0[i][1], atol=2.5)                 # kg m^2 / s about the world origin, 5 m below the robot
            else:
                assert at_limit >= 24 and np.abs(r[:, 10:13]).max() < 40.0, (at_limit, np.abs(r[:, 10:13]).max())
        finally:
            env.close()


@pytest.mark.parametrize("robot,height", [("anymal_c", 0.56), ("cassie", 0.95), ("a1", 0.36), ("anymal_b", 0.56)])
def test_static_stance_supports_weight(robot, height, oracle_built):
    cfg = _cfg(robot, n=4)
    cfg.control.control_type = "P"
    cfg.control.action_scale = 0.5
    lo, hi = {"a1": (0.2, 0.42), "anymal_b": (0.35, 0.62)}.get(robot, (0.35, 0.62))
    env, cm, cfg = _make(robot, oracle_built, cfg=cfg)
    try:
        n, A = 4, cm["num_dofs"]
        root = np.zeros((n, 13), np.float32)
        root[:, 2] = height
        root[:, 6] = 1.0
        dof = np.zeros((n, A, 2), np.float32)
        dof[..., 0] = env.setup.default_dof_pos
        env.set("root_states", root)
        env.set("dof_state", dof)
        env.set("friction", np.ones(n, np.float32))
        env.set_actions(np.zeros((n, A), np.float32))
        fz = []
        for k in range(1200 if robot == "a1" else 300):     # A1's soft gains (kp 20) leave a slow, damped pitch rocking
            env.call("compute_torques")
            env.call("simulate")
            fz.append(env.get("contact_forces")[:, :, 2].sum(1))
        r = env.get("root_states")
        assert np.all(np.isfinite(r))
        weight = float(cm["mass"].sum()) * 9.81
        if robot != "cassie":        # statically stable quadruped stance
            np.testing.assert_allclose(np.mean(fz[-50:], 0), weight, rtol=0.03)
            assert np.all(np.abs(r[:, 2] - r[0, 2]) < 1e-3)
            assert np.all(r[:, 2] > lo) and np.all(r[:, 2] < hi)
            assert np.all(np.abs(r[:, 7:13]) < 0.05)
            feet = env.get("contact_forces")[:, env.setup.feet_indices, 2]
            assert np.all(feet > 0.1 * weight / 4)
        else:                        # a PD-held biped is not statically stable: only sanity
            assert np.max(fz) > 0.5 * weight
    finally:
        env.close()


@pytest.mark.parametrize("robot", ["a1", "cassie"])
def test_joint_limits_hold_against_torque(robot, oracle_built):
    """URDF joint limits are constraints of the solver.  Free flight, large torques: (env 0/1) every joint driven
    into its upper / lower stop at once -- none ends a step beyond its stop (fp32 slack 2e-3 rad); (env 2) joints that
    start 0.05 rad beyond the stop come back inside; (env 3) the last joint of every chain alone against its stop stays
    pinned there while the torque lasts."""
    env, cm, cfg = _make(robot, oracle_built, n=4)
    try:
        n, A, J = 4, cm["num_dofs"], cm["joints_per_leg"]
        lo, hi = cm["q_lower"].astype(np.float64), cm["q_upper"].astype(np.float64)
        assert np.all(hi > lo)
        root = np.zeros((n, 13), np.float32)
        root[:, 2] = 5.0
        root[:, 6] = 1.0
        dof = np.zeros((n, A, 2), np.float32)
        mid = 0.5 * (lo + hi)
        dof[..., 0] = mid
        dof[0, :, 0] = hi - 0.01
        dof[1, :, 0] = lo + 0.01
        dof[2, :, 0] = hi + 0.05
        last = np.arange(J - 1, A, J)
        dof[3, last, 0] = hi[last] - 0.01
        env.set("root_states", root)
        env.set("dof_state", dof)
        tmax = 3.0 if robot == "a1" else 20.0
        tau = np.zeros((n, A), np.float32)
        tau[0], tau[1] = tmax, -tmax
        tau[3, last] = tmax
        env.set("torques", tau)
        worst_hi, worst_lo = -1.0, -1.0
        for _ in range(40):
            env.call("simulate")
            d = env.get("dof_state").astype(np.float64)
            worst_hi = max(worst_hi, float(np.max(d[0, :, 0] - hi)), float(np.max(d[3, last, 0] - hi[last])))
            worst_lo = max(worst_lo, float(np.max(lo - d[1, :, 0])))
        assert np.all(np.isfinite(d))
        assert worst_hi <= 2e-3 and worst_lo <= 2e-3, (worst_hi, worst_lo)
        assert np.all(d[2, :, 0] <= hi + 0.02), (d[2, :, 0] - hi)         # ERP 0.2 per step for 40 steps: back inside
        assert np.all(np.abs(d[3, last, 0] - hi[last]) < 0.03) and np.all(np.abs(d[3, last, 1]) < 0.5)
    finally:
        env.close()


def test_shape_material_parameters_act_on_the_contacts(oracle_built):
    """The randomised rigid-shape properties of legged_robot.py:284-299 as the sphere-set contact model carries them (one material
    per robot, lg_buffers.material): thickness = rest offset (the stance settles that much higher), restitution = the contact leaves
    with e x its approach speed when that exceeds sim.physx.bounce_threshold_velocity (a dropped robot rebounds, e = 0 does not),
    compliance = constraint-force mixing on the normal row (a softer contact sinks deeper under the same weight).  Env 0 carries
    the asset defaults (restitution 0, compliance 0, thickness = cfg.asset.thickness) and must behave exactly as with the
    randomisation switched off."""
    cfg = _cfg("anymal_c", n=4)
    cfg.control.control_type, cfg.control.action_scale = "P", 0.5
    rsp = cfg.domain_rand.rigid_shape_properties
    rsp.randomize_restitution = rsp.randomize_compliance = rsp.randomize_thickness = True
    env, cm, cfg = _make("anymal_c", oracle_built, cfg=cfg)
    ref, _, _ = _make("anymal_c", oracle_built, cfg=(lambda c: (setattr(c.control, "control_type", "P"), setattr(c.control, "action_scale", 0.5), c)[-1])(_cfg("anymal_c", n=4)))
    try:
        assert env.setup.to_structs()[0].material_rand == 1 and ref.setup.to_structs()[0].material_rand == 0
        n, A = 4, cm["num_dofs"]
        mat = np.zeros((n, 4), np.float32)
        mat[:, 2] = cfg.asset.thickness                    # the asset option every shape starts from (legged_robot.py:704)
        mat[1, 2] += 0.02                                  # thickness
        mat[2, 0] = 1.6                                    # restitution (combined with the plane's 0 by averaging: e = 0.8)
        mat[3, 1] = 2.0e-6                                 # compliance, m/N
        env.set("material", mat)

        def settle(e, z0, vz0, steps):
            root = np.zeros((n, 13), np.float32)
            root[:, 2], root[:, 6], root[:, 9] = z0, 1.0, vz0
            dof = np.zeros((n, A, 2), np.float32)
            dof[..., 0] = e.setup.default_dof_pos
            e.set("root_states", root)
            e.set("dof_state", dof)
            e.set("friction", np.ones(n, np.float32))
            e.set_actions(np.zeros((n, A), np.float32))
            fz = []
            for _ in range(steps):
                e.call("compute_torques")
                e.call("simulate")
                fz.append(e.get("contact_forces")[:, e.setup.feet_indices, 2].sum(1))
            return e.get("root_states").copy(), np.array(fz)
        r_env, _ = settle(env, 0.56, 0.0, 400)
        r_ref, _ = settle(ref, 0.56, 0.0, 400)
        np.testing.assert_array_equal(r_env[0], r_ref[0])                     # defaults: bit-identical to the switch being off
        assert abs((r_env[1, 2] - r_env[0, 2]) - 0.02) < 2e-3, r_env[:, 2]      # rests one thickness higher
        assert 2e-4 < r_env[0, 2] - r_env[3, 2] < 5e-3, r_env[:, 2]             # 128 N per foot x 2e-6 m/N: sinks ~0.26 mm deeper
        _, fz = settle(env, 0.75, -1.5, 60)                                   # dropped: the feet arrive at ~2.5 m/s
        first = int(np.argmax(fz[:, 0] > 0))
        airborne = (fz[first:] == 0).sum(0)             # substeps without foot contact after the first touch-down
        assert first > 5 and airborne[0] == 0 and airborne[2] >= 5, (first, airborne)   # e = 0 sticks, e = 0.8 rebounds off the ground
    finally:
        env.close()
        ref.close()
