"""ctypes mirror of include/legged_hip.h (struct layouts, constants, prototypes).

Only declarations live here; ``lib.py`` loads the HIP library.  The CPU oracle under oracle/
(test infrastructure) reuses these struct definitions for its own ``lgo_*`` symbols.
"""
import ctypes as C

MAX_DOF, MAX_BODIES, MAX_SPHERES = 16, 24, 48
MAX_FEET, MAX_PEN, MAX_TERM = 8, 16, 8
LSTM_NW = 972
MAX_HIDDEN = 4

REWARD_NAMES = [
    "action_rate", "ang_vel_xy", "base_height", "collision", "dof_acc", "dof_pos_limits", "dof_vel",
    "dof_vel_limits", "feet_air_time", "feet_contact_forces", "lin_vel_z", "no_fly", "orientation",
    "stand_still", "stumble", "termination", "torque_limits", "torques", "tracking_ang_vel",
    "tracking_lin_vel"]
NUM_REWARDS = len(REWARD_NAMES)
assert REWARD_NAMES == sorted(REWARD_NAMES)

SLOT_CMD, SLOT_PUSH, SLOT_LEVEL, SLOT_DOF = 0, 3, 5, 6
MAX_XTERMS = 4
NUM_TERMS = NUM_REWARDS + MAX_XTERMS
XT_EXP_NEG_WSQ_ERR, XT_WSQ, XT_SLOPED_ERR_CHANGE = 1, 2, 3
SIGNALS = {"zero": (0, 8), "base_lin_vel": (1, 3), "base_ang_vel": (2, 3), "projected_gravity": (3, 3), "commands": (4, 4),
           "root_pos": (5, 3), "traj0": (6, 2), "prev_error": (7, 2), "dof_pos_rel": (8, MAX_DOF), "dof_vel": (9, MAX_DOF),
           "torques": (10, MAX_DOF), "actions": (11, MAX_DOF), "last_actions": (12, MAX_DOF)}      # name -> (lg_signal, length)
TRAJ_MAX_PTS, TG_NDRAW, TG_STRIDE = 17, 20, 28
TG_FIELDS = {"weights": (0, 4), "t_final": (4, 1), "t": (5, 1), "k": (6, 1), "const": (7, 2), "extreme": (9, 2),
             "ramp_t_start": (11, 1), "ramp_v_start": (12, 2), "ramp_v_end": (14, 2), "sin_mag": (16, 2), "sin_freq": (18, 2),
             "sin_off": (20, 2), "sin_mean": (22, 2), "stationary": (24, 1), "v": (25, 2)}                        # LG_TG_* offsets


def tslots(A):
    """LG_TSLOT_* of the trajectory env."""
    n = TG_NDRAW
    return {"tg": 0, "push": n, "timer": n + 2, "level": n + 3, "dof": n + 4, "xy": n + 4 + A, "vel": n + 6 + A,
            "romd": n + 12 + A, "rtg": n + 15 + A, "noise": 2 * n + 15 + A}


def slot_xy(A): return 6 + A
def slot_vel(A): return 8 + A
def slot_rcmd(A): return 14 + A
def slot_noise(A): return 17 + A


f32, i32, i64, u8, u64 = C.c_float, C.c_int32, C.c_int64, C.c_uint8, C.c_uint64
PF, PU8, PI64, PI32 = C.POINTER(f32), C.POINTER(u8), C.POINTER(i64), C.POINTER(i32)


class lg_model(C.Structure):
    _fields_ = [
        ("num_bodies", i32), ("num_dofs", i32), ("num_legs", i32), ("joints_per_leg", i32),
        ("num_spheres", i32), ("_pad0", i32 * 3),
        ("mass", f32 * (MAX_DOF + 1)), ("com", f32 * 3 * (MAX_DOF + 1)), ("inertia", f32 * 9 * (MAX_DOF + 1)),
        ("R_pj", f32 * 9 * MAX_DOF), ("p_pj", f32 * 3 * MAX_DOF), ("axis", f32 * 3 * MAX_DOF),
        ("q_lower", f32 * MAX_DOF), ("q_upper", f32 * MAX_DOF), ("effort", f32 * MAX_DOF),
        ("vel_limit", f32 * MAX_DOF), ("joint_damping", f32 * MAX_DOF),
        ("body_dyn", i32 * MAX_BODIES), ("sph_link", i32 * MAX_SPHERES), ("sph_body", i32 * MAX_SPHERES),
        ("sph_center", f32 * 3 * MAX_SPHERES), ("sph_radius", f32 * MAX_SPHERES)]


class lg_xterm(C.Structure):
    _fields_ = [("kind", i32), ("n", i32), ("sig_a", i32), ("off_a", i32), ("sig_b", i32), ("off_b", i32), ("sig_c", i32),
                ("off_c", i32), ("scale", f32), ("p", f32 * 3), ("w", f32 * 8)]


class lg_traj_cfg(C.Structure):
    _fields_ = [("enabled", i32), ("N", i32), ("dN", i32), ("randomize_rom_distance", i32),
                ("rom_dt", f32), ("t_low", f32), ("t_high", f32), ("freq_low", f32), ("freq_high", f32), ("prob_stationary", f32),
                ("zero_rom_dist_llh", f32), ("max_push_vel_xy", f32),
                ("v_min", f32 * 2), ("v_max", f32 * 2), ("obs_scale", f32 * 2), ("max_rom_dist", f32 * 2),
                ("push_t_lo", f32), ("push_t_hi", f32)]


class lg_cfg(C.Structure):
    _fields_ = [
        ("num_envs", i32), ("num_obs", i32), ("num_actions", i32), ("num_bodies", i32),
        ("num_feet", i32), ("num_pen", i32), ("num_term", i32), ("num_height_points", i32),
        ("feet_idx", i32 * MAX_FEET), ("pen_idx", i32 * MAX_PEN), ("term_idx", i32 * MAX_TERM),
        ("decimation", i32), ("control_type", i32), ("use_actuator_net", i32), ("heading_command", i32),
        ("max_episode_length", i32), ("resample_steps", i32), ("push_interval", i32), ("push_robots", i32),
        ("add_noise", i32), ("measure_heights", i32), ("only_positive_rewards", i32), ("send_timeouts", i32),
        ("terrain_type", i32), ("curriculum", i32), ("custom_origins", i32), ("max_terrain_level", i32),
        ("hf_rows", i32), ("hf_cols", i32), ("terrain_num_cols", i32), ("phys_substeps", i32),
        ("env_offset", i32), ("total_envs", i32), ("solver_iterations", i32), ("material_rand", i32),
        ("seed", u64),
        ("sim_dt", f32), ("dt", f32), ("action_scale", f32), ("clip_actions", f32), ("clip_obs", f32),
        ("max_push_vel", f32), ("episode_length_s", f32), ("ground_restitution", f32),
        ("cmd_lo", f32 * 4), ("cmd_hi", f32 * 4),
        ("obs_scale_lin_vel", f32), ("obs_scale_ang_vel", f32), ("obs_scale_dof_pos", f32),
        ("obs_scale_dof_vel", f32), ("obs_scale_height", f32),
        ("tracking_sigma", f32), ("soft_dof_vel_limit", f32), ("soft_torque_limit", f32),
        ("base_height_target", f32), ("max_contact_force", f32),
        ("hf_hscale", f32), ("hf_vscale", f32), ("border_size", f32), ("terrain_env_length", f32),
        ("rew_scale", f32 * NUM_REWARDS), ("base_init_state", f32 * 13),
        ("default_dof_pos", f32 * MAX_DOF), ("p_gains", f32 * MAX_DOF), ("d_gains", f32 * MAX_DOF),
        ("dof_pos_limits", f32 * 2 * MAX_DOF), ("dof_vel_limits", f32 * MAX_DOF), ("torque_limits", f32 * MAX_DOF),
        ("gravity", f32 * 3), ("ground_friction", f32),
        ("contact_offset", f32), ("max_depenetration_velocity", f32), ("contact_erp", f32), ("bounce_threshold", f32),
        ("max_linear_velocity", f32), ("max_angular_velocity", f32), ("armature", f32), ("rest_offset", f32),
        ("num_xterms", i32), ("feet_air_time_ungated", i32), ("num_terms", i32), ("_pad4", i32),
        ("term_order", i32 * NUM_TERMS), ("xterms", lg_xterm * MAX_XTERMS), ("traj", lg_traj_cfg),
        ("lstm_w", f32 * LSTM_NW),
        ("noise_vec", PF), ("height_points", PF), ("terrain_origins", PF)]


_BUF_FIELDS = [
    ("root_states", PF), ("dof_state", PF), ("contact_forces", PF), ("torques", PF), ("actions", PF),
    ("obs", PF), ("rew", PF), ("reset", PU8), ("time_out", PU8), ("episode_length", PI64),
    ("commands", PF), ("last_actions", PF), ("last_dof_vel", PF), ("last_root_vel", PF), ("feet_air_time", PF),
    ("last_contacts", PU8), ("episode_sums", PF), ("base_lin_vel", PF), ("base_ang_vel", PF),
    ("projected_gravity", PF), ("measured_heights", PF), ("env_origins", PF),
    ("terrain_levels", PI64), ("terrain_types", PI64), ("lstm_h", PF), ("lstm_c", PF),
    ("friction", PF), ("base_mass_delta", PF), ("extras_episode", PF), ("extras_terrain_level", PF),
    ("extras_time_outs", PU8), ("extras_episode_acc", PF), ("n_reset", PI32), ("n_fault", PI32), ("fault_total", PI64),
    ("n_vel_clamp", PI32), ("vel_clamp_total", PI64),
    ("tg_state", PF), ("tg_traj", PF), ("trajectory", PF), ("prev_error", PF), ("push_timer", PF),
    ("inject_uniforms", PF), ("inject_levels", PI64), ("material", PF)]


class lg_buffers(C.Structure):
    _fields_ = _BUF_FIELDS


class lg_stage(C.Structure):
    """Constants a curriculum stage rewrites (include/legged_hip.h lg_stage)."""
    _fields_ = [("cmd_lo", f32 * 4), ("cmd_hi", f32 * 4), ("max_push_vel", f32), ("_pad", f32), ("push_time", C.c_double),
                ("rew_scale", f32 * NUM_REWARDS), ("xterm_scale", f32 * MAX_XTERMS), ("xterm_p0", f32 * MAX_XTERMS),
                ("traj_v_min", f32 * 2), ("traj_v_max", f32 * 2), ("traj_t_low", f32), ("traj_t_high", f32),
                ("traj_max_rom_dist", f32 * 2)]


def buffer_shapes(N, A, B, O, F, H, traj_N=0, traj_dN=1):
    """name -> (shape, numpy dtype string) of every lg_buffers entry."""
    K = (tslots(A)["noise"] if traj_N else slot_noise(A)) + O
    return {
        "tg_state": ((N, TG_STRIDE), "f4"), "tg_traj": ((N, traj_N * traj_dN + 1, 2), "f4"),
        "trajectory": ((N, max(traj_N, 1), 2), "f4"), "prev_error": ((N, 2), "f4"), "push_timer": ((N,), "f4"),
        "root_states": ((N, 13), "f4"), "dof_state": ((N, A, 2), "f4"), "contact_forces": ((N, B, 3), "f4"),
        "torques": ((N, A), "f4"), "actions": ((N, A), "f4"), "obs": ((N, O), "f4"), "rew": ((N,), "f4"),
        "reset": ((N,), "u1"), "time_out": ((N,), "u1"), "episode_length": ((N,), "i8"),
        "commands": ((N, 4), "f4"), "last_actions": ((N, A), "f4"), "last_dof_vel": ((N, A), "f4"),
        "last_root_vel": ((N, 6), "f4"), "feet_air_time": ((N, F), "f4"), "last_contacts": ((N, F), "u1"),
        "episode_sums": ((NUM_TERMS, N), "f4"), "base_lin_vel": ((N, 3), "f4"), "base_ang_vel": ((N, 3), "f4"),
        "projected_gravity": ((N, 3), "f4"), "measured_heights": ((N, max(H, 1)), "f4"),
        "env_origins": ((N, 3), "f4"), "terrain_levels": ((N,), "i8"), "terrain_types": ((N,), "i8"),
        "lstm_h": ((2, N * A, 8), "f4"), "lstm_c": ((2, N * A, 8), "f4"), "friction": ((N,), "f4"),
        "base_mass_delta": ((N,), "f4"), "extras_episode": ((NUM_TERMS,), "f4"),
        "extras_terrain_level": ((1,), "f4"), "extras_time_outs": ((N,), "u1"),
        "extras_episode_acc": ((NUM_TERMS + 2,), "f4"), "n_reset": ((1,), "i4"),
        "n_fault": ((1,), "i4"), "fault_total": ((1,), "i8"), "n_vel_clamp": ((1,), "i4"), "vel_clamp_total": ((1,), "i8"),
        "inject_uniforms": ((N, K), "f4"), "inject_levels": ((N,), "i8"), "material": ((N, 4), "f4")}


class lg_ppo_cfg(C.Structure):
    _fields_ = [
        ("num_envs", i32), ("num_obs", i32), ("num_critic_obs", i32), ("num_actions", i32),
        ("num_hidden", i32), ("actor_hidden", i32 * MAX_HIDDEN), ("critic_hidden", i32 * MAX_HIDDEN),
        ("activation", i32), ("num_steps", i32), ("num_epochs", i32), ("num_mini_batches", i32),
        ("adaptive_schedule", i32), ("use_clipped_value_loss", i32), ("world_size", i32), ("_pad", i32),
        ("seed", u64),
        ("init_noise_std", f32), ("value_loss_coef", f32), ("clip_param", f32), ("entropy_coef", f32),
        ("learning_rate", f32), ("gamma", f32), ("lam", f32), ("desired_kl", f32), ("max_grad_norm", f32),
        ("_padf", f32)]


class lg_ppo_buffers(C.Structure):
    _fields_ = [
        ("params", PF), ("grads", PF), ("adam_m", PF), ("adam_v", PF),
        ("obs", PF), ("critic_obs", PF), ("actions", PF), ("rewards", PF), ("values", PF), ("returns", PF),
        ("advantages", PF), ("log_prob", PF), ("mu", PF), ("sigma", PF), ("dones", PU8),
        ("act_actions", PF), ("act_values", PF), ("act_log_prob", PF), ("act_mu", PF),
        ("stats", PF), ("noise", PF), ("perm", PI32), ("adv_partial", PF),
        ("cur_reward_sum", PF), ("cur_episode_len", PF), ("ep_stats", PF), ("ep_ring", PF), ("ep_ring_count", PI32),
        ("num_params", i64), ("num_reduce", i64)]


def declare_env_api(lib, prefix="lg_"):
    """Attach argtypes/restypes for the env entry points on a loaded library."""
    vp = C.c_void_p
    g = lambda n: getattr(lib, prefix + n)
    g("last_error").restype = C.c_char_p
    g("create").argtypes = [C.POINTER(lg_cfg), C.POINTER(lg_model), C.c_void_p, C.POINTER(vp)]
    g("destroy").argtypes = [vp]
    g("get_buffers").argtypes = [vp, C.POINTER(lg_buffers)]
    g("set_step_counter").argtypes = [vp, i64]
    g("get_step_counter").argtypes = [vp]
    g("get_step_counter").restype = i64
    g("set_init_done").argtypes = [vp, C.c_int]
    g("inject_uniforms").argtypes = [vp, C.c_int]
    g("step").argtypes = [vp, vp]
    g("set_actions").argtypes = [vp, vp]
    for n in ("compute_torques", "simulate", "post_physics_step", "reset_all"):
        g(n).argtypes = [vp]
    g("reset_ids").argtypes = [vp, vp, C.c_int]
    if prefix == "lg_":
        g("finalize").argtypes = [vp]
    g("get_stage").argtypes = [vp, C.POINTER(lg_stage)]
    g("set_curriculum_stage").argtypes = [vp, C.POINTER(lg_stage), C.c_int]
    if prefix == "lg_":
        g("set_stream").argtypes = [vp, vp]
        lib.lg_version.restype = C.c_int


ENV_SYMBOLS = ["last_error", "create", "destroy", "get_buffers", "set_step_counter", "get_step_counter",
               "set_init_done", "inject_uniforms", "step", "set_actions", "compute_torques", "simulate",
               "post_physics_step", "reset_all", "reset_ids", "get_stage", "set_curriculum_stage"]
PPO_SYMBOLS = ["ppo_create", "ppo_destroy", "ppo_get_buffers", "ppo_set_stream", "ppo_param_layout",
               "ppo_inject_noise", "ppo_act", "ppo_process_env_step", "ppo_compute_returns",
               "ppo_normalize_advantages", "ppo_begin_update", "ppo_minibatch_backward", "ppo_minibatch_step",
               "ppo_end_update", "ppo_act_inference"]
HIP_ONLY_SYMBOLS = ["version", "set_stream"]
COMM_SYMBOLS = ["comm_get_unique_id", "comm_init", "comm_destroy", "comm_rank", "comm_size", "comm_allreduce_sum", "comm_broadcast",
                "ppo_set_comm", "ppo_allreduce_adv_moments", "ppo_broadcast_params"]
