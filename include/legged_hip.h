/* legged_hip.h -- C-ABI of liblegged_hip.so (MI355X / gfx950).
 *
 * The reference has no FFI boundary on this path: the hot loop sits behind two Python
 * duck-typed interfaces:
 *   B1  VecEnv            legged_gym/envs/base/base_task.py:60-81,101-122,
 *                         legged_gym/envs/base/legged_robot.py:80-104
 *   B3  gym tensor API    the ~45 gym.* calls of legged_robot.py (acquire_*_tensor :537-539,
 *                         set_dof_actuation_force_tensor :92, simulate :93, refresh_* :96,111-112,
 *                         set_*_tensor_indexed :428,452,461 ...)
 * and, for the learner, rsl_rl's PPO/RolloutStorage/OnPolicyRunner (call sites
 * legged_gym/utils/task_registry.py:148-155).  This header is the C-ABI introduced UNDERNEATH
 * them; each entry cites the reference interface it replaces.  Plain pointers and PODs only,
 * int return codes (0 = ok, negative = error, text via lg_last_error()), no exceptions cross the
 * boundary.  One context = one HIP device + one stream; a context is not thread-safe.
 *
 * All device pointers handed in or out are HBM addresses on the context's device.
 * The same structs (with host pointers) are used by the CPU oracle in oracle/ (lgo_* symbols),
 * which is test infrastructure and never linked into this library.
 */
#ifndef LEGGED_HIP_H
#define LEGGED_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LG_MAX_DOF      16
#define LG_MAX_BODIES   24
#define LG_MAX_SPHERES  48
#define LG_MAX_FEET      8
#define LG_MAX_PEN      16
#define LG_MAX_TERM      8
#define LG_LSTM_NW     972   /* actuator net: see lg_cfg.lstm_w */
#define LG_MAX_HIDDEN    4

/* Reward terms in the order the reference sums them: alphabetical by name (class_to_dict walks
 * dir(), legged_gym/utils/helpers.py:111-126); "termination" is applied last, after the optional
 * positive clip (legged_robot.py:189-206). */
enum lg_reward {
    LG_REW_ACTION_RATE = 0, LG_REW_ANG_VEL_XY, LG_REW_BASE_HEIGHT, LG_REW_COLLISION, LG_REW_DOF_ACC,
    LG_REW_DOF_POS_LIMITS, LG_REW_DOF_VEL, LG_REW_DOF_VEL_LIMITS, LG_REW_FEET_AIR_TIME,
    LG_REW_FEET_CONTACT_FORCES, LG_REW_LIN_VEL_Z, LG_REW_NO_FLY, LG_REW_ORIENTATION, LG_REW_STAND_STILL,
    LG_REW_STUMBLE, LG_REW_TERMINATION, LG_REW_TORQUE_LIMITS, LG_REW_TORQUES, LG_REW_TRACKING_ANG_VEL,
    LG_REW_TRACKING_LIN_VEL, LG_NUM_REWARDS
};

/* Per-env random-draw slots.  In normal operation slot s of env e at step k is
 * Philox4x32-10(key = seed, counter = (global env id, k, s / 4, 0))[s % 4] >> 8 scaled to [0,1);
 * with lg_inject_uniforms() the value is read from the injected (N, K) buffer instead (parity
 * tests replay the reference's torch.rand draws this way).  K = LG_SLOT_NOISE(A) + num_obs
 * (LG_TSLOT_NOISE(A) + num_obs for the trajectory env). */
#define LG_SLOT_CMD        0   /* 3: callback command resample x, y, yaw|heading (legged_robot.py:365-387) */
#define LG_SLOT_PUSH       3   /* 2: push velocity xy (:456-461) */
#define LG_SLOT_LEVEL      5   /* 1: terrain level when the curriculum wraps (:479-483) */
#define LG_SLOT_DOF        6   /* A: reset joint position factors (:415-430) */
#define LG_SLOT_XY(A)     (6 + (A))        /* 2: reset xy offset on terrain (:443-446) */
#define LG_SLOT_VEL(A)    (8 + (A))        /* 6: reset base twist (:451) */
#define LG_SLOT_RCMD(A)   (14 + (A))       /* 3: command resample of reset envs */
#define LG_SLOT_NOISE(A)  (17 + (A))       /* num_obs: observation noise (:224-226) */

/* Collapsed articulated model: what gym.load_asset + get_asset_* give the reference
 * (legged_robot.py:693-724).  Produced by legged_gym_dev_amd/model/robot_model.py.
 * Dynamics links: index 0 = floating base, 1+d = link moved by DOF d.  Topology: L serial
 * chains of J revolute joints, DOF index = leg * J + joint. */
typedef struct lg_model {
    int32_t num_bodies, num_dofs, num_legs, joints_per_leg, num_spheres, _pad0[3];
    float mass[LG_MAX_DOF + 1];
    float com[LG_MAX_DOF + 1][3];         /* link frame */
    float inertia[LG_MAX_DOF + 1][9];     /* about com, link axes, row major */
    float R_pj[LG_MAX_DOF][9];            /* joint frame axes in the parent link frame (columns) */
    float p_pj[LG_MAX_DOF][3];
    float axis[LG_MAX_DOF][3];            /* unit, joint frame */
    float q_lower[LG_MAX_DOF], q_upper[LG_MAX_DOF];   /* equal -> no position limit */
    float effort[LG_MAX_DOF], vel_limit[LG_MAX_DOF], joint_damping[LG_MAX_DOF];
    int32_t body_dyn[LG_MAX_BODIES];      /* body -> dynamics link (-1 base, else DOF) */
    int32_t sph_link[LG_MAX_SPHERES];     /* -1 base, else DOF */
    int32_t sph_body[LG_MAX_SPHERES];     /* row of the net-contact-force tensor it reports to */
    float sph_center[LG_MAX_SPHERES][3];  /* dynamics-link frame */
    float sph_radius[LG_MAX_SPHERES];
} lg_model;

/* ---- Extra reward terms.  The reference binds ANY method named _reward_<name> of the env class to a non-zero
 * rewards.scales.<name> (legged_robot.py:605-629; cassie.py:43-46 and legged_robot_trajectory.py:1060-1110 add terms that way).
 * Python code cannot run inside the step kernel, so a subclass declares such a term as data: one of a few generic kinds over
 * named per-env signals.  The term takes part in the reward sum at its alphabetical position (lg_cfg.term_order) and gets its
 * own episode_sums / extras_episode row (LG_NUM_REWARDS + index). */
#define LG_MAX_XTERMS 4
#define LG_NUM_TERMS (LG_NUM_REWARDS + LG_MAX_XTERMS)
enum lg_xterm_kind {
    LG_XT_NONE = 0,
    LG_XT_EXP_NEG_WSQ_ERR,     /* exp(-sum_k w[k] (a[k] - b[k])^2 / p[0])          e.g. tracking_lin_vel, tracking_rom           */
    LG_XT_WSQ,                 /* sum_k w[k] a[k]^2                                 e.g. orientation, ang_vel_xy                  */
    LG_XT_SLOPED_ERR_CHANGE    /* d = |(a-b)^2|_2 - |c|_2 ; (d < 0 ? p[0] : p[1]) d e.g. differential_error (c = error at reset)  */
};
enum lg_signal {               /* per-env vectors a term may read (length): */
    LG_SIG_ZERO = 0, LG_SIG_BASE_LIN_VEL /*3*/, LG_SIG_BASE_ANG_VEL /*3*/, LG_SIG_PROJ_GRAVITY /*3*/, LG_SIG_COMMANDS /*4*/,
    LG_SIG_ROOT_POS /*3*/, LG_SIG_TRAJ0 /*2: first point of the reference trajectory*/, LG_SIG_PREV_ERROR /*2*/,
    LG_SIG_DOF_POS_REL /*A: q - default*/, LG_SIG_DOF_VEL /*A*/, LG_SIG_TORQUES /*A*/, LG_SIG_ACTIONS /*A*/, LG_SIG_LAST_ACTIONS /*A*/,
    LG_NUM_SIGNALS
};
typedef struct lg_xterm {
    int32_t kind, n;                      /* kind; vector length used (<= 8) */
    int32_t sig_a, off_a, sig_b, off_b, sig_c, off_c;   /* signals and first component */
    float scale;                          /* rewards.scales.<name> * dt */
    float p[3];
    float w[8];
} lg_xterm;

/* ---- Trajectory-tracking env variant (legged_robot_trajectory.py): the velocity commands are replaced by
 * a reference trajectory from a reduced-order model (trajopt/rom_dynamics.py: SingleInt2D, state = xy position, input = xy
 * velocity) driven by the random input generator TrajectoryGenerator (:441-616).  Per env the generator keeps four input
 * laws -- sample-and-hold, ramp, extreme (v_min | 0 | v_max), sinusoid -- mixed with random weights, all redrawn when the
 * env's hold time t_final runs out; the ROM integrates the mixed input every rom_dt, the env observes the N last points
 * interpolated at its own time.  Pushes come from per-env timers (:150-160). */
#define LG_TRAJ_MAX_PTS 17                /* N * dN + 1 points kept per env */
#define LG_TG_NDRAW 20                    /* uniforms of one generator resample: const 2, ramp 2, extreme 2, sin mag/mean/freq/off
                                             4 x 2, hold time 1, weights 4, stationary 1 */
/* layout of one row of lg_buffers.tg_state (LG_TG_STRIDE floats per env): */
#define LG_TG_W 0          /* 4 mixing weights */
#define LG_TG_T_FINAL 4
#define LG_TG_T 5
#define LG_TG_K 6
#define LG_TG_CONST 7      /* 2 */
#define LG_TG_EXTREME 9    /* 2 */
#define LG_TG_RAMP_T0 11
#define LG_TG_RAMP_V0 12   /* 2 */
#define LG_TG_RAMP_V1 14   /* 2 */
#define LG_TG_SIN_MAG 16   /* 2 */
#define LG_TG_SIN_FREQ 18  /* 2 */
#define LG_TG_SIN_OFF 20   /* 2 */
#define LG_TG_SIN_MEAN 22  /* 2 */
#define LG_TG_STATIONARY 24 /* 0 | 1 */
#define LG_TG_V 25         /* 2: the mixed input of the last evaluation (TrajectoryGenerator.v, what dataset rollouts log) */
#define LG_TG_STRIDE 28
typedef struct lg_traj_cfg {
    int32_t enabled, N, dN, randomize_rom_distance;
    float rom_dt, t_low, t_high, freq_low, freq_high, prob_stationary, zero_rom_dist_llh, max_push_vel_xy;
    float v_min[2], v_max[2], obs_scale[2], max_rom_dist[2];
    float push_t_lo, push_t_hi;           /* domain_rand.time_between_pushes */
} lg_traj_cfg;
/* uniform slots of the trajectory env (replace LG_SLOT_* when lg_cfg.traj.enabled): */
#define LG_TSLOT_TG        0                        /* LG_TG_NDRAW: generator resample in the step callback */
#define LG_TSLOT_PUSH      LG_TG_NDRAW              /* 2: push velocity xy */
#define LG_TSLOT_TIMER     (LG_TG_NDRAW + 2)        /* 1: next push time */
#define LG_TSLOT_LEVEL     (LG_TG_NDRAW + 3)
#define LG_TSLOT_DOF       (LG_TG_NDRAW + 4)        /* A */
#define LG_TSLOT_XY(A)     (LG_TG_NDRAW + 4 + (A))  /* 2 */
#define LG_TSLOT_VEL(A)    (LG_TG_NDRAW + 6 + (A))  /* 6 */
#define LG_TSLOT_ROMD(A)   (LG_TG_NDRAW + 12 + (A)) /* 3: start-offset mask draw + xy offset (:224-229) */
#define LG_TSLOT_RTG(A)    (LG_TG_NDRAW + 15 + (A)) /* LG_TG_NDRAW: generator resample of a reset env -- or of a NON-reset env
                                                       whose hold time ran out on a step where some env resets: the reference's
                                                       reset loop re-checks every env (rom_dynamics.py:571-574,598-608) */
#define LG_TSLOT_NOISE(A)  (2 * LG_TG_NDRAW + 15 + (A))

/* Flattened LeggedRobotCfg (+ what _parse_cfg/_init_buffers derive from it,
 * legged_robot.py:533-603,819-837). */
typedef struct lg_cfg {
    int32_t num_envs, num_obs, num_actions, num_bodies;
    int32_t num_feet, num_pen, num_term, num_height_points;
    int32_t feet_idx[LG_MAX_FEET], pen_idx[LG_MAX_PEN], term_idx[LG_MAX_TERM];
    int32_t decimation, control_type /*0 P,1 V,2 T*/, use_actuator_net, heading_command;
    int32_t max_episode_length, resample_steps, push_interval, push_robots;
    int32_t add_noise, measure_heights, only_positive_rewards, send_timeouts;
    int32_t terrain_type /*0 plane, 1 height samples*/, curriculum, custom_origins, max_terrain_level;
    int32_t hf_rows, hf_cols, terrain_num_cols, phys_substeps;
    int32_t env_offset, total_envs;       /* this shard's first global env id / envs over all ranks */
    int32_t solver_iterations;
    int32_t material_rand;                /* lg_buffers.material holds per-env restitution / compliance / thickness draws (else unread) */
    uint64_t seed;
    float sim_dt, dt, action_scale, clip_actions, clip_obs, max_push_vel, episode_length_s;
    float ground_restitution;             /* terrain.restitution; combined with the env's by averaging, like friction */
    float cmd_lo[4], cmd_hi[4];           /* lin_vel_x, lin_vel_y, ang_vel_yaw, heading */
    float obs_scale_lin_vel, obs_scale_ang_vel, obs_scale_dof_pos, obs_scale_dof_vel, obs_scale_height;
    float tracking_sigma, soft_dof_vel_limit, soft_torque_limit, base_height_target, max_contact_force;
    float hf_hscale, hf_vscale, border_size, terrain_env_length;
    float rew_scale[LG_NUM_REWARDS];      /* already multiplied by dt; 0 = term inactive */
    float base_init_state[13];
    float default_dof_pos[LG_MAX_DOF], p_gains[LG_MAX_DOF], d_gains[LG_MAX_DOF];
    float dof_pos_limits[LG_MAX_DOF][2];  /* soft limits (legged_robot.py:313-327) */
    float dof_vel_limits[LG_MAX_DOF], torque_limits[LG_MAX_DOF];
    float gravity[3], ground_friction;    /* ground mu; combined with the env's mu by averaging */
    float contact_offset, max_depenetration_velocity, contact_erp;
    float bounce_threshold;               /* sim.physx.bounce_threshold_velocity: approach speeds below it do not bounce */
    /* asset options the simulator is given (legged_robot.py:692-705): */
    float max_linear_velocity, max_angular_velocity;   /* :701-702: the base's velocities are clamped at these magnitudes after every
                                                          solve, as PhysX clamps a body's (0 = no clamp); each clamp is counted */
    float armature;                       /* :703: added to the inertia every joint sees about its own axis */
    float rest_offset;                    /* :704 thickness: the robot's shapes come to rest this far off a surface; replaced per env by
                                             lg_buffers.material[.][2] when that property is randomised (material_rand) */
    int32_t num_xterms, feet_air_time_ungated /* trajectory env: no command gate (legged_robot_trajectory.py:1071-1080) */;
    int32_t num_terms, _pad4;
    int32_t term_order[LG_NUM_TERMS];   /* active terms (builtin id, or LG_NUM_REWARDS + xterm index) in the order the
                                                             reference sums them (alphabetical); termination is not listed (applied last) */
    lg_xterm xterms[LG_MAX_XTERMS];
    lg_traj_cfg traj;
    float lstm_w[LG_LSTM_NW];             /* in_scale2 out_scale1 | w_ih0 64 w_hh0 256 b_ih0 32 b_hh0 32 |
                                             w_ih1 256 w_hh1 256 b_ih1 32 b_hh1 32 | lin_w 8 lin_b 1 */
    const float *noise_vec;               /* host, num_obs   (legged_robot.py:507-530) */
    const float *height_points;           /* host, num_height_points x 2, x-major grid (:861-875) */
    const float *terrain_origins;         /* host, rows(levels) x terrain_num_cols x 3, or NULL */
} lg_cfg;

/* State tensors.  Layouts are the reference's: root (N,13) =
 * [pos3, quat xyzw, lin vel3, ang vel3] world frame; dof_state (N,A,2) = [q, qdot] interleaved;
 * contact (N,B,3) world N; episode_sums is (LG_NUM_REWARDS + LG_MAX_XTERMS, N) so each term is a contiguous (N,). */
typedef struct lg_buffers {
    float *root_states, *dof_state, *contact_forces, *torques, *actions;
    float *obs, *rew;
    uint8_t *reset, *time_out;
    int64_t *episode_length;
    float *commands, *last_actions, *last_dof_vel, *last_root_vel, *feet_air_time;
    uint8_t *last_contacts;
    float *episode_sums, *base_lin_vel, *base_ang_vel, *projected_gravity, *measured_heights;
    float *env_origins;
    int64_t *terrain_levels, *terrain_types;
    float *lstm_h, *lstm_c;               /* (2, N*A, 8) each, anymal.py:62-69 */
    float *friction, *base_mass_delta;    /* per-env randomised constants (legged_robot.py:259-341) */
    /* extras: filled by the step's finalize pass; episode means only change on steps where at
     * least one env resets, time_outs likewise (the reference's stale-mask quirk, :156-157,186-187) */
    float *extras_episode;                /* LG_NUM_REWARDS + LG_MAX_XTERMS */
    float *extras_terrain_level;          /* 1 */
    uint8_t *extras_time_outs;            /* N */
    float *extras_episode_acc;            /* LG_NUM_REWARDS + LG_MAX_XTERMS + 2: running sums over steps of extras_episode, of extras_terrain_level
                                             and the number of steps summed -- rsl_rl's log() averages infos["episode"] over every
                                             step of an iteration; the reader divides and clears */
    int32_t *n_reset;                     /* 1: envs reset by the last step */
    int32_t *n_fault;                     /* 1: envs whose solve came back non-finite during the last step: brought to rest and reset (they are among n_reset) */
    int64_t *fault_total;                 /* 1: the same, summed since lg_create */
    int32_t *n_vel_clamp;                 /* 1: physics substeps of the last step in which an env's base velocity was clamped at
                                             max_linear_velocity / max_angular_velocity (the env carries on, as under PhysX) */
    int64_t *vel_clamp_total;             /* 1: the same, summed since lg_create */
    /* trajectory env (all unused otherwise): */
    float *tg_state;                      /* (N, LG_TG_STRIDE) generator state, LG_TG_* */
    float *tg_traj;                       /* (N, traj.N * traj.dN + 1, 2) ROM states, oldest first */
    float *trajectory;                    /* (N, traj.N, 2) what the env observes: interpolated at the env's time (:410-411) */
    float *prev_error;                    /* (N, 2) squared tracking error at the last reset (:199) */
    float *push_timer;                    /* (N) time_until_next_push (:150-160) */
    float *inject_uniforms;               /* (N, K) or unused */
    int64_t *inject_levels;               /* N */
    float *material;                      /* (N, 4) per-env randomised shape / body properties beside friction and base mass:
                                             [restitution, compliance, thickness, inverse base mass] (legged_robot.py:284-299,337-339);
                                             zeros = the asset's defaults */
} lg_buffers;

/* ---- Staged curriculum (legged_robot.py:360-363,488-505 base env: command ranges, push magnitude / period;
 * legged_robot_trajectory.py:414-417,519-553 trajectory env: reward scales, tracking sigma, ROM input bounds, hold-time sampler,
 * start-offset range).  The stage machine itself (curriculum_state, `common_step_counter % curriculum_steps[state] == 0`) is host
 * logic, as in the reference; a stage change rewrites these device constants.  In the reference the change happens at the END of
 * _post_physics_step_callback of the step that triggers it: the callback's own command resample / push / generator resample still
 * see the old values, everything after it (rewards, resets, observations of the same step) the new ones.
 * lg_set_curriculum_stage(.., in_callback = 1) reproduces exactly that for the next lg_step / lg_post_physics_step;
 * in_callback = 0 applies at once (the update at construction, legged_robot.py:828-829 / legged_robot_trajectory.py:78-79). */
typedef struct lg_stage {
    float cmd_lo[4], cmd_hi[4];           /* lin_vel_x, lin_vel_y, ang_vel_yaw, heading */
    float max_push_vel, _pad;
    double push_time;                     /* policy steps between pushes; may be fractional (nominal x multiplier): a push fires when
                                             step_counter % push_time == 0 in Python float arithmetic */
    float rew_scale[LG_NUM_REWARDS];      /* x dt, as lg_cfg.rew_scale */
    float xterm_scale[LG_MAX_XTERMS];     /* x dt, as lg_xterm.scale */
    float xterm_p0[LG_MAX_XTERMS];        /* lg_xterm.p[0] (tracking_rom: its sigma) */
    float traj_v_min[2], traj_v_max[2], traj_t_low, traj_t_high, traj_max_rom_dist[2];
} lg_stage;

typedef struct lg_ctx lg_ctx;

const char *lg_last_error(void);
int lg_version(void);

/* Replaces create_sim/_create_envs/prepare_sim/acquire_*_tensor (base_task.py:84-85,
 * legged_robot.py:228-245,537-539).  height_samples: host int16 (hf_rows x hf_cols) or NULL. */
int lg_create(const lg_cfg *cfg, const lg_model *model, const int16_t *height_samples, lg_ctx **out);
int lg_destroy(lg_ctx *ctx);
int lg_get_buffers(lg_ctx *ctx, lg_buffers *out);
/* stream = hipStream_t; all later calls enqueue on it (the caller's torch stream). */
int lg_set_stream(lg_ctx *ctx, void *stream);
/* counterpart of env.common_step_counter (legged_robot.py:115); set by tests/resume. */
int lg_set_step_counter(lg_ctx *ctx, int64_t counter);
int64_t lg_get_step_counter(lg_ctx *ctx);
int lg_set_init_done(lg_ctx *ctx, int init_done);     /* legged_robot.py:472-474 */
/* enable (1) / disable (0) replay of buffers.inject_uniforms / inject_levels. */
int lg_inject_uniforms(lg_ctx *ctx, int enable);

/* LeggedRobot.step (legged_robot.py:80-104): clip actions, decimation x {torque law, physics
 * substep}, post_physics_step, clip observations.  actions: device (N, A) f32.
 * The clip + decimation loop is one kernel launch (state resident on chip).  lg_compute_torques and lg_simulate run
 * the same kernel with one stage switched off, so the result equals the sequence lg_set_actions, decimation x
 * {lg_compute_torques, lg_simulate}, lg_post_physics_step bit for bit. */
int lg_step(lg_ctx *ctx, const float *actions);
/* Finer-grained entry points (tests, teacher forcing): */
int lg_set_actions(lg_ctx *ctx, const float *actions);         /* :86-87 */
int lg_compute_torques(lg_ctx *ctx);                           /* :91 (PD :389-413 / LSTM anymal.py:71-81) */
int lg_simulate(lg_ctx *ctx);                                  /* :92-96, one sim_dt of physics */
int lg_post_physics_step(lg_ctx *ctx);                         /* :106-137 + obs clip :100-103 */
int lg_reset_all(lg_ctx *ctx);                                 /* reset_idx(arange(N)), base_task.py:113 */
/* reset_idx(env_ids) for an arbitrary subset (legged_robot.py:147-187): terrain curriculum, _reset_dofs + _reset_root_states
 * (what the reference pushes through set_dof_state_tensor_indexed :428 / set_actor_root_state_tensor_indexed :452, with the
 * same int32 id tensor), command resample, buffer clears, actuator-net state (anymal.py:56-60), extras["episode"] means
 * over the ids and extras["time_outs"].  ids: DEVICE int32[n], local env indices, no duplicates.  n == 0 returns at once
 * (legged_robot.py:156-157).  Draws come from the env's reset slots at the current step counter. */
int lg_reset_ids(lg_ctx *ctx, const int32_t *ids, int n);
/* Runs the step's single-workgroup epilogue (extras["episode"], extras["time_outs"], counters) if a learner attached with
 * lg_ppo_attach_env left it pending; a no-op otherwise.  Every env entry point does this itself before it touches the env. */
int lg_finalize(lg_ctx *ctx);
int lg_get_stage(lg_ctx *ctx, lg_stage *out);          /* the values in force (at creation: what lg_cfg holds) */
int lg_set_curriculum_stage(lg_ctx *ctx, const lg_stage *stage, int in_callback);

/* ------------------------------------------------------------------ PPO (rsl_rl v1.0.2 semantics,
 * SURVEY.md Appendix B; call sites task_registry.py:148-155, scripts/train.py:44) */
typedef struct lg_ppo_cfg {
    int32_t num_envs, num_obs, num_critic_obs, num_actions;
    int32_t num_hidden, actor_hidden[LG_MAX_HIDDEN], critic_hidden[LG_MAX_HIDDEN];
    int32_t activation /*0 elu, 1 selu, 2 relu, 3 lrelu, 4 tanh, 5 sigmoid (rsl_rl get_activation; its "crelu" is nn.ReLU: 2)*/, num_steps, num_epochs, num_mini_batches;
    int32_t adaptive_schedule, use_clipped_value_loss, world_size, _pad;
    uint64_t seed;
    float init_noise_std, value_loss_coef, clip_param, entropy_coef, learning_rate;
    float gamma, lam, desired_kl, max_grad_norm, _padf;
} lg_ppo_cfg;

typedef struct lg_ppo_buffers {
    float *params, *grads, *adam_m, *adam_v;      /* flat, num_params (+ tail, see num_reduce) */
    float *obs, *critic_obs, *actions, *rewards, *values, *returns, *advantages, *log_prob, *mu, *sigma;
    uint8_t *dones;                                /* storage, time major (T, N, .) */
    float *act_actions, *act_values, *act_log_prob, *act_mu;   /* outputs of the last act() */
    float *stats;                                  /* [lr, kl, value_loss, surrogate_loss, mean_std, n_updates, adv_mean, adv_std] */
    float *noise;                                  /* (N, A) injected N(0,1) for act(), or unused */
    int32_t *perm;                                 /* minibatch permutation (T*N) */
    float *adv_partial;                            /* [sum, sumsq, count] for cross-rank normalisation */
    float *cur_reward_sum, *cur_episode_len;       /* (N) running episode return / length (runner logging) */
    float *ep_stats;                               /* [sum return, sum length, count] of episodes finished since cleared */
    float *ep_ring;                                /* (2, 100): returns / lengths of the last 100 finished episodes (rsl_rl's rewbuffer and
                                                      lenbuffer deques); slot = finish order % 100, unordered within one step */
    int32_t *ep_ring_count;                        /* 1: episodes finished since creation */
    int64_t num_params, num_reduce;                /* floats to all-reduce per optimiser step */
} lg_ppo_buffers;

typedef struct lg_ppo lg_ppo;

int lg_ppo_create(const lg_ppo_cfg *cfg, lg_ppo **out);
int lg_ppo_destroy(lg_ppo *p);
int lg_ppo_get_buffers(lg_ppo *p, lg_ppo_buffers *out);
int lg_ppo_set_stream(lg_ppo *p, void *stream);
int lg_ppo_param_layout(lg_ppo *p, int64_t *offsets, int64_t *shapes, int max_entries); /* returns #tensors */
int lg_ppo_inject_noise(lg_ppo *p, int enable);
/* PPO.act: actor+critic forward, sample, log-prob; stores the transition at the current step. */
int lg_ppo_act(lg_ppo *p, const float *obs, const float *critic_obs);
/* PPO.process_env_step: reward += gamma * V * time_outs; store reward/done; advance step. */
int lg_ppo_process_env_step(lg_ppo *p, const float *rew, const uint8_t *dones, const uint8_t *time_outs);
/* PPO.compute_returns: bootstrap value, GAE reverse scan, local advantage sums. */
int lg_ppo_compute_returns(lg_ppo *p, const float *last_critic_obs);
int lg_ppo_normalize_advantages(lg_ppo *p);        /* after adv_partial was (all-)reduced */
/* PPO.update split so the caller can all-reduce grads between the two halves: */
int lg_ppo_begin_update(lg_ppo *p);                /* new permutation, zero loss stats; re-derives the bf16 weight planes
                                                      from params (so params may be written through lg_ppo_buffers
                                                      between updates: checkpoint load, broadcast) */
int lg_ppo_minibatch_backward(lg_ppo *p, int epoch, int mb);   /* fwd, loss, bwd -> grads (+KL tail) */
int lg_ppo_minibatch_step(lg_ppo *p);              /* KL-adaptive lr, clip_grad_norm, Adam; clears grads */
int lg_ppo_end_update(lg_ppo *p);                  /* finalise mean losses, clear storage */
/* Rollout fusion (optional; OnPolicyRunner.rollout switches it on for the duration of a rollout).  With an env attached, lg_step
 * leaves its single-workgroup epilogue pending and lg_ppo_process_env_step only records its arguments; both run inside the launch
 * of the NEXT lg_ppo_act (extra workgroups beside the two MLPs: 5 launches per policy step become 3).  Any other entry point of
 * either object, and lg_ppo_attach_env(p, NULL), first runs what is pending the ordinary way, so results are identical.
 * While attached, extras / n_reset / the logging sums of a step become visible with the next lg_ppo_act (or flush).
 * Contract: env and learner run on the SAME stream (checked: the fused epilogue reads the env's step outputs in stream order), and
 * the env outlives the attachment -- lg_destroy refuses an env that is still attached; detach with lg_ppo_attach_env(p, NULL) or
 * destroy the learner first. */
int lg_ppo_attach_env(lg_ppo *p, lg_ctx *env);
/* The caller wrote lg_ppo_buffers.params itself (checkpoint load, a broadcast of its own): the weight images the rollout forward
 * reads are re-derived by the next lg_ppo_act.  (lg_ppo_minibatch_step and lg_ppo_broadcast_params mark them stale themselves.) */
int lg_ppo_params_changed(lg_ppo *p);
/* Reproducible runs (debugging aid; off by default).  The learner's sums over the minibatch rows -- weight-gradient slices, bias
 * column sums, the head's row sums, loss statistics, the gradient norm, the advantage moments -- are float atomics, whose order
 * differs from run to run (last-bit differences in every gradient).  on != 0: the same contributions are accumulated as 2^-40
 * fixed-point 64-bit integers (order-independent) and folded into the float buffers before they are read: two runs from the
 * same seed then agree bit for bit, on one rank and across a fixed set of ranks.  Costs three small launches per optimiser step.
 * Not combinable with gradient buckets reduced inside the backward pass (lg_ppo_set_comm): returns an error then. */
int lg_ppo_set_deterministic(lg_ppo *p, int on);
/* actor mean only (act_inference) for play/eval */
int lg_ppo_act_inference(lg_ppo *p, const float *obs, float *actions_out, int64_t rows);

/* ------------------------------------------------------------------ collectives (nothing in the reference to
 * replace -- it has no multi-GPU code; this is the exchange step the env-sharded learner needs).  One process per GPU; rank r
 * owns envs [r N/G, (r+1) N/G).  RCCL over xGMI, resolved from the librccl.so already in the process (no link dependency).
 * The Python runner may use torch.distributed instead (same RCCL underneath); these entries are for hosts without it, and for
 * the reduction overlapped with the backward pass, which needs the learner's own streams. */
#define LG_COMM_ID_BYTES 128
typedef struct lg_comm lg_comm;
int lg_comm_get_unique_id(void *id_out /* LG_COMM_ID_BYTES, host */);   /* rank 0; the host carries the bytes to the other ranks */
int lg_comm_init(int rank, int nranks, const void *id, lg_comm **out);  /* collective over all ranks; device = hipGetDevice() */
int lg_comm_destroy(lg_comm *c);
int lg_comm_rank(lg_comm *c);
int lg_comm_size(lg_comm *c);
int lg_comm_allreduce_sum(lg_comm *c, float *buf /* device, in place */, int64_t n, void *stream);
int lg_comm_broadcast(lg_comm *c, float *buf, int64_t n, int root, void *stream);
/* The learner's three exchanges.  lg_ppo_set_comm(p, c) makes lg_ppo_minibatch_backward reduce the gradients itself, layer by
 * layer as their weight-gradient GEMMs finish (the head's bucket carries std and the KL sum), on the communicator's stream
 * beside the remaining backward GEMMs; lg_ppo_minibatch_step then waits for the last bucket.  c = NULL switches it off. */
int lg_ppo_set_comm(lg_ppo *p, lg_comm *c);
int lg_ppo_allreduce_adv_moments(lg_ppo *p, lg_comm *c);   /* between lg_ppo_compute_returns and lg_ppo_normalize_advantages */
int lg_ppo_broadcast_params(lg_ppo *p, lg_comm *c, int root);   /* identical initial policy on every rank */
/* How long the learner's stream stood waiting for the gradient buckets (what the overlap did NOT hide), measured with HIP events
 * around the wait of every minibatch while timing is enabled (at most 4096 minibatches are recorded, then recording stops).
 * lg_ppo_comm_wait_ms synchronises the learner's stream, returns the sum in ms and the number of minibatches summed, and clears. */
int lg_ppo_comm_timing(lg_ppo *p, int enable);
int lg_ppo_comm_wait_ms(lg_ppo *p, double *ms_total, int64_t *minibatches);

#ifdef __cplusplus
}
#endif
#endif /* LEGGED_HIP_H */
