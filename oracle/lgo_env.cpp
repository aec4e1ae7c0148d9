// CPU ORACLE -- TEST INFRASTRUCTURE ONLY (see lgo_common.h).
// Scalar restatement of the reference's env logic; every block cites the lines it follows in
// /root/reference/legged_gym/envs/base/legged_robot.py ("LR"), envs/anymal_c/anymal.py ("AN"),
// envs/cassie/cassie.py ("CA"), utils/math.py ("MA").  Pinned by tests/golden/*.npz, which were
// produced by running those files themselves (oracle/gen_fixtures.py).
#include "lgo_common.h"

#include <algorithm>

namespace lgo {

static inline void cross3(const float a[3], const float b[3], float o[3]) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
// isaacgym.torch_utils.quat_rotate_inverse: a - b + c
static inline void quat_rotate_inverse(const float q[4], const float v[3], float o[3]) {
    float w = q[3], s = 2.0f * w * w - 1.0f, cr[3];
    cross3(q, v, cr);
    float d = q[0] * v[0] + q[1] * v[1] + q[2] * v[2];
    for (int i = 0; i < 3; ++i) o[i] = v[i] * s - cr[i] * w * 2.0f + q[i] * d * 2.0f;
}
// isaacgym.torch_utils.quat_apply: b + w t + xyz x t, t = 2 xyz x b
static inline void quat_apply(const float q[4], const float b[3], float o[3]) {
    float t[3], u[3];
    cross3(q, b, t);
    for (int i = 0; i < 3; ++i) t[i] *= 2.0f;
    cross3(q, t, u);
    for (int i = 0; i < 3; ++i) o[i] = b[i] + q[3] * t[i] + u[i];
}
static inline float clampf(float x, float lo, float hi) { return std::min(std::max(x, lo), hi); }


// LR:365-387
static void resample_commands(Env &e, int i, int slot0, const float *lo, const float *hi) {
    const lg_cfg &c = e.cfg;
    float *cmd = &e.commands[(size_t)i * 4];
    cmd[0] = (hi[0] - lo[0]) * uni(e, i, slot0 + 0) + lo[0];
    cmd[1] = (hi[1] - lo[1]) * uni(e, i, slot0 + 1) + lo[1];
    if (c.heading_command)
        cmd[3] = (hi[3] - lo[3]) * uni(e, i, slot0 + 2) + lo[3];
    else
        cmd[2] = (hi[2] - lo[2]) * uni(e, i, slot0 + 2) + lo[2];
    float nrm = std::sqrt(cmd[0] * cmd[0] + cmd[1] * cmd[1]);
    float keep = nrm > 0.2f ? 1.0f : 0.0f;
    cmd[0] *= keep;
    cmd[1] *= keep;
}

// LR:877-915 with MA:38-42 (quat_apply_yaw)
static void get_heights(Env &e, int i) {
    const lg_cfg &c = e.cfg;
    float *out = &e.heights[(size_t)i * e.H];
    if (c.terrain_type == 0) {
        for (int h = 0; h < e.H; ++h) out[h] = 0.0f;
        return;
    }
    const float *r = &e.root[(size_t)i * 13];
    float qy[4] = {0.0f, 0.0f, r[5], r[6]};
    float n = std::sqrt(qy[2] * qy[2] + qy[3] * qy[3]);
    n = std::max(n, 1e-9f);
    qy[2] /= n;
    qy[3] /= n;
    for (int h = 0; h < e.H; ++h) {
        float p[3] = {e.height_points[2 * h], e.height_points[2 * h + 1], 0.0f}, w[3];
        quat_apply(qy, p, w);
        float x = (w[0] + r[0] + c.border_size) / c.hf_hscale;
        float y = (w[1] + r[1] + c.border_size) / c.hf_hscale;
        long px = (long)x, py = (long)y;                      // .long(): truncation toward zero
        px = std::min(std::max(px, 0L), (long)c.hf_rows - 2);
        py = std::min(std::max(py, 0L), (long)c.hf_cols - 2);
        int16_t h1 = e.height_samples[px * c.hf_cols + py];
        int16_t h2 = e.height_samples[(px + 1) * c.hf_cols + py];
        int16_t h3 = e.height_samples[px * c.hf_cols + py + 1];
        out[h] = (float)std::min(std::min(h1, h2), h3) * c.hf_vscale;
    }
}

static inline float fnorm3(const float *f) { return std::sqrt(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]); }

struct RewardCtx {
    const float *blv, *bav, *pg, *cmd, *q, *qd, *tau, *act, *lact, *lqd, *cf;
    float root_z;
};

// LR:918-1015, CA:43-46.  Stateful term feet_air_time mutates last_contacts / feet_air_time.
static float reward_term(Env &e, int i, int k, const RewardCtx &x) {
    const lg_cfg &c = e.cfg;
    const int A = e.A, F = e.F;
    float s = 0.0f;
    switch (k) {
    case LG_REW_LIN_VEL_Z: return x.blv[2] * x.blv[2];
    case LG_REW_ANG_VEL_XY: return x.bav[0] * x.bav[0] + x.bav[1] * x.bav[1];
    case LG_REW_ORIENTATION: return x.pg[0] * x.pg[0] + x.pg[1] * x.pg[1];
    case LG_REW_BASE_HEIGHT: {
        for (int h = 0; h < e.H; ++h) s += x.root_z - e.heights[(size_t)i * e.H + h];
        float bh = s / (float)e.H;
        return (bh - c.base_height_target) * (bh - c.base_height_target);
    }
    case LG_REW_TORQUES: for (int j = 0; j < A; ++j) s += x.tau[j] * x.tau[j]; return s;
    case LG_REW_DOF_VEL: for (int j = 0; j < A; ++j) s += x.qd[j] * x.qd[j]; return s;
    case LG_REW_DOF_ACC:
        for (int j = 0; j < A; ++j) { float a = (x.lqd[j] - x.qd[j]) / c.dt; s += a * a; }
        return s;
    case LG_REW_ACTION_RATE:
        for (int j = 0; j < A; ++j) { float a = x.lact[j] - x.act[j]; s += a * a; }
        return s;
    case LG_REW_COLLISION:
        for (int b = 0; b < c.num_pen; ++b) s += fnorm3(x.cf + 3 * c.pen_idx[b]) > 0.1f ? 1.0f : 0.0f;
        return s;
    case LG_REW_TERMINATION: return (e.reset[i] && !e.time_out[i]) ? 1.0f : 0.0f;
    case LG_REW_DOF_POS_LIMITS:
        for (int j = 0; j < A; ++j) {
            float o = -std::min(x.q[j] - c.dof_pos_limits[j][0], 0.0f);
            o += std::max(x.q[j] - c.dof_pos_limits[j][1], 0.0f);
            s += o;
        }
        return s;
    case LG_REW_DOF_VEL_LIMITS:
        for (int j = 0; j < A; ++j)
            s += clampf(std::fabs(x.qd[j]) - c.dof_vel_limits[j] * c.soft_dof_vel_limit, 0.0f, 1.0f);
        return s;
    case LG_REW_TORQUE_LIMITS:
        for (int j = 0; j < A; ++j)
            s += std::max(std::fabs(x.tau[j]) - c.torque_limits[j] * c.soft_torque_limit, 0.0f);
        return s;
    case LG_REW_TRACKING_LIN_VEL: {
        float dx = x.cmd[0] - x.blv[0], dy = x.cmd[1] - x.blv[1];
        return std::exp(-(dx * dx + dy * dy) / c.tracking_sigma);
    }
    case LG_REW_TRACKING_ANG_VEL: {
        float d = x.cmd[2] - x.bav[2];
        return std::exp(-(d * d) / c.tracking_sigma);
    }
    case LG_REW_FEET_AIR_TIME: {
        float *air = &e.feet_air_time[(size_t)i * F];
        uint8_t *lc = &e.last_contacts[(size_t)i * F];
        for (int f = 0; f < F; ++f) {
            bool contact = x.cf[3 * c.feet_idx[f] + 2] > 1.0f;
            bool filt = contact || lc[f];
            lc[f] = contact;
            float first = (air[f] > 0.0f && filt) ? 1.0f : 0.0f;
            air[f] += c.dt;
            s += (air[f] - 0.5f) * first;
            air[f] *= filt ? 0.0f : 1.0f;
        }
        if (c.feet_air_time_ungated) return s;                        // LT:1071-1080: no command gate in the trajectory env
        float cn = std::sqrt(x.cmd[0] * x.cmd[0] + x.cmd[1] * x.cmd[1]);
        return s * (cn > 0.1f ? 1.0f : 0.0f);
    }
    case LG_REW_STUMBLE: {
        bool any = false;
        for (int f = 0; f < F; ++f) {
            const float *ff = x.cf + 3 * c.feet_idx[f];
            any |= std::sqrt(ff[0] * ff[0] + ff[1] * ff[1]) > 5.0f * std::fabs(ff[2]);
        }
        return any ? 1.0f : 0.0f;
    }
    case LG_REW_STAND_STILL: {
        for (int j = 0; j < A; ++j) s += std::fabs(x.q[j] - c.default_dof_pos[j]);
        float cn = std::sqrt(x.cmd[0] * x.cmd[0] + x.cmd[1] * x.cmd[1]);
        return s * (cn < 0.1f ? 1.0f : 0.0f);
    }
    case LG_REW_FEET_CONTACT_FORCES:
        for (int f = 0; f < F; ++f) s += std::max(fnorm3(x.cf + 3 * c.feet_idx[f]) - c.max_contact_force, 0.0f);
        return s;
    case LG_REW_NO_FLY: {
        int n = 0;
        for (int f = 0; f < F; ++f) n += x.cf[3 * c.feet_idx[f] + 2] > 0.1f ? 1 : 0;
        return n == 1 ? 1.0f : 0.0f;
    }
    }
    return 0.0f;
}

// per-env signals an extra reward term may read (include/legged_hip.h lg_signal)
static float signal_at(const Env &e, int i, int sig, int k, const RewardCtx &x) {
    const lg_cfg &c = e.cfg;
    switch (sig) {
    case LG_SIG_BASE_LIN_VEL: return x.blv[k];
    case LG_SIG_BASE_ANG_VEL: return x.bav[k];
    case LG_SIG_PROJ_GRAVITY: return x.pg[k];
    case LG_SIG_COMMANDS: return x.cmd[k];
    case LG_SIG_ROOT_POS: return e.root[(size_t)i * 13 + k];
    case LG_SIG_TRAJ0: return e.trajectory[(size_t)i * (c.traj.enabled ? c.traj.N : 1) * 2 + k];
    case LG_SIG_PREV_ERROR: return e.prev_error[(size_t)i * 2 + k];
    case LG_SIG_DOF_POS_REL: return x.q[k] - c.default_dof_pos[k];
    case LG_SIG_DOF_VEL: return x.qd[k];
    case LG_SIG_TORQUES: return x.tau[k];
    case LG_SIG_ACTIONS: return x.act[k];
    case LG_SIG_LAST_ACTIONS: return x.lact[k];
    }
    return 0.0f;
}
// generic extra terms (legged_hip.h lg_xterm_kind); tracking_rom = LT:1060-1069, differential_error = LT:1100-1110
static float xterm_value(const Env &e, int i, const lg_xterm &t, const RewardCtx &x) {
    float s = 0.0f;
    switch (t.kind) {
    case LG_XT_EXP_NEG_WSQ_ERR:
        for (int k = 0; k < t.n; ++k) {
            const float d = signal_at(e, i, t.sig_a, t.off_a + k, x) - signal_at(e, i, t.sig_b, t.off_b + k, x);
            s += d * d * t.w[k];
        }
        return std::exp(-s / t.p[0]);
    case LG_XT_WSQ:
        for (int k = 0; k < t.n; ++k) { const float a = signal_at(e, i, t.sig_a, t.off_a + k, x); s += t.w[k] * a * a; }
        return s;
    case LG_XT_SLOPED_ERR_CHANGE: {
        float pn = 0.0f;
        for (int k = 0; k < t.n; ++k) {
            const float d = signal_at(e, i, t.sig_a, t.off_a + k, x) - signal_at(e, i, t.sig_b, t.off_b + k, x);
            const float te = d * d, pc = signal_at(e, i, t.sig_c, t.off_c + k, x);
            s += te * te;
            pn += pc * pc;
        }
        const float diff = std::sqrt(s) - std::sqrt(pn);
        return (diff < 0.0f ? t.p[0] : t.p[1]) * diff;
    }
    }
    return 0.0f;
}

// LR:147-187 (+ :415-454, :463-486, AN:56-60) for one env; episode-sum logging is done by the caller
static void reset_env(Env &e, int i) {
    const lg_cfg &c = e.cfg;
    const int A = e.A;
    float *r = &e.root[(size_t)i * 13];
    float *org = &e.env_origins[(size_t)i * 3];
    float *cmd = &e.commands[(size_t)i * 4];
    if (c.curriculum && e.init_done) {                               // LR:463-486
        float dx = r[0] - org[0], dy = r[1] - org[1];
        float dist = std::sqrt(dx * dx + dy * dy);
        bool up = dist > c.terrain_env_length / 2.0f;
        float cn = std::sqrt(cmd[0] * cmd[0] + cmd[1] * cmd[1]);
        bool down = (dist < cn * c.episode_length_s * 0.5f) && !up;
        int64_t lvl = e.terrain_levels[i] + (up ? 1 : 0) - (down ? 1 : 0);
        if (lvl >= c.max_terrain_level) {
            if (e.inject) lvl = e.inj_levels[i];
            else lvl = std::min<int64_t>((int64_t)(uni(e, i, LG_SLOT_LEVEL) * c.max_terrain_level), c.max_terrain_level - 1);
        } else {
            lvl = std::max<int64_t>(lvl, 0);
        }
        e.terrain_levels[i] = lvl;
        const float *to = &e.terrain_origins[((size_t)lvl * c.terrain_num_cols + e.terrain_types[i]) * 3];
        org[0] = to[0]; org[1] = to[1]; org[2] = to[2];
    }
    const bool tj = c.traj.enabled;
    const int s_dof = tj ? LG_TSLOT_DOF : LG_SLOT_DOF, s_xy = tj ? LG_TSLOT_XY(A) : LG_SLOT_XY(A), s_vel = tj ? LG_TSLOT_VEL(A) : LG_SLOT_VEL(A);
    for (int j = 0; j < A; ++j) {                                    // LR:415-430
        float u = uni(e, i, s_dof + j);
        e.dof[((size_t)i * A + j) * 2] = c.default_dof_pos[j] * ((1.5f - 0.5f) * u + 0.5f);
        e.dof[((size_t)i * A + j) * 2 + 1] = 0.0f;
    }
    for (int k = 0; k < 13; ++k) r[k] = c.base_init_state[k];       // LR:432-454
    for (int k = 0; k < 3; ++k) r[k] += org[k];
    if (c.custom_origins)
        for (int k = 0; k < 2; ++k) r[k] += (1.0f - (-1.0f)) * uni(e, i, s_xy + k) + (-1.0f);
    for (int k = 0; k < 6; ++k) r[7 + k] = (0.5f - (-0.5f)) * uni(e, i, s_vel + k) + (-0.5f);
    if (tj) {                                                        // LT:186 reset_traj :222-229 instead of a command resample
        float z[2] = {r[0], r[1]};
        if (c.traj.randomize_rom_distance && uni(e, i, LG_TSLOT_ROMD(A)) > c.traj.zero_rom_dist_llh)
            for (int k = 0; k < 2; ++k)
                z[k] += (c.traj.max_rom_dist[k] - (-c.traj.max_rom_dist[k])) * uni(e, i, LG_TSLOT_ROMD(A) + 1 + k) + (-c.traj.max_rom_dist[k]);
        tg_reset(e, i, z);
        for (int k = 0; k < 2; ++k) {                                // LT:199: against the trajectory the callback computed (not refreshed)
            const float d = e.trajectory[(size_t)i * c.traj.N * 2 + k] - r[k];
            e.prev_error[(size_t)i * 2 + k] = d * d;
        }
    } else {
        resample_commands(e, i, LG_SLOT_RCMD(A), c.cmd_lo, c.cmd_hi);
    }
    for (int j = 0; j < A; ++j) { e.last_actions[(size_t)i * A + j] = 0.0f; e.last_dof_vel[(size_t)i * A + j] = 0.0f; }
    for (int f = 0; f < e.F; ++f) e.feet_air_time[(size_t)i * e.F + f] = 0.0f;
    e.ep_len[i] = 0;
    e.reset[i] = 1;
    if (c.use_actuator_net)                                          // AN:56-60
        for (int l = 0; l < 2; ++l)
            for (int j = 0; j < A; ++j)
                for (int k = 0; k < 8; ++k) {
                    size_t idx = ((size_t)l * e.N * A + (size_t)i * A + j) * 8 + k;
                    e.lstm_h[idx] = 0.0f;
                    e.lstm_c[idx] = 0.0f;
                }
}

// LR:488-505 / LT:519-553 as data (legged_hip.h lg_stage)
void stage_apply(Env &e, const lg_stage &s, int what) {
    lg_cfg &c = e.cfg;
    if (what & 1) {
        for (int k = 0; k < 4; ++k) { c.cmd_lo[k] = s.cmd_lo[k]; c.cmd_hi[k] = s.cmd_hi[k]; }
        c.max_push_vel = s.max_push_vel;
        for (int k = 0; k < LG_NUM_REWARDS; ++k) c.rew_scale[k] = s.rew_scale[k];
        for (int k = 0; k < LG_MAX_XTERMS; ++k) { c.xterms[k].scale = s.xterm_scale[k]; c.xterms[k].p[0] = s.xterm_p0[k]; }
        for (int k = 0; k < 2; ++k) { c.traj.v_min[k] = s.traj_v_min[k]; c.traj.v_max[k] = s.traj_v_max[k]; c.traj.max_rom_dist[k] = s.traj_max_rom_dist[k]; }
        c.traj.t_low = s.traj_t_low; c.traj.t_high = s.traj_t_high;
    }
    if (what & 2) {
        for (int k = 0; k < 4; ++k) { e.cb.cmd_lo[k] = s.cmd_lo[k]; e.cb.cmd_hi[k] = s.cmd_hi[k]; }
        e.cb.max_push_vel = s.max_push_vel;
        e.cb.traj = c.traj;
        for (int k = 0; k < 2; ++k) { e.cb.traj.v_min[k] = s.traj_v_min[k]; e.cb.traj.v_max[k] = s.traj_v_max[k]; }
        e.cb.traj.t_low = s.traj_t_low; e.cb.traj.t_high = s.traj_t_high;
        e.stage = s;
    }
}

// LR:106-137 followed by the obs clip of LR:100-103
void post_physics_step(Env &e) {
    const lg_cfg &c = e.cfg;
    const int N = e.N, A = e.A, B = e.B, O = e.O;
    e.step_counter += 1;                                              // LR:115
    // LR:358 on the push period of the stage in force when the step began (fractional after a curriculum multiplier)
    const double pt = e.stage.push_time;
    const bool push_now = c.push_robots && pt > 0.0 && std::fmod((double)e.step_counter, pt) == 0.0;
    if (e.has_pending) stage_apply(e, e.pending, 1);                  // LR:360-363 / LT:414-417: what follows the callback sees the new stage
    std::vector<uint8_t> was_reset(N, 0), was_fault(N, 0);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; ++i) {
        float *r = &e.root[(size_t)i * 13];
        float *cmd = &e.commands[(size_t)i * 4];
        const float *cf = &e.contact[(size_t)i * B * 3];
        e.ep_len[i] += 1;                                             // LR:114
        float *blv = &e.base_lin_vel[3 * i], *bav = &e.base_ang_vel[3 * i], *pg = &e.proj_grav[3 * i];
        const float gvec[3] = {0.0f, 0.0f, -1.0f};
        quat_rotate_inverse(r + 3, r + 7, blv);                       // LR:118-121
        quat_rotate_inverse(r + 3, r + 10, bav);
        quat_rotate_inverse(r + 3, gvec, pg);
        // ---- _post_physics_step_callback LR:343-363 / LT:405-417
        if (c.traj.enabled) tg_callback_step(e, i);
        else if (e.ep_len[i] % c.resample_steps == 0) resample_commands(e, i, LG_SLOT_CMD, e.cb.cmd_lo, e.cb.cmd_hi);
        if (c.heading_command && !c.traj.enabled) {
            const float fwd0[3] = {1.0f, 0.0f, 0.0f};
            float fwd[3];
            quat_apply(r + 3, fwd0, fwd);
            float heading = std::atan2(fwd[1], fwd[0]);
            float ang = cmd[3] - heading;                             // MA:45-48 wrap_to_pi
            const float two_pi = (float)(2.0 * M_PI);
            ang = std::fmod(ang, two_pi);
            if (ang < 0.0f) ang += two_pi;                            // python-style remainder
            if (ang > (float)M_PI) ang -= two_pi;
            cmd[2] = clampf(0.5f * ang, -1.0f, 1.0f);
        }
        if (c.measure_heights) get_heights(e, i);
        if (c.traj.enabled) {                                         // LT:150-160: per-env push timers (pushes are unconditional there)
            e.push_timer[i] -= c.dt;
            if (e.push_timer[i] <= 0.0f) {
                const float mv = c.traj.max_push_vel_xy;                  // LT:483-486
                r[7] = (mv - (-mv)) * uni(e, i, LG_TSLOT_PUSH) + (-mv);
                r[8] = (mv - (-mv)) * uni(e, i, LG_TSLOT_PUSH + 1) + (-mv);
                e.push_timer[i] = (c.traj.push_t_hi - c.traj.push_t_lo) * uni(e, i, LG_TSLOT_TIMER) + c.traj.push_t_lo;
            }
        } else if (push_now) {                                        // LR:456-461
            const float mv = e.cb.max_push_vel;
            r[7] = (mv - (-mv)) * uni(e, i, LG_SLOT_PUSH) + (-mv);
            r[8] = (mv - (-mv)) * uni(e, i, LG_SLOT_PUSH + 1) + (-mv);
        }
        // ---- check_termination LR:139-145
        bool rst = false;
        for (int b = 0; b < c.num_term; ++b) rst |= fnorm3(cf + 3 * c.term_idx[b]) > 1.0f;
        if (e.fault[i]) { rst = true; e.fault[i] = 0; was_fault[i] = 1; }   // physics fault guard (lgo_physics.cpp)
        bool to = e.ep_len[i] > c.max_episode_length;
        e.time_out[i] = to;
        e.reset[i] = rst || to;
        // ---- compute_reward LR:189-206
        RewardCtx x;
        x.blv = blv; x.bav = bav; x.pg = pg; x.cmd = cmd; x.cf = cf; x.root_z = r[2];
        float q[LG_MAX_DOF], qd[LG_MAX_DOF];
        for (int j = 0; j < A; ++j) { q[j] = e.dof[((size_t)i * A + j) * 2]; qd[j] = e.dof[((size_t)i * A + j) * 2 + 1]; }
        x.q = q; x.qd = qd; x.tau = &e.torques[(size_t)i * A]; x.act = &e.actions[(size_t)i * A];
        x.lact = &e.last_actions[(size_t)i * A]; x.lqd = &e.last_dof_vel[(size_t)i * A];
        float rew = 0.0f;
        for (int o = 0; o < c.num_terms; ++o) {                       // alphabetical over builtin and extra terms (LR:605-629)
            const int k = c.term_order[o];
            float v;
            if (k < LG_NUM_REWARDS) v = reward_term(e, i, k, x) * c.rew_scale[k];
            else v = xterm_value(e, i, c.xterms[k - LG_NUM_REWARDS], x) * c.xterms[k - LG_NUM_REWARDS].scale;
            rew += v;
            e.episode_sums[(size_t)k * N + i] += v;
        }
        if (c.only_positive_rewards) rew = std::max(rew, 0.0f);
        if (c.rew_scale[LG_REW_TERMINATION] != 0.0f) {
            float v = reward_term(e, i, LG_REW_TERMINATION, x) * c.rew_scale[LG_REW_TERMINATION];
            rew += v;
            e.episode_sums[(size_t)LG_REW_TERMINATION * N + i] += v;
        }
        e.rew[i] = rew;
        was_reset[i] = e.reset[i];
    }
    // ---- reset_idx LR:147-187: logging needs the pre-reset episode sums of all resetting envs
    int n_reset = 0;
    for (int i = 0; i < N; ++i) n_reset += was_reset[i];
    e.n_reset[0] = n_reset;
    int n_fault = 0;
    for (int i = 0; i < N; ++i) n_fault += was_fault[i];
    e.n_fault[0] = n_fault;
    e.fault_total[0] += n_fault;
    e.n_vel_clamp[0] = e.clamp_count; e.vel_clamp_total[0] += e.clamp_count; e.clamp_count = 0;
    if (n_reset > 0) {
        for (int k = 0; k < LG_NUM_TERMS; ++k) {
            const float sc = k < LG_NUM_REWARDS ? c.rew_scale[k] : (k - LG_NUM_REWARDS < c.num_xterms ? c.xterms[k - LG_NUM_REWARDS].scale : 0.0f);
            if (sc == 0.0f) { e.extras_episode[k] = 0.0f; continue; }
            float s = 0.0f;
            for (int i = 0; i < N; ++i)
                if (was_reset[i]) { s += e.episode_sums[(size_t)k * N + i]; e.episode_sums[(size_t)k * N + i] = 0.0f; }
            e.extras_episode[k] = (s / (float)n_reset) / c.episode_length_s;
        }
#pragma omp parallel for schedule(static)
        for (int i = 0; i < N; ++i)
            if (was_reset[i]) reset_env(e, i);
        if (c.curriculum) {
            float s = 0.0f;
            for (int i = 0; i < N; ++i) s += (float)e.terrain_levels[i];
            e.extras_terrain_level[0] = s / (float)N;
        }
        if (c.send_timeouts) std::memcpy(e.extras_time_outs.data(), e.time_out.data(), N);
        if (c.traj.enabled)                                           // the reset loop's get_input_t reaches every env (lgo_traj.cpp)
            for (int i = 0; i < N; ++i)
                if (!was_reset[i]) tg_late_resample(e, i);
    }
    for (int k = 0; k < LG_NUM_TERMS; ++k) e.extras_episode_acc[k] += e.extras_episode[k];   // rsl_rl log(): mean over the steps
    e.extras_episode_acc[LG_NUM_TERMS] += e.extras_terrain_level[0];
    e.extras_episode_acc[LG_NUM_TERMS + 1] += 1.0f;
    // ---- compute_observations LR:208-226, clip LR:100-103, bookkeeping LR:132-134
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; ++i) {
        float *o = &e.obs[(size_t)i * O];
        const float *r = &e.root[(size_t)i * 13];
        const float *cmd = &e.commands[(size_t)i * 4];
        for (int k = 0; k < 3; ++k) {
            o[k] = e.base_lin_vel[3 * i + k] * c.obs_scale_lin_vel;
            o[3 + k] = e.base_ang_vel[3 * i + k] * c.obs_scale_ang_vel;
            o[6 + k] = e.proj_grav[3 * i + k];
        }
        int ob = 12;                                                  // first joint entry
        if (c.traj.enabled) {                                         // LT:280-288: trajectory relative to the robot, scaled per dim
            ob = 9 + 2 * c.traj.N;
            for (int p = 0; p < c.traj.N; ++p)
                for (int d = 0; d < 2; ++d)
                    o[9 + 2 * p + d] = (e.trajectory[((size_t)i * c.traj.N + p) * 2 + d] - r[d]) * c.traj.obs_scale[d];
        } else {
            o[9] = cmd[0] * c.obs_scale_lin_vel;
            o[10] = cmd[1] * c.obs_scale_lin_vel;
            o[11] = cmd[2] * c.obs_scale_ang_vel;
        }
        for (int j = 0; j < A; ++j) {
            float q = e.dof[((size_t)i * A + j) * 2], qd = e.dof[((size_t)i * A + j) * 2 + 1];
            o[ob + j] = (q - c.default_dof_pos[j]) * c.obs_scale_dof_pos;
            o[ob + A + j] = qd * c.obs_scale_dof_vel;
            o[ob + 2 * A + j] = e.actions[(size_t)i * A + j];
        }
        if (c.measure_heights)
            for (int h = 0; h < e.H; ++h)
                o[ob + 3 * A + h] = clampf(r[2] - 0.5f - e.heights[(size_t)i * e.H + h], -1.0f, 1.0f) * c.obs_scale_height;
        for (int k = 0; k < O; ++k) {
            float v = o[k];
            if (c.add_noise) v += (2.0f * uni(e, i, (c.traj.enabled ? LG_TSLOT_NOISE(A) : LG_SLOT_NOISE(A)) + k) - 1.0f) * e.noise_vec[k];
            o[k] = clampf(v, -c.clip_obs, c.clip_obs);
        }
        for (int j = 0; j < A; ++j) {
            e.last_actions[(size_t)i * A + j] = e.actions[(size_t)i * A + j];
            e.last_dof_vel[(size_t)i * A + j] = e.dof[((size_t)i * A + j) * 2 + 1];
        }
        for (int k = 0; k < 6; ++k) e.last_root_vel[(size_t)i * 6 + k] = r[7 + k];
    }
    if (e.has_pending) { stage_apply(e, e.pending, 2); e.has_pending = 0; }   // from the next step on the callback sees it too
}

// base_task.py:113 reset_idx(arange(N)) -- no extras bookkeeping needed by callers
void reset_all(Env &e) {
    for (int i = 0; i < e.N; ++i) {
        for (int k = 0; k < LG_NUM_TERMS; ++k) e.episode_sums[(size_t)k * e.N + i] = 0.0f;
        reset_env(e, i);
    }
}

// LR:147-187 called with a caller-given id list (what play / data-collection style callers do)
void reset_ids(Env &e, const int32_t *ids, int n) {
    const lg_cfg &c = e.cfg;
    if (n == 0) return;                                               // LR:156-157
    for (int k = 0; k < LG_NUM_TERMS; ++k) {
        const float sc = k < LG_NUM_REWARDS ? c.rew_scale[k] : (k - LG_NUM_REWARDS < c.num_xterms ? c.xterms[k - LG_NUM_REWARDS].scale : 0.0f);
        if (sc == 0.0f) { e.extras_episode[k] = 0.0f; continue; }
        float s = 0.0f;
        for (int q = 0; q < n; ++q) { s += e.episode_sums[(size_t)k * e.N + ids[q]]; e.episode_sums[(size_t)k * e.N + ids[q]] = 0.0f; }
        e.extras_episode[k] = (s / (float)n) / c.episode_length_s;
    }
    for (int q = 0; q < n; ++q) reset_env(e, ids[q]);
    if (c.traj.enabled) {                                             // the generator's reset loop reaches every env (lgo_traj.cpp)
        std::vector<uint8_t> in(e.N, 0);
        for (int q = 0; q < n; ++q) in[ids[q]] = 1;
        for (int i = 0; i < e.N; ++i)
            if (!in[i]) tg_late_resample(e, i);
    }
    if (c.curriculum) {
        float s = 0.0f;
        for (int i = 0; i < e.N; ++i) s += (float)e.terrain_levels[i];
        e.extras_terrain_level[0] = s / (float)e.N;
    }
    if (c.send_timeouts) std::memcpy(e.extras_time_outs.data(), e.time_out.data(), e.N);
    e.n_reset[0] = n;
    e.n_fault[0] = 0;
}

// ---------------------------------------------------------------- torque laws
static inline float sigm(float x) { return 1.0f / (1.0f + std::exp(-x)); }

// AN:71-81 + the archive's forward (x*in_scale -> LSTM(2->8->8) -> Linear -> *out_scale);
// torch gate order i,f,g,o.  One (env, joint) row.
static float lstm_row(const float *w, float x0, float x1, float *h, float *cst, size_t stride_layer) {
    const float *in_scale = w, *out_scale = w + 2;
    const float *wih0 = w + 3, *whh0 = wih0 + 64, *bih0 = whh0 + 256, *bhh0 = bih0 + 32;
    const float *wih1 = bhh0 + 32, *whh1 = wih1 + 256, *bih1 = whh1 + 256, *bhh1 = bih1 + 32;
    const float *lw = bhh1 + 32, *lb = lw + 8;
    float in0[2] = {x0 * in_scale[0], x1 * in_scale[1]};
    float hin[8];
    for (int layer = 0; layer < 2; ++layer) {
        float *hl = h + layer * stride_layer, *cl = cst + layer * stride_layer;
        const float *wih = layer ? wih1 : wih0, *whh = layer ? whh1 : whh0;
        const float *bih = layer ? bih1 : bih0, *bhh = layer ? bhh1 : bhh0;
        const int nin = layer ? 8 : 2;
        const float *xin = layer ? hin : in0;
        float g[32];
        for (int r = 0; r < 32; ++r) {
            float s = bih[r] + bhh[r];
            for (int k = 0; k < nin; ++k) s += wih[r * nin + k] * xin[k];
            for (int k = 0; k < 8; ++k) s += whh[r * 8 + k] * hl[k];
            g[r] = s;
        }
        for (int k = 0; k < 8; ++k) {
            float ig = sigm(g[k]), fg = sigm(g[8 + k]), gg = std::tanh(g[16 + k]), og = sigm(g[24 + k]);
            float cn = fg * cl[k] + ig * gg;
            cl[k] = cn;
            hin[k] = og * std::tanh(cn);
        }
        for (int k = 0; k < 8; ++k) hl[k] = hin[k];
    }
    float y = lb[0];
    for (int k = 0; k < 8; ++k) y += lw[k] * hin[k];
    return out_scale[0] * y;
}

// LR:389-413 (P/V/T + clip) or AN:71-81 (actuator net, no clip)
void compute_torques(Env &e) {
    const lg_cfg &c = e.cfg;
    const int N = e.N, A = e.A;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; ++i) {
        for (int j = 0; j < A; ++j) {
            size_t ij = (size_t)i * A + j;
            float q = e.dof[ij * 2], qd = e.dof[ij * 2 + 1];
            float as = e.actions[ij] * c.action_scale;
            float tau;
            if (c.use_actuator_net) {
                tau = lstm_row(c.lstm_w, as + c.default_dof_pos[j] - q, qd, &e.lstm_h[ij * 8], &e.lstm_c[ij * 8],
                               (size_t)N * A * 8);
            } else {
                if (c.control_type == 0) tau = c.p_gains[j] * (as + c.default_dof_pos[j] - q) - c.d_gains[j] * qd;
                else if (c.control_type == 1)
                    tau = c.p_gains[j] * (as - qd) - c.d_gains[j] * (qd - e.last_dof_vel[ij]) / c.sim_dt;
                else tau = as;
                tau = clampf(tau, -c.torque_limits[j], c.torque_limits[j]);
            }
            e.torques[ij] = tau;
        }
    }
}

}  // namespace lgo
