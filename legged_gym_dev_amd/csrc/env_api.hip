// C-ABI (include/legged_hip.h) for the environment step: context, HBM allocation, launches.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>

#include "lg_device.h"

static thread_local std::string g_err;
void lg_set_error(const std::string &s) { g_err = s; }

extern "C" void lgk_set_actions(const DevParams *P, const float *a, int n, hipStream_t s);
extern "C" int lgk_substeps(const DevParams *P, const float *a_in, int N, int L, int J, int lstm, int mode, int iters, hipStream_t s);
extern "C" void lgk_post_step(const DevParams *P, int N, int64_t counter, int inject, int init_done, int traj, int push_now, int finalize,
                              hipStream_t s);
extern "C" void lgk_finalize(const DevParams *P, int accumulate, hipStream_t s);
extern "C" void lgk_set_stage(DevParams *P, const lg_stage *st, int what, hipStream_t s);
extern "C" void lgk_reset_all(const DevParams *P, int N, int64_t counter, int inject, int init_done, hipStream_t s);
extern "C" void lgk_reset_ids(const DevParams *P, const int32_t *ids, int n, int N, int64_t counter, int inject, int init_done, int traj,
                              hipStream_t s);

#define HIPCHK(x)                                                                                  \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) {                                                                    \
            g_err = std::string(#x) + ": " + hipGetErrorString(e_);                                \
            return -100;                                                                           \
        }                                                                                          \
    } while (0)

template <typename T>
static int dalloc(lg_ctx *c, T **p, size_t n, bool zero = true) {
    void *q = nullptr;
    size_t bytes = (n ? n : 1) * sizeof(T);
    if (hipMalloc(&q, bytes) != hipSuccess) { g_err = "hipMalloc failed"; return -100; }
    if (zero && hipMemset(q, 0, bytes) != hipSuccess) { g_err = "hipMemset failed"; return -100; }
    if (c->n_allocs >= 128) { g_err = "alloc table full"; return -101; }
    c->allocs[c->n_allocs++] = q;
    *p = (T *)q;
    return 0;
}
#define DA(ptr, n) do { int rc_ = dalloc(c, &(ptr), (n)); if (rc_) { lg_destroy(c); return rc_; } } while (0)

extern "C" {

const char *lg_last_error(void) { return g_err.c_str(); }
int lg_version(void) { return 1; }

int lg_destroy(lg_ctx *c) {
    if (!c) return 0;
    // a learner attached with lg_ppo_attach_env holds this pointer (and a pending epilogue): detach it first
    if (c->defer_finalize) { g_err = "lg_destroy: a learner is still attached (lg_ppo_attach_env(p, NULL) or lg_ppo_destroy first)"; return -20; }
    if (c->d) (void)hipStreamSynchronize(c->stream);
    for (int i = 0; i < c->n_allocs; ++i) (void)hipFree(c->allocs[i]);
    delete c;
    return 0;
}

int lg_create(const lg_cfg *cfg, const lg_model *model, const int16_t *height_samples, lg_ctx **out) {
    if (!cfg || !model || !out) { g_err = "null argument"; return -1; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        g_err = "no HIP device: liblegged_hip has no CPU fallback";
        return -2;
    }
    const int N = cfg->num_envs, A = cfg->num_actions, B = cfg->num_bodies, O = cfg->num_obs;
    const int F = cfg->num_feet, H = cfg->num_height_points;
    const int L = model->num_legs, J = model->joints_per_leg;
    if (N <= 0 || A != L * J || A > LG_MAX_DOF || B > LG_MAX_BODIES || F > LG_MAX_FEET || L > 8) {
        g_err = "unsupported sizes (num_envs/num_actions/bodies/feet/legs)";
        return -3;
    }
    if (!((L == 4 && J == 3) || (L == 2 && J == 6))) {
        g_err = "no physics kernel instantiated for this topology (have 4x3 and 2x6)";
        return -4;
    }
    if (cfg->use_actuator_net && !(L == 4 && J == 3)) {
        g_err = "the actuator-net control loop is instantiated for the 4x3 topology only (have 4x3 PD, 4x3 actuator net, 2x6 PD)";
        return -4;
    }
    const int ncmd = cfg->traj.enabled ? 2 * cfg->traj.N : 3;
    if (O != 9 + ncmd + 3 * A + (cfg->measure_heights ? H : 0)) { g_err = "num_obs inconsistent with the observation layout"; return -5; }
    if (cfg->traj.enabled && (cfg->traj.N < 1 || cfg->traj.dN != 1 || cfg->traj.N * cfg->traj.dN + 1 > LG_TRAJ_MAX_PTS)) {
        g_err = "trajectory window outside 1..LG_TRAJ_MAX_PTS-1 points (dN must be 1)"; return -5;
    }
    if (cfg->num_xterms < 0 || cfg->num_xterms > LG_MAX_XTERMS || cfg->num_terms < 0 || cfg->num_terms > LG_NUM_TERMS) {
        g_err = "bad reward term table"; return -5;
    }
    for (int k = 0; k < cfg->num_terms; ++k)
        if (cfg->term_order[k] < 0 || cfg->term_order[k] >= LG_NUM_REWARDS + cfg->num_xterms) { g_err = "term_order entry out of range"; return -5; }
    for (int k = 0; k < cfg->num_xterms; ++k)
        if (cfg->xterms[k].n < 0 || cfg->xterms[k].n > 8 || cfg->xterms[k].kind < 0 || cfg->xterms[k].kind > LG_XT_SLOPED_ERR_CHANGE) {
            g_err = "bad extra reward term"; return -5;
        }
    if (cfg->terrain_type == 1 && !height_samples) { g_err = "terrain_type=1 needs height samples"; return -6; }
    if (!cfg->noise_vec) { g_err = "cfg.noise_vec is null"; return -7; }

    lg_ctx *c = new lg_ctx();
    memset(c, 0, sizeof(*c));
    c->init_done = 1;
    DevParams &h = c->h;
    h.cfg = *cfg;
    for (int o = 0; o < LG_LSTM_LDS; ++o) h.lstm_img[o] = lstm_lds_image(cfg->lstm_w, o);
    h.model = *model;
    h.cfg.noise_vec = h.cfg.height_points = h.cfg.terrain_origins = nullptr;
    h.K = (cfg->traj.enabled ? LG_TSLOT_NOISE(A) : LG_SLOT_NOISE(A)) + O;
    {   // the stage in force = what the cfg holds
        lg_stage &s = c->stage;
        memset(&s, 0, sizeof(s));
        memcpy(s.cmd_lo, cfg->cmd_lo, sizeof(s.cmd_lo)); memcpy(s.cmd_hi, cfg->cmd_hi, sizeof(s.cmd_hi));
        s.max_push_vel = cfg->max_push_vel;
        s.push_time = (double)cfg->push_interval;
        memcpy(s.rew_scale, cfg->rew_scale, sizeof(s.rew_scale));
        for (int k = 0; k < LG_MAX_XTERMS; ++k) { s.xterm_scale[k] = cfg->xterms[k].scale; s.xterm_p0[k] = cfg->xterms[k].p[0]; }
        memcpy(s.traj_v_min, cfg->traj.v_min, 8); memcpy(s.traj_v_max, cfg->traj.v_max, 8);
        s.traj_t_low = cfg->traj.t_low; s.traj_t_high = cfg->traj.t_high;
        memcpy(s.traj_max_rom_dist, cfg->traj.max_rom_dist, 8);
        StageCb &b = h.cb;
        memcpy(b.cmd_lo, s.cmd_lo, sizeof(b.cmd_lo)); memcpy(b.cmd_hi, s.cmd_hi, sizeof(b.cmd_hi));
        b.max_push_vel = s.max_push_vel;
        memcpy(b.v_min, s.traj_v_min, 8); memcpy(b.v_max, s.traj_v_max, 8);
        b.t_low = s.traj_t_low; b.t_high = s.traj_t_high;
    }

    // sphere slots per leg (same link pattern on every leg) + base spheres one per lane
    {
        int cnt[8] = {0};
        h.n_leg_slots = 0;
        h.n_base_spheres = 0;
        for (int k = 0; k < model->num_spheres; ++k) {
            int l = model->sph_link[k];
            if (l < 0) {
                int b = h.n_base_spheres++;
                if (b >= L * LG_MAX_BASE_PER_LANE) { g_err = "too many collision spheres on the base"; delete c; return -8; }
                if (b > 0 && model->sph_body[k] != h.base_body[0]) { g_err = "base collision spheres report to different bodies"; delete c; return -8; }
                h.base_body[b] = model->sph_body[k];
                memcpy(h.base_center[b], model->sph_center[k], 12);
                h.base_radius[b] = model->sph_radius[k];
            } else {
                int leg = l / J, s = cnt[leg]++;
                if (s >= LG_MAX_LEG_SLOTS) { g_err = "too many collision spheres on one leg"; delete c; return -9; }
                if (leg == 0) h.slot_link[s] = l % J;
                else if (h.slot_link[s] != l % J) { g_err = "legs differ in their sphere->link pattern"; delete c; return -10; }
                h.slot_body[s][leg] = model->sph_body[k];
                memcpy(h.slot_center[s][leg], model->sph_center[k], 12);
                h.slot_radius[s][leg] = model->sph_radius[k];
            }
        }
        h.n_leg_slots = cnt[0];
        h.slot_link_pk = 0ull;
        for (int s = 0; s < h.n_leg_slots; ++s) h.slot_link_pk |= (unsigned long long)(h.slot_link[s] & 15) << (4 * s);
        for (int l = 1; l < L; ++l)
            if (cnt[l] != cnt[0]) { g_err = "legs differ in their number of collision spheres"; delete c; return -11; }
        if (J > LG_LT_MAXJ || L > 8) { g_err = "kinematic tree beyond the per-leg table"; delete c; return -12; }
        for (int l = 0; l < L; ++l) {
            float *t = h.leg_tab[l];
            for (int j = 0; j < J; ++j) {
                const int d = l * J + j;
                float *q = t + LG_LT_JOINT * j;
                memcpy(q, model->R_pj[d], 36); memcpy(q + 9, model->p_pj[d], 12); memcpy(q + 12, model->axis[d], 12);
                memcpy(q + 15, model->inertia[d + 1], 36); memcpy(q + 24, model->com[d + 1], 12);
                q[27] = model->mass[d + 1]; q[28] = model->joint_damping[d]; q[29] = model->vel_limit[d];
                q[30] = model->q_lower[d]; q[31] = model->q_upper[d];
            }
            for (int s = 0; s < h.n_leg_slots; ++s) {
                memcpy(t + LG_LT_SLOTS + 4 * s, h.slot_center[s][l], 12);
                t[LG_LT_SLOTS + 4 * s + 3] = h.slot_radius[s][l];
            }
            for (int u = 0; u < LG_MAX_BASE_PER_LANE; ++u) {
                const int b = l + u * L;
                if (b < h.n_base_spheres) {
                    memcpy(t + LG_LT_BASE + 4 * u, h.base_center[b], 12);
                    t[LG_LT_BASE + 4 * u + 3] = h.base_radius[b];
                }
            }
        }
    }

    lg_buffers &b = h.buf;
    DA(b.root_states, (size_t)N * 13); DA(b.dof_state, (size_t)N * A * 2); DA(b.contact_forces, (size_t)N * B * 3);
    DA(b.torques, (size_t)N * A); DA(b.actions, (size_t)N * A); DA(b.obs, (size_t)N * O); DA(b.rew, N);
    DA(b.reset, N); DA(b.time_out, N); DA(b.episode_length, N);
    DA(b.commands, (size_t)N * 4); DA(b.last_actions, (size_t)N * A); DA(b.last_dof_vel, (size_t)N * A);
    DA(b.last_root_vel, (size_t)N * 6); DA(b.feet_air_time, (size_t)N * F); DA(b.last_contacts, (size_t)N * F);
    DA(b.episode_sums, (size_t)LG_NUM_TERMS * N); DA(b.base_lin_vel, (size_t)N * 3); DA(b.base_ang_vel, (size_t)N * 3);
    DA(b.projected_gravity, (size_t)N * 3); DA(b.measured_heights, (size_t)N * (H ? H : 1));
    DA(b.env_origins, (size_t)N * 3); DA(b.terrain_levels, N); DA(b.terrain_types, N);
    DA(b.lstm_h, (size_t)2 * N * A * 8); DA(b.lstm_c, (size_t)2 * N * A * 8);
    DA(b.friction, N); DA(b.base_mass_delta, N);
    DA(b.extras_episode, LG_NUM_TERMS); DA(b.extras_terrain_level, 1); DA(b.extras_time_outs, N); DA(b.n_reset, 1);
    DA(b.n_fault, 1); DA(b.fault_total, 1); DA(h.fault_count, 1); DA(b.n_vel_clamp, 1); DA(b.vel_clamp_total, 1); DA(h.clamp_count, 1); DA(b.extras_episode_acc, LG_NUM_TERMS + 2);
    {
        const int npts = cfg->traj.enabled ? cfg->traj.N * cfg->traj.dN + 1 : 1, nobs = cfg->traj.enabled ? cfg->traj.N : 1;
        DA(b.tg_state, (size_t)N * LG_TG_STRIDE); DA(b.tg_traj, (size_t)N * npts * 2); DA(b.trajectory, (size_t)N * nobs * 2);
        DA(b.prev_error, (size_t)N * 2); DA(b.push_timer, N); DA(h.reset_mark, N); DA(h.dbg_cycles, 64 * 8);
    }
    DA(b.inject_uniforms, (size_t)N * h.K); DA(b.inject_levels, N); DA(b.material, (size_t)N * 4);
    DA(h.ep_accum, LG_NUM_TERMS); DA(h.reset_count, 1); DA(h.fault, N); DA(h.any_reset_step, 1);
    {
        const int64_t never = -1;
        (void)hipMemcpy(h.any_reset_step, &never, sizeof(never), hipMemcpyHostToDevice);
    }
    {   // defaults: identity quaternion, unit friction, reset flags = 1 (base_task.py:72)
        float *tmp = new float[(size_t)N * 13]();
        for (int i = 0; i < N; ++i) tmp[(size_t)i * 13 + 6] = 1.0f;
        (void)hipMemcpy(b.root_states, tmp, sizeof(float) * N * 13, hipMemcpyHostToDevice);
        for (int i = 0; i < N; ++i) tmp[i] = 1.0f;
        (void)hipMemcpy(b.friction, tmp, sizeof(float) * N, hipMemcpyHostToDevice);
        delete[] tmp;
        (void)hipMemset(b.reset, 1, N);
    }
    float *nv = nullptr, *hp = nullptr, *to = nullptr;
    int16_t *hs = nullptr;
    DA(nv, O);
    HIPCHK(hipMemcpy(nv, cfg->noise_vec, sizeof(float) * O, hipMemcpyHostToDevice));
    h.noise_vec = nv;
    if (H) {
        if (!cfg->height_points) { g_err = "cfg.height_points is null"; lg_destroy(c); return -12; }
        DA(hp, (size_t)H * 2);
        HIPCHK(hipMemcpy(hp, cfg->height_points, sizeof(float) * H * 2, hipMemcpyHostToDevice));
    }
    h.height_points = hp;
    if (cfg->terrain_origins && cfg->max_terrain_level > 0) {
        size_t n = (size_t)cfg->max_terrain_level * cfg->terrain_num_cols * 3;
        DA(to, n);
        HIPCHK(hipMemcpy(to, cfg->terrain_origins, sizeof(float) * n, hipMemcpyHostToDevice));
    }
    h.terrain_origins = to;
    if (cfg->terrain_type == 1) {
        size_t n = (size_t)cfg->hf_rows * cfg->hf_cols;
        DA(hs, n);
        HIPCHK(hipMemcpy(hs, height_samples, sizeof(int16_t) * n, hipMemcpyHostToDevice));
    }
    h.height_samples = hs;
    DA(c->d, 1);
    HIPCHK(hipMemcpy(c->d, &h, sizeof(DevParams), hipMemcpyHostToDevice));
    HIPCHK(hipDeviceSynchronize());
    *out = c;
    return 0;
}

int lg_get_buffers(lg_ctx *c, lg_buffers *out) { if (!c || !out) return -1; *out = c->h.buf; return 0; }
int lg_set_stream(lg_ctx *c, void *stream) { c->stream = (hipStream_t)stream; return 0; }
int lg_set_step_counter(lg_ctx *c, int64_t v) { c->step_counter = v; return 0; }
int64_t lg_get_step_counter(lg_ctx *c) { return c->step_counter; }
int lg_set_init_done(lg_ctx *c, int v) { c->init_done = v; return 0; }
int lg_inject_uniforms(lg_ctx *c, int enable) { c->inject = enable; return 0; }

static int chk_launch() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_err = std::string("kernel launch: ") + hipGetErrorString(e); return -100; }
    return 0;
}

// A deferred epilogue (lg_ctx.defer_finalize) that nobody took over runs before anything else touches the env.
static void flush_finalize(lg_ctx *c) {
    if (c->finalize_pending) { lgk_finalize(c->d, 1, c->stream); c->finalize_pending = 0; }
}
int lg_finalize(lg_ctx *c) { if (!c) return -1; flush_finalize(c); return chk_launch(); }
// ---- internal (same library, not in the header): the learner's fused rollout epilogue (ppo_api.hip, lg_ppo_attach_env)
void lg_internal_defer_finalize(lg_ctx *c, int on) { if (!on) flush_finalize(c); c->defer_finalize = on; }
int lg_internal_finalize_pending(lg_ctx *c) { return c->finalize_pending; }
// the caller's launch runs the epilogue of the last step: hand over what it needs and clear the flag
const DevParams *lg_internal_take_finalize(lg_ctx *c, int64_t *counter) { c->finalize_pending = 0; *counter = c->step_counter; return c->d; }
const DevParams *lg_internal_host_params(lg_ctx *c) { return &c->h; }
hipStream_t lg_internal_stream(lg_ctx *c) { return c->stream; }

int lg_set_actions(lg_ctx *c, const float *actions) {
    lgk_set_actions(c->d, actions, c->h.cfg.num_envs * c->h.cfg.num_actions, c->stream);
    return chk_launch();
}
static int run_substeps(lg_ctx *c, const float *actions, int mode, int iters) {
    const lg_cfg &f = c->h.cfg;
    if (lgk_substeps(c->d, actions, f.num_envs, c->h.model.num_legs, c->h.model.joints_per_leg, f.use_actuator_net, mode, iters, c->stream)) {
        g_err = "no control-loop kernel for this topology / actuator combination (have 4x3 PD, 4x3 actuator net, 2x6 PD)";
        return -4;
    }
    return chk_launch();
}
int lg_compute_torques(lg_ctx *c) { return run_substeps(c, c->h.buf.actions, 1 /*torque stage*/, 1); }
int lg_simulate(lg_ctx *c) { return run_substeps(c, c->h.buf.actions, 2 /*physics stage*/, 1); }
// a curriculum stage into the host copy of the constants (the device copy follows through k_set_stage)
static void stage_to_host(lg_ctx *c, const lg_stage &s) {
    lg_cfg &f = c->h.cfg;
    memcpy(f.cmd_lo, s.cmd_lo, sizeof(f.cmd_lo)); memcpy(f.cmd_hi, s.cmd_hi, sizeof(f.cmd_hi));
    f.max_push_vel = s.max_push_vel;
    memcpy(f.rew_scale, s.rew_scale, sizeof(f.rew_scale));
    for (int k = 0; k < LG_MAX_XTERMS; ++k) { f.xterms[k].scale = s.xterm_scale[k]; f.xterms[k].p[0] = s.xterm_p0[k]; }
    memcpy(f.traj.v_min, s.traj_v_min, 8); memcpy(f.traj.v_max, s.traj_v_max, 8); memcpy(f.traj.max_rom_dist, s.traj_max_rom_dist, 8);
    f.traj.t_low = s.traj_t_low; f.traj.t_high = s.traj_t_high;
    c->stage = s;
}
int lg_get_stage(lg_ctx *c, lg_stage *out) { if (!c || !out) return -1; *out = c->has_pending ? c->pending : c->stage; return 0; }
int lg_set_curriculum_stage(lg_ctx *c, const lg_stage *s, int in_callback) {
    if (!c || !s) { g_err = "null argument"; return -1; }
    if (!(s->push_time >= 0.0)) { g_err = "lg_stage.push_time must be >= 0"; return -1; }
    if (in_callback) { c->pending = *s; c->has_pending = 1; return 0; }   // applied by the next lg_post_physics_step
    c->has_pending = 0;
    stage_to_host(c, *s);
    lgk_set_stage(c->d, s, 3, c->stream);
    return chk_launch();
}
int lg_post_physics_step(lg_ctx *c) {
    flush_finalize(c);
    c->step_counter += 1;                                       // legged_robot.py:115
    // legged_robot.py:358: common_step_counter % push_time == 0 on the period of the stage in force when the step began
    // (push_time may be fractional after a curriculum multiplier: Python's float modulo)
    const double pt = c->stage.push_time;
    const int push_now = c->h.cfg.push_robots && pt > 0.0 && std::fmod((double)c->step_counter, pt) == 0.0;
    if (c->has_pending) lgk_set_stage(c->d, &c->pending, 1, c->stream);          // what follows the callback sees the new stage
    lgk_post_step(c->d, c->h.cfg.num_envs, c->step_counter, c->inject, c->init_done, c->h.cfg.traj.enabled, push_now, !c->defer_finalize,
                  c->stream);
    c->finalize_pending = c->defer_finalize;
    if (c->has_pending) {                                                        // ... and from the next step on the callback too
        lgk_set_stage(c->d, &c->pending, 2, c->stream);
        stage_to_host(c, c->pending);
        c->has_pending = 0;
    }
    return chk_launch();
}
int lg_reset_all(lg_ctx *c) {
    flush_finalize(c);
    lgk_reset_all(c->d, c->h.cfg.num_envs, c->step_counter, c->inject, c->init_done, c->stream);
    return chk_launch();
}
int lg_debug_post_step_cycles(lg_ctx *c, unsigned long long *out /* host, 64 x 8 */) {
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(out, c->h.dbg_cycles, 64 * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return 0;
}
int lg_reset_ids(lg_ctx *c, const int32_t *ids, int n) {        // legged_robot.py:147-187
    if (n < 0 || (n > 0 && !ids)) { g_err = "lg_reset_ids: bad id list"; return -1; }
    if (n == 0) return 0;                                       // :156-157
    flush_finalize(c);
    lgk_reset_ids(c->d, ids, n, c->h.cfg.num_envs, c->step_counter, c->inject, c->init_done, c->h.cfg.traj.enabled, c->stream);
    return chk_launch();
}
// the control loop alone (clip + decimation x {torques, physics}), without the post-step: timing / profiling entry
int lg_debug_control_loop(lg_ctx *c, const float *actions) { return run_substeps(c, actions, 3, c->h.cfg.decimation); }
static int g_fused_substeps = 1;
int lg_debug_set_fused(int v) { g_fused_substeps = v; return 0; }
extern "C" void lgk_debug_set_substeps_nw(int v);
extern "C" void lgk_debug_set_substeps_occ(int v);
int lg_debug_set_substeps_occ(int v) { lgk_debug_set_substeps_occ(v); return 0; }     // control-loop register budget: 0 by grid size, 1 / 2 forced
int lg_debug_set_substeps_nw(int v) { lgk_debug_set_substeps_nw(v); return 0; }   // control-loop block shape: 4 waves / 64/L envs, or 2 / 32/L
extern "C" void lgk_debug_set_phys_pair(int v);
int lg_debug_set_phys_pair(int v) { lgk_debug_set_phys_pair(v); return 0; }   // physics lane map: 1 pair-lane (default), 0 one lane per leg

int lg_step(lg_ctx *c, const float *actions) {                  // legged_robot.py:80-104
    int rc;
    flush_finalize(c);
    if (g_fused_substeps) {                                     // one launch for clip + decimation x {torques, physics}
        rc = run_substeps(c, actions, 3, c->h.cfg.decimation);
    } else {                                                    // launch per substep (A/B and debugging)
        rc = lg_set_actions(c, actions);
        for (int d = 0; d < c->h.cfg.decimation && !rc; ++d) {
            rc = lg_compute_torques(c);
            if (!rc) rc = lg_simulate(c);
        }
    }
    if (!rc) rc = lg_post_physics_step(c);
    return rc;
}

}  // extern "C"
