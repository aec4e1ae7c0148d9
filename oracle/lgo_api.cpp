// CPU ORACLE -- TEST INFRASTRUCTURE ONLY (see lgo_common.h).
// extern "C" surface mirroring include/legged_hip.h with host pointers (lgo_* instead of lg_*).
#include "lgo_common.h"

using namespace lgo;

static thread_local std::string g_err;

extern "C" {

const char *lgo_last_error(void) { return g_err.c_str(); }

int lgo_create(const lg_cfg *cfg, const lg_model *model, const int16_t *height_samples, void **out) {
    if (!cfg || !model || !out) { g_err = "null argument"; return -1; }
    Env *e = new Env();
    e->cfg = *cfg;
    e->model = *model;
    const int N = e->N = cfg->num_envs, A = e->A = cfg->num_actions, B = e->B = cfg->num_bodies;
    const int O = e->O = cfg->num_obs, F = e->F = cfg->num_feet, H = e->H = cfg->num_height_points;
    e->K = (cfg->traj.enabled ? LG_TSLOT_NOISE(A) : LG_SLOT_NOISE(A)) + O;
    if (A > LG_MAX_DOF || B > LG_MAX_BODIES || F > LG_MAX_FEET) { g_err = "model exceeds LG_MAX_*"; delete e; return -2; }
    const int ocmd = cfg->traj.enabled ? 2 * cfg->traj.N : 3;
    if (O != 9 + ocmd + 3 * A + (cfg->measure_heights ? H : 0)) { g_err = "num_obs inconsistent with layout"; delete e; return -3; }
    e->noise_vec.assign(cfg->noise_vec, cfg->noise_vec + O);
    if (H) e->height_points.assign(cfg->height_points, cfg->height_points + 2 * H);
    if (cfg->terrain_type == 1) {
        if (!height_samples) { g_err = "terrain_type=1 needs height samples"; delete e; return -4; }
        e->height_samples.assign(height_samples, height_samples + (size_t)cfg->hf_rows * cfg->hf_cols);
    }
    if (cfg->terrain_origins && cfg->max_terrain_level > 0)
        e->terrain_origins.assign(cfg->terrain_origins,
                                  cfg->terrain_origins + (size_t)cfg->max_terrain_level * cfg->terrain_num_cols * 3);
    e->cfg.noise_vec = e->cfg.height_points = e->cfg.terrain_origins = nullptr;
    {   // the stage in force = what the cfg holds
        lg_stage s;
        std::memset(&s, 0, sizeof(s));
        std::memcpy(s.cmd_lo, cfg->cmd_lo, sizeof(s.cmd_lo)); std::memcpy(s.cmd_hi, cfg->cmd_hi, sizeof(s.cmd_hi));
        s.max_push_vel = cfg->max_push_vel; s.push_time = (double)cfg->push_interval;
        std::memcpy(s.rew_scale, cfg->rew_scale, sizeof(s.rew_scale));
        for (int k = 0; k < LG_MAX_XTERMS; ++k) { s.xterm_scale[k] = cfg->xterms[k].scale; s.xterm_p0[k] = cfg->xterms[k].p[0]; }
        std::memcpy(s.traj_v_min, cfg->traj.v_min, 8); std::memcpy(s.traj_v_max, cfg->traj.v_max, 8);
        s.traj_t_low = cfg->traj.t_low; s.traj_t_high = cfg->traj.t_high;
        std::memcpy(s.traj_max_rom_dist, cfg->traj.max_rom_dist, 8);
        stage_apply(*e, s, 3);
    }
    auto z = [](std::vector<float> &v, size_t n) { v.assign(n, 0.0f); };
    z(e->root, (size_t)N * 13); z(e->dof, (size_t)N * A * 2); z(e->contact, (size_t)N * B * 3);
    z(e->torques, (size_t)N * A); z(e->actions, (size_t)N * A); z(e->obs, (size_t)N * O); z(e->rew, N);
    z(e->commands, (size_t)N * 4); z(e->last_actions, (size_t)N * A); z(e->last_dof_vel, (size_t)N * A);
    z(e->last_root_vel, (size_t)N * 6); z(e->feet_air_time, (size_t)N * F);
    z(e->episode_sums, (size_t)LG_NUM_TERMS * N); z(e->base_lin_vel, (size_t)N * 3);
    z(e->base_ang_vel, (size_t)N * 3); z(e->proj_grav, (size_t)N * 3); z(e->heights, (size_t)N * (H ? H : 1));
    z(e->env_origins, (size_t)N * 3); z(e->lstm_h, (size_t)2 * N * A * 8); z(e->lstm_c, (size_t)2 * N * A * 8);
    z(e->friction, N); z(e->base_mass_delta, N); z(e->extras_episode, LG_NUM_TERMS); z(e->material, (size_t)N * 4);
    z(e->extras_terrain_level, 1); z(e->extras_episode_acc, LG_NUM_TERMS + 2);
    {
        const int npts = cfg->traj.enabled ? cfg->traj.N * cfg->traj.dN + 1 : 1, nobs = cfg->traj.enabled ? cfg->traj.N : 1;
        z(e->tg_state, (size_t)N * LG_TG_STRIDE); z(e->tg_traj, (size_t)N * npts * 2); z(e->trajectory, (size_t)N * nobs * 2);
        z(e->prev_error, (size_t)N * 2); z(e->push_timer, N);
    } z(e->inj_u, (size_t)N * e->K);
    for (int i = 0; i < N; ++i) { e->root[(size_t)i * 13 + 6] = 1.0f; e->friction[i] = 1.0f; }
    e->reset.assign(N, 1); e->time_out.assign(N, 0); e->last_contacts.assign((size_t)N * F, 0);
    e->extras_time_outs.assign(N, 0); e->fault.assign(N, 0); e->ep_len.assign(N, 0); e->terrain_levels.assign(N, 0);
    e->terrain_types.assign(N, 0); e->inj_levels.assign(N, 0); e->n_reset.assign(1, 0);
    e->n_fault.assign(1, 0); e->fault_total.assign(1, 0); e->n_vel_clamp.assign(1, 0); e->vel_clamp_total.assign(1, 0);
    *out = e;
    return 0;
}

int lgo_destroy(void *ctx) { delete (Env *)ctx; return 0; }

int lgo_get_buffers(void *ctx, lg_buffers *b) {
    Env *e = (Env *)ctx;
    b->root_states = e->root.data(); b->dof_state = e->dof.data(); b->contact_forces = e->contact.data();
    b->torques = e->torques.data(); b->actions = e->actions.data(); b->obs = e->obs.data(); b->rew = e->rew.data();
    b->reset = e->reset.data(); b->time_out = e->time_out.data(); b->episode_length = e->ep_len.data();
    b->commands = e->commands.data(); b->last_actions = e->last_actions.data(); b->last_dof_vel = e->last_dof_vel.data();
    b->last_root_vel = e->last_root_vel.data(); b->feet_air_time = e->feet_air_time.data();
    b->last_contacts = e->last_contacts.data(); b->episode_sums = e->episode_sums.data();
    b->base_lin_vel = e->base_lin_vel.data(); b->base_ang_vel = e->base_ang_vel.data();
    b->projected_gravity = e->proj_grav.data(); b->measured_heights = e->heights.data();
    b->env_origins = e->env_origins.data(); b->terrain_levels = e->terrain_levels.data();
    b->terrain_types = e->terrain_types.data(); b->lstm_h = e->lstm_h.data(); b->lstm_c = e->lstm_c.data();
    b->friction = e->friction.data(); b->base_mass_delta = e->base_mass_delta.data();
    b->extras_episode = e->extras_episode.data(); b->extras_terrain_level = e->extras_terrain_level.data();
    b->extras_time_outs = e->extras_time_outs.data(); b->n_reset = e->n_reset.data();
    b->extras_episode_acc = e->extras_episode_acc.data(); b->n_fault = e->n_fault.data(); b->fault_total = e->fault_total.data();
    b->n_vel_clamp = e->n_vel_clamp.data(); b->vel_clamp_total = e->vel_clamp_total.data();
    b->tg_state = e->tg_state.data(); b->tg_traj = e->tg_traj.data(); b->trajectory = e->trajectory.data();
    b->prev_error = e->prev_error.data(); b->push_timer = e->push_timer.data();
    b->inject_uniforms = e->inj_u.data(); b->inject_levels = e->inj_levels.data(); b->material = e->material.data();
    return 0;
}

int lgo_set_step_counter(void *ctx, int64_t c) { ((Env *)ctx)->step_counter = c; return 0; }
int64_t lgo_get_step_counter(void *ctx) { return ((Env *)ctx)->step_counter; }
int lgo_set_init_done(void *ctx, int v) { ((Env *)ctx)->init_done = v; return 0; }
int lgo_inject_uniforms(void *ctx, int enable) { ((Env *)ctx)->inject = enable; return 0; }

int lgo_get_stage(void *ctx, lg_stage *out) { Env *e = (Env *)ctx; *out = e->has_pending ? e->pending : e->stage; return 0; }
int lgo_set_curriculum_stage(void *ctx, const lg_stage *s, int in_callback) {
    Env *e = (Env *)ctx;
    if (in_callback) { e->pending = *s; e->has_pending = 1; return 0; }
    e->has_pending = 0;
    stage_apply(*e, *s, 3);
    return 0;
}

int lgo_set_actions(void *ctx, const float *actions) {                 // legged_robot.py:86-87
    Env *e = (Env *)ctx;
    const float c = e->cfg.clip_actions;
    for (size_t k = 0; k < (size_t)e->N * e->A; ++k) e->actions[k] = std::fmin(std::fmax(actions[k], -c), c);
    return 0;
}
int lgo_compute_torques(void *ctx) { compute_torques(*(Env *)ctx); return 0; }
int lgo_simulate(void *ctx) { simulate(*(Env *)ctx); return 0; }
int lgo_post_physics_step(void *ctx) { post_physics_step(*(Env *)ctx); return 0; }
int lgo_reset_all(void *ctx) { reset_all(*(Env *)ctx); return 0; }
int lgo_reset_ids(void *ctx, const int32_t *ids, int n) { reset_ids(*(Env *)ctx, ids, n); return 0; }

int lgo_step(void *ctx, const float *actions) {                        // legged_robot.py:80-104
    Env *e = (Env *)ctx;
    lgo_set_actions(ctx, actions);
    for (int d = 0; d < e->cfg.decimation; ++d) { compute_torques(*e); simulate(*e); }
    post_physics_step(*e);
    return 0;
}

}  // extern "C"
