// CPU ORACLE -- TEST INFRASTRUCTURE ONLY (see lgo_common.h).
// Scalar restatement of the reduced-order-model trajectory generator the trajectory-tracking env steps every policy step:
// /root/reference/trajopt/rom_dynamics.py ("RD": TrajectoryGenerator :441-616, SingleInt2D :182-212, RomDynamics :108-124)
// with the samplers of deep_tube_learning/utils.py ("DU": UniformSampleHoldDT :27-43, UniformWeightSampler :46-53), as
// legged_robot_trajectory.py ("LT") drives it.  float32 throughout, operations in the order torch evaluates them.  Pinned
// by tests/golden/anymal_c_flat_trajectory.npz (oracle/gen_fixtures_trajectory.py ran those files themselves).
#include "lgo_common.h"

namespace lgo {

float uni(const Env &e, int env, int slot) {
    if (e.inject) return e.inj_u[(size_t)env * e.K + slot];
    return philox_uniform(e.cfg.seed, (uint32_t)(e.cfg.env_offset + env), (uint64_t)e.step_counter, (uint32_t)slot);
}

// RD:507-515: redraw every input law, the hold time, the mixing weights and the stationary flag of env i.
// uniform(low, high) = (high - low) * rand + low (RD:478-479)
static void tg_resample(Env &e, const lg_traj_cfg &t, int i, int slot0) {
    float *s = &e.tg_state[(size_t)i * LG_TG_STRIDE];
    const float pi = 3.14159265358979323846f;
    for (int d = 0; d < 2; ++d)                                             // _resample_const_input RD:523-524
        s[LG_TG_CONST + d] = (t.v_max[d] - t.v_min[d]) * uni(e, i, slot0 + d) + t.v_min[d];
    for (int d = 0; d < 2; ++d) {                                           // _resample_ramp_input RD:526-529 (clip_v_z is the identity)
        s[LG_TG_RAMP_V0 + d] = s[LG_TG_RAMP_V1 + d];
        s[LG_TG_RAMP_V1 + d] = (t.v_max[d] - t.v_min[d]) * uni(e, i, slot0 + 2 + d) + t.v_min[d];
    }
    s[LG_TG_RAMP_T0] = s[LG_TG_T_FINAL];
    for (int d = 0; d < 2; ++d) {                                           // _resample_extreme_input RD:531-534: v_min | 0 | v_max
        const float u = uni(e, i, slot0 + 4 + d);
        int c = e.inject ? (int)u : (int)(u * 3.0f);                        // the fixture stores torch.randint's value itself
        c = c > 2 ? 2 : c;
        s[LG_TG_EXTREME + d] = c == 0 ? t.v_min[d] : c == 1 ? 0.0f : t.v_max[d];
    }
    for (int d = 0; d < 2; ++d) {                                           // _resample_sinusoid_input RD:536-540
        const float half = (t.v_max[d] - t.v_min[d]) / 2.0f;
        s[LG_TG_SIN_MAG + d] = (half - 0.0f) * uni(e, i, slot0 + 6 + d) + 0.0f;
    }
    for (int d = 0; d < 2; ++d) {
        const float lo = t.v_min[d] + s[LG_TG_SIN_MAG + d], hi = t.v_max[d] - s[LG_TG_SIN_MAG + d];
        s[LG_TG_SIN_MEAN + d] = (hi - lo) * uni(e, i, slot0 + 8 + d) + lo;
    }
    for (int d = 0; d < 2; ++d) s[LG_TG_SIN_FREQ + d] = (t.freq_high - t.freq_low) * uni(e, i, slot0 + 10 + d) + t.freq_low;
    for (int d = 0; d < 2; ++d) s[LG_TG_SIN_OFF + d] = (pi - (-pi)) * uni(e, i, slot0 + 12 + d) + (-pi);
    s[LG_TG_T_FINAL] += (t.t_high - t.t_low) * uni(e, i, slot0 + 14) + t.t_low;          // RD:517-518, DU:38-39
    float w[4], sum = 0.0f;                                                 // DU:51-53
    for (int k = 0; k < 4; ++k) { w[k] = uni(e, i, slot0 + 15 + k); sum += w[k]; }
    for (int k = 0; k < 4; ++k) s[LG_TG_W + k] = w[k] / sum;
    s[LG_TG_STATIONARY] = ((1.0f - 0.0f) * uni(e, i, slot0 + 19) + 0.0f) < t.prob_stationary ? 1.0f : 0.0f;   // RD:515
}

// RD:559-565 without the resample: the mixed input at time tt
static void tg_input(const Env &e, int i, float tt, float v[2]) {
    const float *s = &e.tg_state[(size_t)i * LG_TG_STRIDE];
    const float r = (tt - s[LG_TG_RAMP_T0]) / (s[LG_TG_T_FINAL] - s[LG_TG_RAMP_T0]);         // RD:545-547
    for (int d = 0; d < 2; ++d) {
        const float ramp = s[LG_TG_RAMP_V0 + d] + (s[LG_TG_RAMP_V1 + d] - s[LG_TG_RAMP_V0 + d]) * r;
        const float sinus = s[LG_TG_SIN_MAG + d] * std::sin(s[LG_TG_SIN_FREQ + d] * tt + s[LG_TG_SIN_OFF + d]) + s[LG_TG_SIN_MEAN + d];   // RD:552-553
        v[d] = s[LG_TG_W + 0] * s[LG_TG_CONST + d] + s[LG_TG_W + 1] * ramp + s[LG_TG_W + 2] * s[LG_TG_EXTREME + d] + s[LG_TG_W + 3] * sinus;
        if (s[LG_TG_STATIONARY] != 0.0f) v[d] = 0.0f;                        // RD:580
    }
}
static void tg_keep_v(Env &e, int i, const float v[2]) {                     // self.v (RD:579-580)
    e.tg_state[(size_t)i * LG_TG_STRIDE + LG_TG_V] = v[0];
    e.tg_state[(size_t)i * LG_TG_STRIDE + LG_TG_V + 1] = v[1];
}

// RD:578-592: one ROM step of env i: z+ = A z + B v with A = I, B = rom_dt I (SingleInt2D.f RD:192-193), window shifted by one
static void tg_rom_step(Env &e, int i, const float v[2]) {
    const lg_traj_cfg &t = e.cfg.traj;
    const int npts = t.N * t.dN + 1;
    float *z = &e.tg_traj[(size_t)i * npts * 2];
    const float zn[2] = {z[2 * (npts - 1)] + t.rom_dt * v[0], z[2 * (npts - 1) + 1] + t.rom_dt * v[1]};
    for (int p = 0; p + 1 < npts; ++p) { z[2 * p] = z[2 * (p + 1)]; z[2 * p + 1] = z[2 * (p + 1) + 1]; }
    z[2 * (npts - 1)] = zn[0]; z[2 * (npts - 1) + 1] = zn[1];
    e.tg_state[(size_t)i * LG_TG_STRIDE + LG_TG_K] += 1.0f;
}

// RD:610-615 get_trajectory: the window interpolated at the env's time
static void tg_interpolate(Env &e, int i) {
    const lg_traj_cfg &t = e.cfg.traj;
    const int npts = t.N * t.dN + 1;
    const float *s = &e.tg_state[(size_t)i * LG_TG_STRIDE];
    const float *z = &e.tg_traj[(size_t)i * npts * 2];
    const float frac = s[LG_TG_T] - (s[LG_TG_K] - 1.0f) * t.rom_dt;
    for (int p = 0; p < t.N; ++p)
        for (int d = 0; d < 2; ++d) {
            const float a = z[2 * (p * t.dN) + d], b = z[2 * (p * t.dN + 1) + d];
            e.trajectory[((size_t)i * t.N + p) * 2 + d] = a + (b - a) * frac / t.rom_dt;
        }
}

// LT:409-411: traj_gen.step() (RD:567-576) + get_trajectory for env i
void tg_callback_step(Env &e, int i) {
    const lg_traj_cfg &t = e.cfg.traj;
    float *s = &e.tg_state[(size_t)i * LG_TG_STRIDE];
    const float tt = s[LG_TG_T];
    if (tt > s[LG_TG_T_FINAL]) tg_resample(e, e.cb.traj, i, LG_TSLOT_TG);  // get_input_t RD:560-561 (every env, every step)
    float v[2];
    tg_input(e, i, tt, v);
    tg_keep_v(e, i, v);
    if (tt >= s[LG_TG_K] * t.rom_dt - 1e-5f) tg_rom_step(e, i, v);         // RD:572
    s[LG_TG_T] = tt + e.cfg.dt;                                             // RD:574 (dt_loop = env dt)
    tg_interpolate(e, i);
}

// LT:222-229 hands z = proj_z(root) (+ random start offset) to traj_gen.reset_idx RD:597-608
void tg_reset(Env &e, int i, const float z0[2]) {
    const lg_traj_cfg &t = e.cfg.traj;
    const int npts = t.N * t.dN + 1, A = e.A;
    float *s = &e.tg_state[(size_t)i * LG_TG_STRIDE];
    float *z = &e.tg_traj[(size_t)i * npts * 2];
    for (int p = 0; p < 2 * npts; ++p) z[p] = 0.0f;
    z[2 * (npts - 1)] = z0[0]; z[2 * (npts - 1) + 1] = z0[1];
    s[LG_TG_K] = -(float)(t.N * t.dN);
    s[LG_TG_T] = s[LG_TG_K] * t.rom_dt;
    s[LG_TG_T_FINAL] = s[LG_TG_K] * t.rom_dt;
    tg_resample(e, e.cfg.traj, i, LG_TSLOT_RTG(A));
    for (int it = 0; it < t.N * t.dN; ++it) {                               // step_rom_idx(idx, increment_rom_time=True)
        float v[2];
        tg_input(e, i, s[LG_TG_T], v);
        tg_keep_v(e, i, v);
        tg_rom_step(e, i, v);
        s[LG_TG_T] += t.rom_dt;
    }
}

// The reset loop above calls get_input_t for EVERY env (RD:579), so on a step where at least one env resets, an env that
// did not reset but whose hold time ran out after the callback's time increment is resampled there.
void tg_late_resample(Env &e, int i) {
    const float *s = &e.tg_state[(size_t)i * LG_TG_STRIDE];
    if (s[LG_TG_T] > s[LG_TG_T_FINAL]) tg_resample(e, e.cfg.traj, i, LG_TSLOT_RTG(e.A));
    float v[2];                                                              // and leaves self.v evaluated at the env's new time
    tg_input(e, i, s[LG_TG_T], v);
    tg_keep_v(e, i, v);
}

}  // namespace lgo
