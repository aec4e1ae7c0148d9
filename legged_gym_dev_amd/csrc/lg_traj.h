// Reduced-order-model trajectory generator of the trajectory-tracking env, per env, on the device.
// Reference: trajopt/rom_dynamics.py ("RD": TrajectoryGenerator :441-616, SingleInt2D :182-212) with the samplers of
// deep_tube_learning/utils.py ("DU" :27-53), driven by legged_gym/envs/base/legged_robot_trajectory.py ("LT" :405-417,
// 222-229).  The generator is a small per-env state machine (lg_buffers.tg_state, LG_TG_* in legged_hip.h): one lane steps
// one env -- a few dozen flops and, on the rare resample, 20 Philox draws -- inside the post-step kernel; nothing about it is
// bandwidth- or MFMA-shaped.  fp32 operations in the order torch evaluates them (the discrete events -- resample when
// t > t_final, ROM step when t >= k rom_dt - 1e-5 -- must fall on the same steps as in the reference).
#pragma once
#include "lg_device.h"

__device__ __forceinline__ float tg_uni(const DevParams *P, int env, int slot, int64_t counter, int inject) {
    if (inject) return P->buf.inject_uniforms[(size_t)env * P->K + slot];
    return philox_uniform(P->cfg.seed, (uint32_t)(P->cfg.env_offset + env), (uint64_t)counter, (uint32_t)slot);
}

// the generator parameters a curriculum stage rewrites (LT:533-546): ROM input bounds and the hold-time sampler
struct TgPar { float v_min[2], v_max[2], t_low, t_high, freq_low, freq_high, prob_stationary; };
__device__ __forceinline__ TgPar tg_par_cfg(const DevParams *P) {          // the values in force after the step callback
    const lg_traj_cfg &c = P->cfg.traj;
    return {{c.v_min[0], c.v_min[1]}, {c.v_max[0], c.v_max[1]}, c.t_low, c.t_high, c.freq_low, c.freq_high, c.prob_stationary};
}
__device__ __forceinline__ TgPar tg_par_cb(const DevParams *P) {           // ... inside the callback (lg_device.h StageCb)
    const lg_traj_cfg &c = P->cfg.traj;
    const StageCb &b = P->cb;
    return {{b.v_min[0], b.v_min[1]}, {b.v_max[0], b.v_max[1]}, b.t_low, b.t_high, c.freq_low, c.freq_high, c.prob_stationary};
}
// RD:507-515
__device__ inline void tg_resample(const DevParams *P, const TgPar &t, int i, int slot0, int64_t counter, int inject) {
#pragma clang fp contract(off)      // one rounding per torch op: the event comparisons below must agree with the reference
    float *s = P->buf.tg_state + (size_t)i * LG_TG_STRIDE;
    const float pi = 3.14159265358979323846f;
#pragma unroll
    for (int d = 0; d < 2; ++d) s[LG_TG_CONST + d] = (t.v_max[d] - t.v_min[d]) * tg_uni(P, i, slot0 + d, counter, inject) + t.v_min[d];
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        s[LG_TG_RAMP_V0 + d] = s[LG_TG_RAMP_V1 + d];
        s[LG_TG_RAMP_V1 + d] = (t.v_max[d] - t.v_min[d]) * tg_uni(P, i, slot0 + 2 + d, counter, inject) + t.v_min[d];
    }
    s[LG_TG_RAMP_T0] = s[LG_TG_T_FINAL];
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        const float u = tg_uni(P, i, slot0 + 4 + d, counter, inject);
        int c = inject ? (int)u : (int)(u * 3.0f);                 // injected: torch.randint's value itself
        c = c > 2 ? 2 : c;
        s[LG_TG_EXTREME + d] = c == 0 ? t.v_min[d] : c == 1 ? 0.0f : t.v_max[d];
    }
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        const float half = (t.v_max[d] - t.v_min[d]) / 2.0f;
        s[LG_TG_SIN_MAG + d] = (half - 0.0f) * tg_uni(P, i, slot0 + 6 + d, counter, inject) + 0.0f;
    }
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        const float lo = t.v_min[d] + s[LG_TG_SIN_MAG + d], hi = t.v_max[d] - s[LG_TG_SIN_MAG + d];
        s[LG_TG_SIN_MEAN + d] = (hi - lo) * tg_uni(P, i, slot0 + 8 + d, counter, inject) + lo;
    }
#pragma unroll
    for (int d = 0; d < 2; ++d) s[LG_TG_SIN_FREQ + d] = (t.freq_high - t.freq_low) * tg_uni(P, i, slot0 + 10 + d, counter, inject) + t.freq_low;
#pragma unroll
    for (int d = 0; d < 2; ++d) s[LG_TG_SIN_OFF + d] = (pi - (-pi)) * tg_uni(P, i, slot0 + 12 + d, counter, inject) + (-pi);
    s[LG_TG_T_FINAL] += (t.t_high - t.t_low) * tg_uni(P, i, slot0 + 14, counter, inject) + t.t_low;
    float w[4], sum = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) { w[k] = tg_uni(P, i, slot0 + 15 + k, counter, inject); sum += w[k]; }
#pragma unroll
    for (int k = 0; k < 4; ++k) s[LG_TG_W + k] = w[k] / sum;
    s[LG_TG_STATIONARY] = ((1.0f - 0.0f) * tg_uni(P, i, slot0 + 19, counter, inject) + 0.0f) < t.prob_stationary ? 1.0f : 0.0f;
}

// RD:559-565 without the resample
__device__ inline void tg_input(const DevParams *P, int i, float tt, float v[2]) {
#pragma clang fp contract(off)      // one rounding per torch op: the event comparisons below must agree with the reference
    const float *s = P->buf.tg_state + (size_t)i * LG_TG_STRIDE;
    const float r = (tt - s[LG_TG_RAMP_T0]) / (s[LG_TG_T_FINAL] - s[LG_TG_RAMP_T0]);
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        const float ramp = s[LG_TG_RAMP_V0 + d] + (s[LG_TG_RAMP_V1 + d] - s[LG_TG_RAMP_V0 + d]) * r;
        const float sinus = s[LG_TG_SIN_MAG + d] * sinf(s[LG_TG_SIN_FREQ + d] * tt + s[LG_TG_SIN_OFF + d]) + s[LG_TG_SIN_MEAN + d];
        float x = s[LG_TG_W + 0] * s[LG_TG_CONST + d] + s[LG_TG_W + 1] * ramp + s[LG_TG_W + 2] * s[LG_TG_EXTREME + d] + s[LG_TG_W + 3] * sinus;
        v[d] = s[LG_TG_STATIONARY] != 0.0f ? 0.0f : x;
    }
}

// The window of ROM states of env i while it is being stepped: a workspace of LG_TG_WIN floats in LDS handed in by the kernel
// (one per lane that runs a generator; odd stride between lanes).  As a register array it was the one object the post-step
// kernel kept in scratch (34 floats per lane: the allocator would not keep the dynamically indexed copy in VGPRs under that
// kernel's register budget).  Loaded whole before the first store to the window's home in HBM, as before.
#define LG_TG_WIN (2 * LG_TRAJ_MAX_PTS + 1)
__device__ inline void tg_window_load(const DevParams *P, int i, float *__restrict__ w) {
    const int n2 = 2 * (P->cfg.traj.N * P->cfg.traj.dN + 1);
    const float *z = P->buf.tg_traj + (size_t)i * n2;
    for (int p = 0; p < n2; ++p) w[p] = z[p];
}
__device__ inline void tg_window_store(const DevParams *P, int i, const float *__restrict__ w) {
    const int n2 = 2 * (P->cfg.traj.N * P->cfg.traj.dN + 1);
    float *z = P->buf.tg_traj + (size_t)i * n2;
    for (int p = 0; p < n2; ++p) z[p] = w[p];
}
// RD:578-592 (SingleInt2D.f RD:192-193: z+ = z + rom_dt v): shift the window by one point and append the new state
__device__ inline void tg_window_step(const DevParams *P, float *__restrict__ w, const float v[2]) {
#pragma clang fp contract(off)
    const lg_traj_cfg &t = P->cfg.traj;
    const int n2 = 2 * (t.N * t.dN + 1);
    const float z0 = w[n2 - 2] + t.rom_dt * v[0], z1 = w[n2 - 1] + t.rom_dt * v[1];
    for (int p = 0; p < n2 - 2; ++p) w[p] = w[p + 2];
    w[n2 - 2] = z0; w[n2 - 1] = z1;
}
// RD:610-615: the window interpolated at the env's time (t, k already advanced)
__device__ inline void tg_window_interpolate(const DevParams *P, int i, const float *__restrict__ w, float tnow, float know) {
#pragma clang fp contract(off)
    const lg_traj_cfg &t = P->cfg.traj;
    const float frac = tnow - (know - 1.0f) * t.rom_dt;
    float *out = P->buf.trajectory + (size_t)i * t.N * 2;
    for (int p = 0; p < t.N; ++p)                        // dN == 1 (checked at lg_create): points p and p + 1
        for (int d = 0; d < 2; ++d) {
            const float a = w[2 * p + d], b = w[2 * p + 2 + d];
            out[2 * p + d] = a + (b - a) * frac / t.rom_dt;
        }
}

// LT:409-411: traj_gen.step() (RD:567-576) + get_trajectory
__device__ inline void tg_callback_step(const DevParams *P, int i, int64_t counter, int inject, float *__restrict__ w) {
#pragma clang fp contract(off)      // one rounding per torch op: the event comparisons below must agree with the reference
    const lg_traj_cfg &t = P->cfg.traj;
    float *s = P->buf.tg_state + (size_t)i * LG_TG_STRIDE;
    tg_window_load(P, i, w);
    const float tt = s[LG_TG_T];
    if (tt > s[LG_TG_T_FINAL]) tg_resample(P, tg_par_cb(P), i, LG_TSLOT_TG, counter, inject);
    float v[2];
    tg_input(P, i, tt, v);
    float k = s[LG_TG_K];
    const bool rom = tt >= k * t.rom_dt - 1e-5f;
    if (rom) { tg_window_step(P, w, v); k += 1.0f; }
    const float tn = tt + P->cfg.dt;
    s[LG_TG_V] = v[0]; s[LG_TG_V + 1] = v[1];                      // self.v (RD:579-580)
    s[LG_TG_K] = k;
    s[LG_TG_T] = tn;
    if (rom) tg_window_store(P, i, w);
    tg_window_interpolate(P, i, w, tn, k);
}

// RD:597-608 with the start state z0 (LT:222-229)
__device__ inline void tg_reset(const DevParams *P, int i, float z0x, float z0y, int64_t counter, int inject, float *__restrict__ w) {
#pragma clang fp contract(off)      // one rounding per torch op: the event comparisons below must agree with the reference
    const lg_traj_cfg &t = P->cfg.traj;
    const int npts = t.N * t.dN + 1, A = P->cfg.num_actions;
    float *s = P->buf.tg_state + (size_t)i * LG_TG_STRIDE;
    for (int p = 0; p < 2 * (npts - 1); ++p) w[p] = 0.0f;
    w[2 * (npts - 1)] = z0x; w[2 * (npts - 1) + 1] = z0y;
    s[LG_TG_K] = -(float)(t.N * t.dN);
    s[LG_TG_T] = s[LG_TG_K] * t.rom_dt;
    s[LG_TG_T_FINAL] = s[LG_TG_K] * t.rom_dt;
    tg_resample(P, tg_par_cfg(P), i, LG_TSLOT_RTG(A), counter, inject);
    float tt = s[LG_TG_T], k = s[LG_TG_K], v[2] = {0.0f, 0.0f};
    for (int it = 0; it < t.N * t.dN; ++it) {
        tg_input(P, i, tt, v);
        tg_window_step(P, w, v);
        k += 1.0f;
        tt += t.rom_dt;
    }
    s[LG_TG_V] = v[0]; s[LG_TG_V + 1] = v[1];
    s[LG_TG_K] = k;
    s[LG_TG_T] = tt;
    tg_window_store(P, i, w);
}

// the reset loop's get_input_t reaches every env (RD:579): see legged_hip.h LG_TSLOT_RTG
__device__ inline void tg_late_resample(const DevParams *P, int i, int64_t counter, int inject) {
#pragma clang fp contract(off)      // one rounding per torch op: the event comparisons below must agree with the reference
    float *s = P->buf.tg_state + (size_t)i * LG_TG_STRIDE;
    if (s[LG_TG_T] > s[LG_TG_T_FINAL]) tg_resample(P, tg_par_cfg(P), i, LG_TSLOT_RTG(P->cfg.num_actions), counter, inject);
    float v[2];                                                    // and leaves self.v evaluated at the env's new time
    tg_input(P, i, s[LG_TG_T], v);
    s[LG_TG_V] = v[0]; s[LG_TG_V + 1] = v[1];
}
