// Device-side common definitions for liblegged_hip.so (gfx950 / wave64 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/legged_hip.h"

#define LG_WAVE 64
#ifndef LG_TILE_THREADS
#define LG_TILE_THREADS 512         // threads of a post-step workgroup (16 envs): the height scan, the observation segments and the noise
                                    // pass are lane-parallel over (env, entry), so 32 lanes per env halve their trips -- 23.9 -> 18.8 us
                                    // rough terrain, 49.8 -> 46.7 Cassie, flat unchanged (256: phase A's 16 lanes per env only)
#endif
#define LG_MAX_LEG_SLOTS 6
#define LG_MAX_BASE_PER_LANE 2      // base collision spheres are dealt to the lanes of an env: sphere b -> lane b % L, slot b / L
#define LG_NUM_SLOTS (LG_MAX_LEG_SLOTS + LG_MAX_BASE_PER_LANE)
// per-leg constant table: joint j at LG_LT_JOINT*j: R_pj 0..8, p_pj 9..11, axis 12..14, child-link inertia 15..23,
// com 24..26, mass 27, joint damping 28, velocity limit 29, position limits 30 (lower) 31 (upper); then sphere slot s:
// centre 3 + radius; then the
// base spheres of this lane (radius 0 = none).  Stride odd in 4-byte words: the L legs of a wave sit in different LDS banks.
#define LG_LT_JOINT 32
#define LG_LT_MAXJ 6
#define LG_LT_SLOTS (LG_LT_JOINT * LG_LT_MAXJ)
#define LG_LT_BASE (LG_LT_SLOTS + 4 * LG_MAX_LEG_SLOTS)
#define LG_LT_STRIDE (LG_LT_BASE + 4 * LG_MAX_BASE_PER_LANE + 1)

// Everything a kernel needs, resident in HBM and passed by pointer: uniform (scalar-unit) loads.
// LDS image of the actuator-net weights for k_substeps (env_kernels.hip: one aligned record per hidden unit), built on the host
#define LG_LSTM_REC 116
#define LG_LSTM_LDS (4 + 8 * LG_LSTM_REC)
// float o of the image, from the lg_cfg.lstm_w blob (the two bias vectors of a layer pre-added, as the kernel adds them first)
static inline float lstm_lds_image(const float *__restrict__ w, int o) {
    constexpr int WIH0 = 3, WHH0 = WIH0 + 64, BIH0 = WHH0 + 256, BHH0 = BIH0 + 32, WIH1 = BHH0 + 32, WHH1 = WIH1 + 256,
                  BIH1 = WHH1 + 256, BHH1 = BIH1 + 32, LW = BHH1 + 32;
    if (o < 4) return o < 3 ? w[o] : w[LG_LSTM_NW - 1];                    // in_scale 2, out_scale, lin_b
    const int k = (o - 4) / LG_LSTM_REC, f = (o - 4) % LG_LSTM_REC;
    if (f < 4) return w[BIH0 + f * 8 + k] + w[BHH0 + f * 8 + k];
    if (f < 8) return w[BIH1 + (f - 4) * 8 + k] + w[BHH1 + (f - 4) * 8 + k];
    if (f < 16) return w[WIH0 + (((f - 8) >> 1) * 8 + k) * 2 + ((f - 8) & 1)];
    if (f < 112) {
        // the three 32 x 8 matrices in the order lane k CONSUMES them (env_kernels.hip lstm_acc): for rotation r = 0..3 of the row's hidden
        // values inside their quad, four gate weights for the unit that lands on lane k itself and four for the unit that lands on lane 7 - k
        const int m = (f - 16) >> 5, e = (f - 16) & 31, r = e >> 3, other = (e >> 2) & 1, gt = e & 3;
        const int src = other ? 7 - k : k, unit = (src & 4) + ((src + r) & 3);
        const int base = m == 0 ? WHH0 : m == 1 ? WIH1 : WHH1;
        return w[base + (gt * 8 + k) * 8 + unit];
    }
    return f == 112 ? w[LW + k] : 0.0f;
}

// What the step CALLBACK reads of a curriculum stage (legged_hip.h lg_stage): the command resample, the push and the generator
// resample of _post_physics_step_callback see the values of the stage that was in force when the step began; everything after the
// callback reads cfg.  Outside a stage-change step the two agree.
struct StageCb {
    float cmd_lo[4], cmd_hi[4];
    float max_push_vel;
    float v_min[2], v_max[2], t_low, t_high;
};
struct DevParams {
    lg_cfg cfg;          // host pointers inside are NOT valid on device (use the ones below)
    StageCb cb;
    lg_model model;
    lg_buffers buf;
    const float *noise_vec;        // num_obs
    const float *height_points;    // H x 2
    const float *terrain_origins;  // levels x types x 3
    const int16_t *height_samples; // hf_rows x hf_cols
    float *ep_accum;               // LG_NUM_REWARDS sums of episode_sums over resetting envs
    int32_t *reset_count;          // 1
    int32_t *fault_count;          // 1: faults consumed by the running step (block atomics), folded by k_finalize
    uint8_t *fault;                // N: set by the physics fault guard, consumed by the post-step
    int32_t *clamp_count;          // 1: base-velocity clamps (lg_cfg.max_linear_velocity / max_angular_velocity) since the last k_finalize
    uint8_t *reset_mark;           // N: envs reset since the last k_finalize (which clears it)
    int64_t *any_reset_step;       // 1: step counter of the last step on which at least one env reset (written by every post-step
                                   // workgroup that resets one; read by the fused rollout epilogue, ppo_mlp_fused.hip)
    unsigned long long *dbg_cycles; // 8 per post-step workgroup (first 64 workgroups): s_memtime at the phase boundaries (tools/post_step_phases.py)
    int K;                         // uniforms per env
    // per-leg sphere tables for the lane-parallel physics: slot-major [slot][leg]
    int n_leg_slots, n_base_spheres;
    int slot_link[LG_MAX_SPHERES];           // joint index within the chain (same for every leg)
    unsigned long long slot_link_pk;
    float lstm_img[LG_LSTM_LDS];         // the same, 4 bits per slot (one scalar load for the physics lanes)
    int slot_body[LG_MAX_SPHERES][4 + 4];    // [slot][leg] body row (legs <= 8)
    float slot_center[LG_MAX_SPHERES][8][3];
    float slot_radius[LG_MAX_SPHERES][8];
    int base_body[8 * LG_MAX_BASE_PER_LANE];
    float base_center[8 * LG_MAX_BASE_PER_LANE][3];
    float base_radius[8 * LG_MAX_BASE_PER_LANE];
    // every model constant one (env, leg) lane of the physics reads, flattened per leg (host-built once;
    // the kernel copies it into LDS with one coalesced pass): LG_LT_* offsets below
    float leg_tab[8][LG_LT_STRIDE];
};

struct lg_ctx {
    DevParams h;              // host copy
    DevParams *d;             // device copy
    hipStream_t stream;
    int64_t step_counter;
    int init_done, inject;
    lg_stage stage, pending;  // curriculum stage in force / to take effect inside the next post-step's callback
    int has_pending;
    int defer_finalize;       // lg_step leaves its single-workgroup epilogue (k_finalize) to the learner's next act launch
    int finalize_pending;     // ... and it has not run yet
    void *allocs[128];
    int n_allocs;
};

// ------------------------------------------------------------------ small vector algebra
struct V3 { float x, y, z; };
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
struct M3 { float m[3][3]; };
__device__ __forceinline__ V3 mul(const M3 &A, V3 v) {
    return {A.m[0][0] * v.x + A.m[0][1] * v.y + A.m[0][2] * v.z, A.m[1][0] * v.x + A.m[1][1] * v.y + A.m[1][2] * v.z,
            A.m[2][0] * v.x + A.m[2][1] * v.y + A.m[2][2] * v.z};
}
__device__ __forceinline__ V3 mulT(const M3 &A, V3 v) {
    return {A.m[0][0] * v.x + A.m[1][0] * v.y + A.m[2][0] * v.z, A.m[0][1] * v.x + A.m[1][1] * v.y + A.m[2][1] * v.z,
            A.m[0][2] * v.x + A.m[1][2] * v.y + A.m[2][2] * v.z};
}
__device__ __forceinline__ M3 mul(const M3 &A, const M3 &B) {
    M3 C;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) C.m[i][j] = A.m[i][0] * B.m[0][j] + A.m[i][1] * B.m[1][j] + A.m[i][2] * B.m[2][j];
    return C;
}
__device__ __forceinline__ M3 mulBT(const M3 &A, const M3 &B) {   // A * B^T
    M3 C;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) C.m[i][j] = A.m[i][0] * B.m[j][0] + A.m[i][1] * B.m[j][1] + A.m[i][2] * B.m[j][2];
    return C;
}
__device__ __forceinline__ M3 load3(const float *p) {
    M3 A;
#pragma unroll
    for (int i = 0; i < 9; ++i) A.m[i / 3][i % 3] = p[i];
    return A;
}
__device__ __forceinline__ V3 ld3(const float *p) { return {p[0], p[1], p[2]}; }

// isaacgym.torch_utils.quat_rotate_inverse / quat_apply, q = xyzw
__device__ __forceinline__ V3 quat_rotate_inverse(const float *q, V3 v) {
    V3 qv = {q[0], q[1], q[2]};
    float w = q[3], s = 2.0f * w * w - 1.0f;
    V3 cr = cross(qv, v);
    float d = dot(qv, v);
    return {v.x * s - cr.x * w * 2.0f + qv.x * d * 2.0f, v.y * s - cr.y * w * 2.0f + qv.y * d * 2.0f,
            v.z * s - cr.z * w * 2.0f + qv.z * d * 2.0f};
}
__device__ __forceinline__ V3 quat_apply(const float *q, V3 b) {
    V3 qv = {q[0], q[1], q[2]};
    V3 t = 2.0f * cross(qv, b);
    return b + q[3] * t + cross(qv, t);
}
__device__ __forceinline__ float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
// 1 / x as ONE v_rcp_f32 (1 ulp).  __frcp_rn is the correctly rounded reciprocal: hipcc expands it to the full
// v_div_scale / v_div_fmas / v_div_fixup sequence (~10 instructions) -- 60 of them per control-loop round in the actuator net.
__device__ __forceinline__ float frcp(float x) { return __builtin_amdgcn_rcpf(x); }
// lane ^ 1 / lane ^ 2 exchange inside a quad as a DPP operand modifier (quad_perm): no LDS crossbar round trip, unlike the
// ds_bpermute_b32 that __shfl_xor compiles to.  A lone wave per SIMD cannot hide that latency.
__device__ __forceinline__ float quad_xor1(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1 /*quad_perm [1,0,3,2]*/, 0xF, 0xF, true));
}
// sum over the 16 lanes of a DPP row, result in every lane: quad butterflies, then the two mirror permutes
__device__ __forceinline__ float row16_sum(float x);
__device__ __forceinline__ float quad_xor2(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E /*quad_perm [2,3,0,1]*/, 0xF, 0xF, true));
}
__device__ __forceinline__ float row16_sum(float x) {
    x += quad_xor1(x);
    x += quad_xor2(x);
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141 /*row_half_mirror*/, 0xF, 0xF, true));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140 /*row_mirror*/, 0xF, 0xF, true));
    return x;
}

// ------------------------------------------------------------------ Philox4x32-10
__device__ __forceinline__ void philox4x32(uint32_t k0, uint32_t k1, uint32_t c[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
        uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
        uint32_t n0 = hi1 ^ c[1] ^ k0, n1 = lo1, n2 = hi0 ^ c[3] ^ k1, n3 = lo0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}
// the four uniforms of slots 4 * block .. 4 * block + 3 from ONE Philox evaluation (each is what philox_uniform returns for its slot)
__device__ __forceinline__ void philox_uniform4(uint64_t seed, uint32_t env, uint64_t step, uint32_t block, float out[4]) {
    uint32_t c[4] = {env, (uint32_t)step, block, (uint32_t)(step >> 32)};
    philox4x32((uint32_t)seed, (uint32_t)(seed >> 32), c);
#pragma unroll
    for (int r = 0; r < 4; ++r) out[r] = (float)(c[r] >> 8) * (1.0f / 16777216.0f);
}
__device__ __forceinline__ float philox_uniform(uint64_t seed, uint32_t env, uint64_t step, uint32_t slot) {
    uint32_t c[4] = {env, (uint32_t)step, slot >> 2, (uint32_t)(step >> 32)};
    philox4x32((uint32_t)seed, (uint32_t)(seed >> 32), c);
    uint32_t r = (slot & 3) == 0 ? c[0] : (slot & 3) == 1 ? c[1] : (slot & 3) == 2 ? c[2] : c[3];
    return (float)(r >> 8) * (1.0f / 16777216.0f);
}
