// Lane-parallel articulated-body physics for gfx950: one lane per (environment, leg).
//
// Replaces gym.simulate() (reference call site legged_gym/envs/base/legged_robot.py:92-96).  The
// algorithm is the build's own specification (oracle/lgo_physics.cpp restates it scalar, generic
// tree): Featherstone ABA in base-frame coordinates about the base origin, exact 3x3 contact-space
// inverse inertia per sphere contact from test impulses, projected-Jacobi sweeps over contacts and
// joint-limit constraints, one tree impulse propagation per sweep, semi-implicit Euler.
//
// Mapping to CDNA4: a wave64 holds 64/L environments; the L lanes of an environment own one leg
// chain each (J revolute joints, unrolled; S, U, 1/D, u and the base inverse inertia in VGPRs), and
// meet only at the floating base through L-lane butterfly sums (DPP quad_perm operands): 27 floats for
// the articulated base inertia + bias, 6 floats per sweep.  Per-leg model constants, per-link tiles,
// contact-slot and joint-limit records live in LDS columns (field-major, one column per lane:
// conflict-free).  No barriers inside a step, no divergence between legs; contact work walks each
// lane's own list of active slots, limit work is skipped wave-uniformly when no lane needs it.
#pragma once
#include "lg_device.h"

struct Sv { V3 w, v; };
__device__ __forceinline__ Sv operator+(Sv a, Sv b) { return {a.w + b.w, a.v + b.v}; }
__device__ __forceinline__ Sv operator-(Sv a, Sv b) { return {a.w - b.w, a.v - b.v}; }
__device__ __forceinline__ Sv operator*(float s, Sv a) { return {s * a.w, s * a.v}; }
__device__ __forceinline__ float sdot(Sv a, Sv b) { return dot(a.w, b.w) + dot(a.v, b.v); }
__device__ __forceinline__ Sv crm(Sv a, Sv b) { return {cross(a.w, b.w), cross(a.w, b.v) + cross(a.v, b.w)}; }
__device__ __forceinline__ Sv crf(Sv a, Sv f) { return {cross(a.w, f.w) + cross(a.v, f.v), cross(a.w, f.v)}; }
__device__ __forceinline__ Sv sv_zero() { return {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}}; }

struct M6 { float m[6][6]; };
__device__ __forceinline__ Sv mul6(const M6 &I, Sv a) {
    float x[6] = {a.w.x, a.w.y, a.w.z, a.v.x, a.v.y, a.v.z}, y[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 6; ++j) s += I.m[i][j] * x[j];
        y[i] = s;
    }
    return {{y[0], y[1], y[2]}, {y[3], y[4], y[5]}};
}
__device__ __forceinline__ M6 rigid_inertia(float m, V3 c, const M3 &Ic) {
    M6 I;
    float cc = dot(c, c);
    float cv[3] = {c.x, c.y, c.z};
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            I.m[i][j] = Ic.m[i][j] + m * ((i == j ? cc : 0.0f) - cv[i] * cv[j]);
            I.m[3 + i][3 + j] = (i == j) ? m : 0.0f;
        }
    float hx[3][3] = {{0.f, -m * c.z, m * c.y}, {m * c.z, 0.f, -m * c.x}, {-m * c.y, m * c.x, 0.f}};
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) { I.m[i][3 + j] = hx[i][j]; I.m[3 + i][j] = hx[j][i]; }
    return I;
}
__device__ __forceinline__ bool spd_inverse6(const M6 &A, M6 &Ainv) {
    float Lm[6][6], Li[6];            // Li = 1 / diag(L): v_rsq instead of IEEE divisions
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) Lm[i][j] = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            float s = A.m[i][j];
#pragma unroll
            for (int k = 0; k < j; ++k) s -= Lm[i][k] * Lm[j][k];
            if (i == j) { ok = ok && (s > 0.0f); Li[i] = rsqrtf(fmaxf(s, 1e-30f)); Lm[i][i] = s * Li[i]; }
            else Lm[i][j] = s * Li[j];
        }
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        float y[6], x[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            float s = (i == c) ? 1.0f : 0.0f;
#pragma unroll
            for (int k = 0; k < i; ++k) s -= Lm[i][k] * y[k];
            y[i] = s * Li[i];
        }
#pragma unroll
        for (int i = 5; i >= 0; --i) {
            float s = y[i];
#pragma unroll
            for (int k = i + 1; k < 6; ++k) s -= Lm[k][i] * x[k];
            x[i] = s * Li[i];
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) Ainv.m[i][c] = x[i];
    }
    return ok;
}
__device__ __forceinline__ M3 rodrigues(V3 a, float th) {
    const float s = __sinf(th), c = __cosf(th);      // |th| is a joint angle: the fast forms are accurate to ~1e-6
    float t = 1.0f - c;
    M3 R = {{{c + t * a.x * a.x, t * a.x * a.y - s * a.z, t * a.x * a.z + s * a.y},
             {t * a.x * a.y + s * a.z, c + t * a.y * a.y, t * a.y * a.z - s * a.x},
             {t * a.x * a.z - s * a.y, t * a.y * a.z + s * a.x, c + t * a.z * a.z}}};
    return R;
}
__device__ __forceinline__ M3 quat_to_mat(const float *q) {
    float x = q[0], y = q[1], z = q[2], w = q[3];
    M3 R = {{{1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)},
             {2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)},
             {2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)}}};
    return R;
}

template <int L>
__device__ __forceinline__ float leg_sum(float x) {     // sum over the L lanes of one environment (L = 2 or 4: inside a quad)
    static_assert(L == 2 || L == 4, "legs of an environment sit in one quad of lanes");
    x += quad_xor1(x);
    if (L == 4) x += quad_xor2(x);
    return x;
}
template <int L>
__device__ __forceinline__ Sv leg_sum(Sv a) {
    return {{leg_sum<L>(a.w.x), leg_sum<L>(a.w.y), leg_sum<L>(a.w.z)}, {leg_sum<L>(a.v.x), leg_sum<L>(a.v.y), leg_sum<L>(a.v.z)}};
}

// The scalars of DevParams the physics reads in every substep, copied into registers ONCE per launch (phys_cfg): read through P they are
// scalar loads the compiler must repeat after every workgroup barrier of the control loop (29 s_load per substep pass, each cluster a
// ~150-cycle wait for the lone physics wave).  Field names follow lg_cfg, so the physics reads `c.<name>` either way.
struct PhysCfg {
    float gravity[3];
    float max_depenetration_velocity, contact_erp, ground_restitution, ground_friction, contact_offset, bounce_threshold;
    float max_linear_velocity, max_angular_velocity, armature, rest_offset;
    float border_size, hf_hscale, hf_vscale, hf_inv_hscale;
    int hf_rows, hf_cols, terrain_type, solver_iterations, material_rand;
    int n_leg_slots, n_base_spheres;
    unsigned long long slot_link_pk;
    const int16_t *height_samples;
    float base_mass, base_com[3], base_inertia[9];
};
__device__ __forceinline__ PhysCfg phys_cfg(const DevParams *__restrict__ P) {
    const lg_cfg &c = P->cfg;
    PhysCfg k;
    for (int i = 0; i < 3; ++i) k.gravity[i] = c.gravity[i];
    k.max_depenetration_velocity = c.max_depenetration_velocity; k.contact_erp = c.contact_erp;
    k.ground_restitution = c.ground_restitution; k.ground_friction = c.ground_friction;
    k.contact_offset = c.contact_offset; k.bounce_threshold = c.bounce_threshold;
    k.max_linear_velocity = c.max_linear_velocity; k.max_angular_velocity = c.max_angular_velocity;
    k.armature = c.armature; k.rest_offset = c.rest_offset;
    k.border_size = c.border_size; k.hf_hscale = c.hf_hscale; k.hf_vscale = c.hf_vscale;
    k.hf_inv_hscale = 1.0f / c.hf_hscale;
    k.hf_rows = c.hf_rows; k.hf_cols = c.hf_cols; k.terrain_type = c.terrain_type;
    k.solver_iterations = c.solver_iterations; k.material_rand = c.material_rand;
    k.n_leg_slots = P->n_leg_slots; k.n_base_spheres = P->n_base_spheres; k.slot_link_pk = P->slot_link_pk;
    k.height_samples = P->height_samples;
    k.base_mass = P->model.mass[0];
    for (int i = 0; i < 3; ++i) k.base_com[i] = P->model.com[0][i];
    for (int i = 0; i < 9; ++i) k.base_inertia[i] = P->model.inertia[0][i];
    return k;
}

struct Ground { float h; V3 n; };
// The heightfield lookup in two halves, so that a caller with several points can have all their samples in flight before it needs the
// first (lg_physics_pair.h: eight collision spheres per lane, eight round trips to the L2 one after the other otherwise).
struct GroundTap { int16_t s00, s01, s10, s11; float tx, ty; };
template <typename CFG>
__device__ __forceinline__ GroundTap ground_fetch(const CFG &c, const int16_t *__restrict__ height_samples, float x, float y) {
    if (c.terrain_type == 0) return {0, 0, 0, 0, 0.f, 0.f};
    const float inv_h = 1.0f / c.hf_hscale;          // products with the reciprocal pitch: the definition all three implementations share
    float gx = (x + c.border_size) * inv_h, gy = (y + c.border_size) * inv_h;
    gx = fminf(fmaxf(gx, 0.0f), (float)(c.hf_rows - 1) - 1e-3f);
    gy = fminf(fmaxf(gy, 0.0f), (float)(c.hf_cols - 1) - 1e-3f);
    int ix = (int)gx, iy = (int)gy;
    const int16_t *hs = height_samples + (size_t)ix * c.hf_cols + iy;
    return {hs[0], hs[1], hs[c.hf_cols], hs[c.hf_cols + 1], gx - ix, gy - iy};
}
template <typename CFG>
__device__ __forceinline__ Ground ground_finish(const CFG &c, const GroundTap &t) {
    if (c.terrain_type == 0) return {0.0f, {0.0f, 0.0f, 1.0f}};
    const float tx = t.tx, ty = t.ty;
    float h00 = (float)t.s00 * c.hf_vscale, h01 = (float)t.s01 * c.hf_vscale;
    float h10 = (float)t.s10 * c.hf_vscale, h11 = (float)t.s11 * c.hf_vscale;
    float h = (1 - tx) * (1 - ty) * h00 + tx * (1 - ty) * h10 + (1 - tx) * ty * h01 + tx * ty * h11;
    const float inv_h = 1.0f / c.hf_hscale;
    float dhdx = ((1 - ty) * (h10 - h00) + ty * (h11 - h01)) * inv_h;
    float dhdy = ((1 - tx) * (h01 - h00) + tx * (h11 - h10)) * inv_h;
    float inv = rsqrtf(dhdx * dhdx + dhdy * dhdy + 1.0f);
    return {h, {-dhdx * inv, -dhdy * inv, inv}};
}
template <typename CFG>
__device__ __forceinline__ Ground ground_at(const CFG &c, const int16_t *__restrict__ height_samples, float x, float y) {
    return ground_finish(c, ground_fetch(c, height_samples, x, y));
}
// The control loop's own pair (PhysCfg: launch constants in registers): the four divisions by the grid pitch per point are products with
// its reciprocal (32 IEEE divisions per lane and substep otherwise, a third of the detection's instructions; the one-lane map and the oracle define the lookup the same way).
__device__ __forceinline__ GroundTap ground_fetch(const PhysCfg &c, const int16_t *__restrict__ height_samples, float x, float y) {
    if (c.terrain_type == 0) return {0, 0, 0, 0, 0.f, 0.f};
    float gx = (x + c.border_size) * c.hf_inv_hscale, gy = (y + c.border_size) * c.hf_inv_hscale;
    gx = fminf(fmaxf(gx, 0.0f), (float)(c.hf_rows - 1) - 1e-3f);
    gy = fminf(fmaxf(gy, 0.0f), (float)(c.hf_cols - 1) - 1e-3f);
    int ix = (int)gx, iy = (int)gy;
    const int16_t *hs = height_samples + (size_t)ix * c.hf_cols + iy;
    return {hs[0], hs[1], hs[c.hf_cols], hs[c.hf_cols + 1], gx - ix, gy - iy};
}
__device__ __forceinline__ Ground ground_finish(const PhysCfg &c, const GroundTap &t) {
    if (c.terrain_type == 0) return {0.0f, {0.0f, 0.0f, 1.0f}};
    const float tx = t.tx, ty = t.ty;
    float h00 = (float)t.s00 * c.hf_vscale, h01 = (float)t.s01 * c.hf_vscale;
    float h10 = (float)t.s10 * c.hf_vscale, h11 = (float)t.s11 * c.hf_vscale;
    float h = (1 - tx) * (1 - ty) * h00 + tx * (1 - ty) * h10 + (1 - tx) * ty * h01 + tx * ty * h11;
    float dhdx = ((1 - ty) * (h10 - h00) + ty * (h11 - h01)) * c.hf_inv_hscale;
    float dhdy = ((1 - tx) * (h01 - h00) + tx * (h11 - h10)) * c.hf_inv_hscale;
    float inv = rsqrtf(dhdx * dhdx + dhdy * dhdy + 1.0f);
    return {h, {-dhdx * inv, -dhdy * inv, inv}};
}



#define LG_CT_NF 19      // floats per contact-slot record: Pc 3 (sphere centre before detection), n 3, W 6, target, impulses 3, first tangent 3 (pair-lane physics)
#define LG_LK_NF 24      // floats per link record: R 9, p 3, vel 6, c 6
__device__ __forceinline__ void tangents(V3 n, V3 &t1, V3 &t2) {
    V3 ref = fabsf(n.x) < 0.9f ? V3{1.f, 0.f, 0.f} : V3{0.f, 1.f, 0.f};
    V3 t = cross(n, ref);
    t1 = rsqrtf(dot(t, t)) * t;
    t2 = cross(n, t1);
}

// One physics step of length dt for lane (env, leg).  State in/out through registers:
//   root[13] (world: pos, quat xyzw, lin vel, ang vel), q[J], qd[J] of this leg's joints.
// fslot[s] / fbase receive the world-frame contact force (N) of this lane's sphere slots.
template <int L, int J>
__device__ __forceinline__ int physics_lane(const DevParams *__restrict__ P, int leg, float dt, float *root, float *q,
                                             float *qd, const float *tau, float friction, float dmass,
                                             const float *__restrict__ mat /* LDS: this env's restitution, compliance, thickness */,
                                             V3 *fslot, V3 &fbase, float *__restrict__ cst, float *__restrict__ lkt,
                                             const float *__restrict__ ltab, float *__restrict__ lmt) {
    const lg_cfg &c = P->cfg;
    const lg_model &m = P->model;
    const float *__restrict__ lt = ltab + leg * LG_LT_STRIDE;       // this leg's constants (LDS)
    const M3 Rb = quat_to_mat(root + 3);
    const V3 xw = {root[0], root[1], root[2]};
    const V3 vb = mulT(Rb, V3{root[7], root[8], root[9]}), wb = mulT(Rb, V3{root[10], root[11], root[12]});
    const V3 gb = mulT(Rb, V3{c.gravity[0], c.gravity[1], c.gravity[2]});
    const Sv vel0 = {wb, vb};

    // Per-link tile staged in LDS (field-major, one column per lane): rotation, origin, velocity and
    // velocity-product term of every link of this leg.  Only S, U, 1/D, u stay in VGPRs across phases.
    const int lane = threadIdx.x & 63;
#define LK(j, f) lkt[((j) * LG_LK_NF + (f)) * 64 + lane]
    Sv S[J], U[J];
    float iD[J], u[J];
    const float inv_dt = frcp(dt);
    {   // outward kinematics
        M3 Rpar = {{{1.f, 0.f, 0.f}, {0.f, 1.f, 0.f}, {0.f, 0.f, 1.f}}};
        V3 ppar = {0.f, 0.f, 0.f};
        Sv vpar = vel0;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const float *jt = lt + LG_LT_JOINT * j;
            M3 Rj = mul(Rpar, load3(jt));
            V3 ax = ld3(jt + 12);
            V3 pj = ppar + mul(Rpar, ld3(jt + 9));
            V3 axb = mul(Rj, ax);
            M3 Rlj = mul(Rj, rodrigues(ax, q[j]));
            S[j] = {axb, cross(pj, axb)};
            Sv vj = qd[j] * S[j];
            Sv velj = vpar + vj;
            Sv cbj = crm(velj, vj);
#pragma unroll
            for (int e = 0; e < 9; ++e) LK(j, e) = Rlj.m[e / 3][e % 3];
            LK(j, 9) = pj.x; LK(j, 10) = pj.y; LK(j, 11) = pj.z;
            LK(j, 12) = velj.w.x; LK(j, 13) = velj.w.y; LK(j, 14) = velj.w.z;
            LK(j, 15) = velj.v.x; LK(j, 16) = velj.v.y; LK(j, 17) = velj.v.z;
            LK(j, 18) = cbj.w.x; LK(j, 19) = cbj.w.y; LK(j, 20) = cbj.w.z;
            LK(j, 21) = cbj.v.x; LK(j, 22) = cbj.v.y; LK(j, 23) = cbj.v.z;
            Rpar = Rlj; ppar = pj; vpar = velj;
        }
    }
    // inward pass along the chain
    M6 Ia_run;
    Sv pa_run = sv_zero();
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b = 0; b < 6; ++b) Ia_run.m[a][b] = 0.f;
#pragma unroll
    for (int j = J - 1; j >= 0; --j) {
        const float *jt = lt + LG_LT_JOINT * j;
        M3 Rlj;
#pragma unroll
        for (int e = 0; e < 9; ++e) Rlj.m[e / 3][e % 3] = LK(j, e);
        const V3 pj = {LK(j, 9), LK(j, 10), LK(j, 11)};
        const Sv velj = {{LK(j, 12), LK(j, 13), LK(j, 14)}, {LK(j, 15), LK(j, 16), LK(j, 17)}};
        const Sv cbj = {{LK(j, 18), LK(j, 19), LK(j, 20)}, {LK(j, 21), LK(j, 22), LK(j, 23)}};
        M3 Ic = mulBT(mul(Rlj, load3(jt + 15)), Rlj);
        M6 IA = rigid_inertia(jt[27], pj + mul(Rlj, ld3(jt + 24)), Ic);
        Sv pA = crf(velj, mul6(IA, velj));
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int b = 0; b < 6; ++b) IA.m[a][b] += Ia_run.m[a][b];
        pA = pA + pa_run;
        U[j] = mul6(IA, S[j]);
        const float Dj = sdot(S[j], U[j]) + c.armature;
        u[j] = (tau[j] - jt[28] * qd[j]) - sdot(S[j], pA);
        float Uv[6] = {U[j].w.x, U[j].w.y, U[j].w.z, U[j].v.x, U[j].v.y, U[j].v.z};
        const float invD = frcp(Dj);
        iD[j] = invD;
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int b = 0; b < 6; ++b) Ia_run.m[a][b] = IA.m[a][b] - Uv[a] * Uv[b] * invD;
        pa_run = pA + mul6(Ia_run, cbj) + (u[j] * invD) * U[j];
    }
    // floating base: own inertia (every lane redundantly) + butterfly sum of the L leg contributions
    M6 I0;
    {
        float mb = m.mass[0] + dmass;
        I0 = rigid_inertia(mb, ld3(m.com[0]), load3(m.inertia[0]));
    }
    Sv pA0 = crf(vel0, mul6(I0, vel0)) + leg_sum<L>(pa_run);
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b = a; b < 6; ++b) {
            float s = leg_sum<L>(Ia_run.m[a][b]);
            I0.m[a][b] += s;
            if (b != a) I0.m[b][a] += s;
        }
    M6 I0inv;
    const bool ok = spd_inverse6(I0, I0inv);
    const Sv a0 = -1.0f * mul6(I0inv, pA0);
    // outward accelerations -> free velocities
    Sv velf[J], velf0;
    float qdf[J];
    {
        const Sv grav = {{0.f, 0.f, 0.f}, gb};
        velf0 = vel0 + dt * (a0 + grav);
        Sv apar = a0;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const Sv velj = {{LK(j, 12), LK(j, 13), LK(j, 14)}, {LK(j, 15), LK(j, 16), LK(j, 17)}};
            const Sv cbj = {{LK(j, 18), LK(j, 19), LK(j, 20)}, {LK(j, 21), LK(j, 22), LK(j, 23)}};
            Sv ap = apar + cbj;
            float qdd = (u[j] - sdot(U[j], ap)) * iD[j];
            Sv acc = ap + qdd * S[j];
            velf[j] = velj + dt * (acc + grav);
            qdf[j] = qd[j] + dt * qdd;
            apar = acc;
        }
    }

    // ---- contact detection + W per slot.  Slot records live in this wave's LDS region (one column
    // per lane, field-major: conflict-free) so the slot loops stay rolled and the VGPR file is left
    // to the articulated-body quantities.
    const float mu = 0.5f * (friction + c.ground_friction);
    const int nslots = P->n_leg_slots;
#define CF(si, f) cst[((si) * LG_CT_NF + (f)) * 64 + lane]
    unsigned amask = 0u;                       // bit si: slot si of this lane is in contact
    // sphere -> link table packed 4 bits per slot (wave-uniform scalar loads); the base slot maps to
    // link -1 = "no joint between the contact and the base"
    const unsigned long long link_pk = P->slot_link_pk;
    // (1) detection over every slot: geometry only
    const int nbase_it = (P->n_base_spheres + L - 1) / L;       // base spheres per lane this robot needs (wave-uniform)
    for (int s = 0; s < nslots + nbase_it; ++s) {
        const bool is_base = s >= nslots;
        const int ub = s - nslots;                              // which of this lane's base spheres
        const int si = is_base ? LG_MAX_LEG_SLOTS + ub : s;
        const bool exists = is_base ? (leg + ub * L < P->n_base_spheres) : true;
        V3 cbk = {0.f, 0.f, 0.f}, Pc = {0.f, 0.f, 0.f}, nb = {0.f, 0.f, 1.f};
        float rad = 0.f, vtarget = 0.f;
        bool active = false;
        if (exists) {
            if (is_base) {
                cbk = ld3(lt + LG_LT_BASE + 4 * ub);
                rad = lt[LG_LT_BASE + 4 * ub + 3];
            } else {
                const int jl = (int)((link_pk >> (4 * s)) & 15ull);
                M3 Rk;
#pragma unroll
                for (int e = 0; e < 9; ++e) Rk.m[e / 3][e % 3] = LK(jl, e);
                const V3 pk = {LK(jl, 9), LK(jl, 10), LK(jl, 11)};
                cbk = pk + mul(Rk, ld3(lt + LG_LT_SLOTS + 4 * s));
                rad = lt[LG_LT_SLOTS + 4 * s + 3];
            }
            V3 cw = xw + mul(Rb, cbk);
            Ground g = ground_at(P->cfg, P->height_samples, cw.x, cw.y);
            float gap = (cw.z - g.h) * g.n.z - rad;
            gap -= c.material_rand ? mat[2] : c.rest_offset;    // shape thickness (asset option, or the env's draw): the robot rests that far off the surface
            if (gap < c.contact_offset) {
                active = true;
                nb = mulT(Rb, g.n);
                Pc = cbk - rad * nb;
                vtarget = gap >= 0.0f ? -gap * inv_dt : fminf(-gap * c.contact_erp * inv_dt, c.max_depenetration_velocity);
            }
        }
        if (active) {                                   // records of inactive slots are never read for a result
            CF(si, 0) = Pc.x; CF(si, 1) = Pc.y; CF(si, 2) = Pc.z;
            CF(si, 3) = nb.x; CF(si, 4) = nb.y; CF(si, 5) = nb.z;
            CF(si, 12) = vtarget;
            CF(si, 13) = 0.f; CF(si, 14) = 0.f; CF(si, 15) = 0.f;
            amask |= 1u << si;
        }
    }
    // (2) W per ACTIVE slot: every lane walks its own list of set bits, so the trip count is the largest
    // number of simultaneous contacts of any lane of the wave, not the number of slots in contact
    // anywhere in it.  Per-lane order stays ascending in si (the base slot last).
    for (unsigned rem = amask; __any(rem != 0u); rem &= rem - 1u) {
        const bool valid = rem != 0u;
        const int si = valid ? __ffs(rem) - 1 : 0;
        const int jl = (si >= LG_MAX_LEG_SLOTS) ? -1 : (int)((link_pk >> (4 * si)) & 15ull);
        const V3 Pc = {CF(si, 0), CF(si, 1), CF(si, 2)}, nb = {CF(si, 3), CF(si, 4), CF(si, 5)};
        V3 t1, t2;
        tangents(nb, t1, t2);
        V3 dirs[3] = {nb, t1, t2};
        float Wc[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            Sv pAi = {-1.0f * cross(Pc, dirs[a]), -1.0f * dirs[a]};
            float ui[J];
#pragma unroll
            for (int k = J - 1; k >= 0; --k) {
                if (k <= jl) {
                    ui[k] = -sdot(S[k], pAi);
                    pAi = pAi + (ui[k] * iD[k]) * U[k];
                } else ui[k] = 0.f;
            }
            Sv dv = -1.0f * mul6(I0inv, pAi);
#pragma unroll
            for (int k = 0; k < J; ++k)
                if (k <= jl) {
                    float dq = (ui[k] - sdot(U[k], dv)) * iD[k];
                    dv = dv + dq * S[k];
                }
            V3 dvP = dv.v + cross(dv.w, Pc);
#pragma unroll
            for (int b = 0; b < 3; ++b) Wc[b][a] = dot(dirs[b], dvP);
        }
        if (valid) {
            if (c.material_rand) Wc[0][0] += mat[1] * inv_dt * inv_dt;      // compliance (m/N) as constraint-force mixing on the normal row
            CF(si, 6) = Wc[0][0] > 1e-9f ? frcp(Wc[0][0]) : 0.f; CF(si, 7) = Wc[1][0]; CF(si, 8) = Wc[2][0];
            CF(si, 9) = Wc[1][1] > 1e-9f ? frcp(Wc[1][1]) : 0.f; CF(si, 10) = Wc[2][1];
            CF(si, 11) = Wc[2][2] > 1e-9f ? frcp(Wc[2][2]) : 0.f;
        }
    }
    const int n_base_active = (int)leg_sum<L>((float)__popc(amask >> LG_MAX_LEG_SLOTS));
    const int n_leg_active = __popc(amask & ((1u << LG_MAX_LEG_SLOTS) - 1u));
    const float rl = frcp((float)max(n_leg_active, 1)), rb = frcp((float)max(n_base_active, 1));

    // ---- joint position limits (URDF lower / upper, equal = none) as unilateral constraints on the joint rate, active
    // when the free motion would carry the joint past its stop within this step; W = response of the joint rate to a
    // unit joint impulse through the same factors; records (sign, target rate, 1/W, impulse) in LDS columns
#define LM(j, f) lmt[((j) * 4 + (f)) * 64 + lane]
    unsigned lmask = 0u;
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const float lo = lt[LG_LT_JOINT * j + 30], hi = lt[LG_LT_JOINT * j + 31], vlim = lt[LG_LT_JOINT * j + 29];
        float sgn = 0.f, gap = 0.f;
        if (hi > lo) {
            const float qn = q[j] + dt * qdf[j];
            if (qn > hi) { sgn = 1.0f; gap = hi - q[j]; }
            else if (qn < lo) { sgn = -1.0f; gap = q[j] - lo; }
        }
        // the joint velocity limit rides in the same row (see lg_physics_pair.h): a joint-space impulse conserves momentum
        float vtarget = gap >= 0.0f ? -gap * inv_dt : fminf(-gap * c.contact_erp * inv_dt, c.max_depenetration_velocity);
        if (vlim > 0.0f) {
            if (sgn == 0.f) {
                if (qdf[j] > vlim) { sgn = 1.0f; vtarget = -vlim; }
                else if (qdf[j] < -vlim) { sgn = -1.0f; vtarget = -vlim; }
            } else vtarget = fmaxf(vtarget, -vlim);
        }
        const bool act = sgn != 0.f;
        if (!__any(act)) continue;                                   // wave-uniform: the usual case
        Sv pAi = sv_zero();
        float ui[J];
#pragma unroll
        for (int k = J - 1; k >= 0; --k) {
            if (k > j) ui[k] = 0.f;
            else {
                ui[k] = (k == j ? 1.0f : 0.f) - sdot(S[k], pAi);
                pAi = pAi + (ui[k] * iD[k]) * U[k];
            }
        }
        Sv dv = -1.0f * mul6(I0inv, pAi);
        float Wj = 0.f;
#pragma unroll
        for (int k = 0; k < J; ++k)
            if (k <= j) {
                const float dq = (ui[k] - sdot(U[k], dv)) * iD[k];
                dv = dv + dq * S[k];
                if (k == j) Wj = dq;
            }
        if (act) {
            lmask |= 1u << j;
            LM(j, 0) = sgn;
            LM(j, 1) = vtarget;
            LM(j, 2) = Wj > 1e-9f ? frcp(Wj) : 0.f;
            LM(j, 3) = 0.f;
        }
    }
    const float rlim = frcp((float)max(__popc(lmask), 1));

    // ---- projected Jacobi sweeps (wave-uniform trip counts; waves without contacts or active limits skip them)
    if (__any((amask | lmask) != 0u)) {
        for (int it = 0; it < c.solver_iterations; ++it) {
            Sv fimp[J], fb = sv_zero();
#pragma unroll
            for (int k = 0; k < J; ++k) fimp[k] = sv_zero();
            for (unsigned rem = amask; __any(rem != 0u); rem &= rem - 1u) {
                const bool active = rem != 0u;
                const int si = active ? __ffs(rem) - 1 : 0;
                const bool is_base = si >= LG_MAX_LEG_SLOTS;
                const int jl = is_base ? -1 : (int)((link_pk >> (4 * si)) & 15ull);
                Sv vl = velf0;
#pragma unroll
                for (int k = 0; k < J; ++k)
                    if (jl == k) vl = velf[k];
                if (active) {
                    const V3 Pc = {CF(si, 0), CF(si, 1), CF(si, 2)}, nb = {CF(si, 3), CF(si, 4), CF(si, 5)};
                    const float oln = CF(si, 13), ol1 = CF(si, 14), ol2 = CF(si, 15), relax = is_base ? rb : rl;
                    V3 t1, t2;
                    tangents(nb, t1, t2);
                    V3 vP = vl.v + cross(vl.w, Pc);
                    float vc0 = dot(nb, vP), vc1 = dot(t1, vP), vc2 = dot(t2, vP);
                    if (c.material_rand && it == 0 && vc0 < -c.bounce_threshold)      // restitution: leave with e x the approach speed
                        CF(si, 12) = fmaxf(CF(si, 12), -0.5f * (mat[0] + c.ground_restitution) * vc0);
                    float ln = fmaxf(0.0f, oln - relax * (vc0 - CF(si, 12)) * CF(si, 6));
                    float dn = ln - oln;
                    vc1 += CF(si, 7) * dn;
                    vc2 += CF(si, 8) * dn;
                    float l1 = ol1 - relax * vc1 * CF(si, 9);
                    vc2 += CF(si, 10) * (l1 - ol1);
                    float l2 = ol2 - relax * vc2 * CF(si, 11);
                    float lim = mu * ln, mag = sqrtf(l1 * l1 + l2 * l2);
                    if (mag > lim) { float sc = lim * frcp(fmaxf(mag, 1e-12f)); l1 *= sc; l2 *= sc; }
                    V3 dl = (ln - oln) * nb + (l1 - ol1) * t1 + (l2 - ol2) * t2;
                    CF(si, 13) = ln; CF(si, 14) = l1; CF(si, 15) = l2;
                    Sv f = {cross(Pc, dl), dl};
                    if (is_base) fb = fb + f;
#pragma unroll
                    for (int k = 0; k < J; ++k)
                        if (jl == k) fimp[k] = fimp[k] + f;
                }
            }
            float ui[J], timp[J];
#pragma unroll
            for (int k = 0; k < J; ++k) {
                timp[k] = 0.f;
                if ((lmask >> k) & 1u) {
                    const float sgn = LM(k, 0), old = LM(k, 3);
                    const float ln = fmaxf(0.0f, old - rlim * (-sgn * qdf[k] - LM(k, 1)) * LM(k, 2));
                    timp[k] = -sgn * (ln - old);
                    LM(k, 3) = ln;
                }
            }
            Sv run = sv_zero();
#pragma unroll
            for (int k = J - 1; k >= 0; --k) {
                Sv cur = run - fimp[k];
                ui[k] = timp[k] - sdot(S[k], cur);
                run = cur + (ui[k] * iD[k]) * U[k];
            }
            Sv pAi0 = leg_sum<L>(run - fb);
            Sv dv = -1.0f * mul6(I0inv, pAi0);
            velf0 = velf0 + dv;
#pragma unroll
            for (int k = 0; k < J; ++k) {
                float dq = (ui[k] - sdot(U[k], dv)) * iD[k];
                dv = dv + dq * S[k];
                velf[k] = velf[k] + dv;
                qdf[k] += dq;
            }
        }
    }

    // ---- contact forces out (world frame, N)
    fbase = {0.f, 0.f, 0.f};
#pragma unroll
    for (int si = 0; si < LG_NUM_SLOTS; ++si) {
        V3 f = {0.f, 0.f, 0.f};
        if ((amask >> si) & 1u) {
            const V3 nb = {CF(si, 3), CF(si, 4), CF(si, 5)};
            V3 t1, t2;
            tangents(nb, t1, t2);
            f = inv_dt * mul(Rb, CF(si, 13) * nb + CF(si, 14) * t1 + CF(si, 15) * t2);
        }
        if (si >= LG_MAX_LEG_SLOTS) fbase = fbase + f; else fslot[si] = f;
    }
#undef CF
#undef LK
#undef LM
    // ---- what PhysX does with a runaway body (asset options max_linear_velocity / max_angular_velocity, legged_robot.py:701-702):
    // it clamps the velocity and carries on.  The guard is for non-finite state only (PhysX never hands back NaN / Inf; neither
    // may we): such an env keeps its pose, is brought to rest and is reported so that the post-step terminates and resets it.
    float chk = dot(velf0.w, velf0.w) + dot(velf0.v, velf0.v);
#pragma unroll
    for (int j = 0; j < J; ++j) chk += qdf[j] * qdf[j] * 1e-4f;
    chk = leg_sum<L>(chk);
    if (!ok || !(chk < 3.0e38f)) {
#pragma unroll
        for (int j = 0; j < J; ++j) qd[j] = 0.f;
#pragma unroll
        for (int k = 7; k < 13; ++k) root[k] = 0.f;
        return 1;
    }
    int code = 0;
    // ---- integrate
#pragma unroll
    for (int j = 0; j < J; ++j) {
        float v = qdf[j];
        {   // hard stop behind the limit constraints: never end the step further out than max(limit, start)
            const float lo = lt[LG_LT_JOINT * j + 30], hi = lt[LG_LT_JOINT * j + 31];
            if (hi > lo) {
                const float qn = fminf(fmaxf(q[j] + dt * v, fminf(lo, q[j])), fmaxf(hi, q[j]));
                v = (qn - q[j]) * inv_dt;
            }
        }
        float vl = lt[LG_LT_JOINT * j + 29];
        if (vl > 0.0f) v = fminf(fmaxf(v, -vl), vl);
        qd[j] = v;
        q[j] += dt * v;
    }
    V3 wn = velf0.w;
    V3 vn = velf0.v + dt * cross(wb, vb);
    {   // the base's velocities as they are published, clamped at the asset's maxima
        const float w2 = dot(wn, wn), v2 = dot(vn, vn);
        if (c.max_angular_velocity > 0.0f && w2 > c.max_angular_velocity * c.max_angular_velocity) { wn = (c.max_angular_velocity * rsqrtf(w2)) * wn; code = 2; }
        if (c.max_linear_velocity > 0.0f && v2 > c.max_linear_velocity * c.max_linear_velocity) { vn = (c.max_linear_velocity * rsqrtf(v2)) * vn; code = 2; }
    }
    V3 vw = mul(Rb, vn), ww = mul(Rb, wn);
    root[0] += dt * vw.x; root[1] += dt * vw.y; root[2] += dt * vw.z;
    root[7] = vw.x; root[8] = vw.y; root[9] = vw.z;
    root[10] = ww.x; root[11] = ww.y; root[12] = ww.z;
    float ang = sqrtf(dot(wn, wn)) * dt;
    float sh, ch = cosf(0.5f * ang);
    V3 ax;
    if (ang > 1e-8f) { sh = sinf(0.5f * ang); ax = (dt / ang) * wn; } else { sh = 0.5f * dt; ax = wn; }
    float dq[4] = {sh * ax.x, sh * ax.y, sh * ax.z, ch};
    float *qq = root + 3;
    float qn[4] = {qq[3] * dq[0] + qq[0] * dq[3] + qq[1] * dq[2] - qq[2] * dq[1],
                   qq[3] * dq[1] - qq[0] * dq[2] + qq[1] * dq[3] + qq[2] * dq[0],
                   qq[3] * dq[2] + qq[0] * dq[1] - qq[1] * dq[0] + qq[2] * dq[3],
                   qq[3] * dq[3] - qq[0] * dq[0] - qq[1] * dq[1] - qq[2] * dq[2]};
    float nrm = rsqrtf(qn[0] * qn[0] + qn[1] * qn[1] + qn[2] * qn[2] + qn[3] * qn[3]);
#pragma unroll
    for (int k = 0; k < 4; ++k) qq[k] = qn[k] * nrm;
    return code;
}
