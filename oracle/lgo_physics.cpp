// CPU ORACLE -- TEST INFRASTRUCTURE ONLY (see lgo_common.h).
//
// One sim_dt of articulated rigid-body physics: the step the reference delegates to Isaac Gym /
// PhysX through gym.simulate (legged_robot.py:92-96, solver settings legged_robot_config.py:216-233).
// PhysX is closed source and absent: THIS IS THE BUILD'S OWN SPECIFICATION OF THE
// PHYSICS, PARITY WITH PHYSX IS UNPINNED.  It is validated by invariants (tests/test_physics_oracle.py:
// float64 mass-matrix dynamics, momentum, static stance) and is what the HIP kernel must match.
//
// Algorithm (scalar, generic tree given by parent_dof; the HIP kernel is lane-parallel per leg):
//   1. forward kinematics + Featherstone articulated-body algorithm with all spatial quantities
//      expressed in base-frame coordinates about the base origin (no per-joint Plücker transforms);
//      gravity enters as a uniform spatial acceleration added after the base solve;
//   2. free velocities v* = v + dt a;
//   3. sphere-vs-ground contacts (plane or bilinear height samples), active when gap < contact_offset;
//      per contact the exact 3x3 contact-space inverse inertia W = J M^-1 J^T from three test impulses
//      propagated through the articulated-body factors of step 1;
//   4. `solver_iterations` sweeps of projected Jacobi over contacts (Gauss-Seidel inside a contact:
//      normal row, then the two friction rows; Coulomb disc |lt| <= mu ln; speculative margin for
//      open gaps, ERP-limited depenetration capped at max_depenetration_velocity), each sweep followed
//      by ONE impulse propagation through the tree; relaxation 1/(active contacts in the same chain);
//   4b. URDF joint position limits ride in the same sweeps as unilateral constraints on the joint rate (active when the
//      free motion would pass the stop within the step; W from a unit joint impulse; relaxation 1/(active limits in the
//      chain)), with a hard stop at integration: a step never ends further out than max(limit, start);
//   5. semi-implicit Euler: positions advance with the post-contact velocities; net contact force per
//      body = sum of its spheres' impulses / dt, world frame.
#include "lgo_common.h"

#include <algorithm>

namespace lgo {

struct V3 { float x, y, z; };
static inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
static inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
struct M3 { float m[3][3]; };
static inline V3 mul(const M3 &A, V3 v) {
    return {A.m[0][0] * v.x + A.m[0][1] * v.y + A.m[0][2] * v.z, A.m[1][0] * v.x + A.m[1][1] * v.y + A.m[1][2] * v.z,
            A.m[2][0] * v.x + A.m[2][1] * v.y + A.m[2][2] * v.z};
}
static inline V3 mulT(const M3 &A, V3 v) {
    return {A.m[0][0] * v.x + A.m[1][0] * v.y + A.m[2][0] * v.z, A.m[0][1] * v.x + A.m[1][1] * v.y + A.m[2][1] * v.z,
            A.m[0][2] * v.x + A.m[1][2] * v.y + A.m[2][2] * v.z};
}
static inline M3 mul(const M3 &A, const M3 &B) {
    M3 C;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) C.m[i][j] = A.m[i][0] * B.m[0][j] + A.m[i][1] * B.m[1][j] + A.m[i][2] * B.m[2][j];
    return C;
}
static inline M3 load3(const float *p) { M3 A; std::memcpy(A.m, p, 36); return A; }
static inline M3 rodrigues(V3 a, float th) {
    float c = std::cos(th), s = std::sin(th), t = 1.0f - c;
    M3 R = {{{c + t * a.x * a.x, t * a.x * a.y - s * a.z, t * a.x * a.z + s * a.y},
             {t * a.x * a.y + s * a.z, c + t * a.y * a.y, t * a.y * a.z - s * a.x},
             {t * a.x * a.z - s * a.y, t * a.y * a.z + s * a.x, c + t * a.z * a.z}}};
    return R;
}
static inline M3 quat_to_mat(const float *q) {   // xyzw, body -> world
    float x = q[0], y = q[1], z = q[2], w = q[3];
    M3 R = {{{1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)},
             {2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)},
             {2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)}}};
    return R;
}

struct Sv { V3 w, v; };                             // spatial motion (w = angular) / force (w = moment)
static inline Sv operator+(Sv a, Sv b) { return {a.w + b.w, a.v + b.v}; }
static inline Sv operator*(float s, Sv a) { return {s * a.w, s * a.v}; }
static inline float sdot(Sv a, Sv b) { return dot(a.w, b.w) + dot(a.v, b.v); }
static inline Sv crm(Sv a, Sv b) { return {cross(a.w, b.w), cross(a.w, b.v) + cross(a.v, b.w)}; }   // motion x motion
static inline Sv crf(Sv a, Sv f) { return {cross(a.w, f.w) + cross(a.v, f.v), cross(a.w, f.v)}; }   // motion x* force
struct M6 { float m[6][6]; };
static inline Sv mul(const M6 &I, Sv a) {
    float x[6] = {a.w.x, a.w.y, a.w.z, a.v.x, a.v.y, a.v.z}, y[6];
    for (int i = 0; i < 6; ++i) { float s = 0; for (int j = 0; j < 6; ++j) s += I.m[i][j] * x[j]; y[i] = s; }
    return {{y[0], y[1], y[2]}, {y[3], y[4], y[5]}};
}
// rigid-body spatial inertia about the origin: mass m, com c, rotational inertia Ic about com (all base coords)
static M6 rigid_inertia(float m, V3 c, const M3 &Ic) {
    M6 I;
    float cc = dot(c, c);
    float cv[3] = {c.x, c.y, c.z};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            I.m[i][j] = Ic.m[i][j] + m * ((i == j ? cc : 0.0f) - cv[i] * cv[j]);
            I.m[3 + i][3 + j] = (i == j) ? m : 0.0f;
        }
    float hx[3][3] = {{0, -m * c.z, m * c.y}, {m * c.z, 0, -m * c.x}, {-m * c.y, m * c.x, 0}};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) { I.m[i][3 + j] = hx[i][j]; I.m[3 + i][j] = hx[j][i]; }
    return I;
}
// inverse of a symmetric positive definite 6x6 (Cholesky)
static bool spd_inverse6(const M6 &A, M6 &Ainv) {
    float L[6][6] = {};
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j <= i; ++j) {
            float s = A.m[i][j];
            for (int k = 0; k < j; ++k) s -= L[i][k] * L[j][k];
            if (i == j) { if (s <= 0.0f) return false; L[i][i] = std::sqrt(s); }
            else L[i][j] = s / L[j][j];
        }
    for (int c = 0; c < 6; ++c) {
        float y[6], x[6];
        for (int i = 0; i < 6; ++i) { float s = (i == c) ? 1.0f : 0.0f; for (int k = 0; k < i; ++k) s -= L[i][k] * y[k]; y[i] = s / L[i][i]; }
        for (int i = 5; i >= 0; --i) { float s = y[i]; for (int k = i + 1; k < 6; ++k) s -= L[k][i] * x[k]; x[i] = s / L[i][i]; }
        for (int i = 0; i < 6; ++i) Ainv.m[i][c] = x[i];
    }
    return true;
}

struct Ground { float h; V3 n; };
static Ground ground_at(const Env &e, float x, float y) {
    const lg_cfg &c = e.cfg;
    if (c.terrain_type == 0) return {0.0f, {0.0f, 0.0f, 1.0f}};
    const float inv_h = 1.0f / c.hf_hscale;      // grid coordinates and slopes as products with the reciprocal pitch (what the HIP path computes)
    float gx = (x + c.border_size) * inv_h, gy = (y + c.border_size) * inv_h;
    gx = std::min(std::max(gx, 0.0f), (float)(c.hf_rows - 1) - 1e-3f);
    gy = std::min(std::max(gy, 0.0f), (float)(c.hf_cols - 1) - 1e-3f);
    int ix = (int)gx, iy = (int)gy;
    float tx = gx - ix, ty = gy - iy;
    auto H = [&](int a, int b) { return (float)e.height_samples[(size_t)a * c.hf_cols + b] * c.hf_vscale; };
    float h00 = H(ix, iy), h10 = H(ix + 1, iy), h01 = H(ix, iy + 1), h11 = H(ix + 1, iy + 1);
    float h = (1 - tx) * (1 - ty) * h00 + tx * (1 - ty) * h10 + (1 - tx) * ty * h01 + tx * ty * h11;
    float dhdx = ((1 - ty) * (h10 - h00) + ty * (h11 - h01)) * inv_h;
    float dhdy = ((1 - tx) * (h01 - h00) + tx * (h11 - h10)) * inv_h;
    float inv = 1.0f / std::sqrt(dhdx * dhdx + dhdy * dhdy + 1.0f);
    return {h, {-dhdx * inv, -dhdy * inv, inv}};
}

struct Contact {
    int link, body, group;
    V3 P, n, t1, t2;           // base coords
    float W[3][3];             // contact frame (n,t1,t2)
    float vtarget, mu, relax;
    float lam[3], iW[3];
};

static int simulate_env(Env &e, int i, float dt, float *cf_accum, float cf_weight) {
    const lg_cfg &c = e.cfg;
    const lg_model &m = e.model;
    const int A = e.A, NL = A + 1;
    float *root = &e.root[(size_t)i * 13];
    float *dofs = &e.dof[(size_t)i * A * 2];
    const float *tau = &e.torques[(size_t)i * A];

    M3 Rb = quat_to_mat(root + 3);
    V3 xw = {root[0], root[1], root[2]};
    V3 vb = mulT(Rb, V3{root[7], root[8], root[9]}), wb = mulT(Rb, V3{root[10], root[11], root[12]});
    V3 gb = mulT(Rb, V3{c.gravity[0], c.gravity[1], c.gravity[2]});

    M3 R[LG_MAX_DOF + 1];
    V3 p[LG_MAX_DOF + 1];
    Sv S[LG_MAX_DOF], vel[LG_MAX_DOF + 1], cb[LG_MAX_DOF], pA[LG_MAX_DOF + 1], U[LG_MAX_DOF], acc[LG_MAX_DOF + 1];
    M6 IA[LG_MAX_DOF + 1];
    float D[LG_MAX_DOF], u[LG_MAX_DOF], qdd[LG_MAX_DOF];
    int par[LG_MAX_DOF];

    R[0] = {{{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}};
    p[0] = {0, 0, 0};
    vel[0] = {wb, vb};
    {   // base link (+ randomised payload as a point mass at the base com, legged_robot.py:332-334)
        float mb = m.mass[0] + e.base_mass_delta[i];
        V3 cb0 = {m.com[0][0], m.com[0][1], m.com[0][2]};
        IA[0] = rigid_inertia(mb, cb0, load3(m.inertia[0]));
        pA[0] = crf(vel[0], mul(IA[0], vel[0]));
    }
    for (int d = 0; d < A; ++d) {
        int pl = par[d] = (d % m.joints_per_leg == 0) ? 0 : d;         // parent LINK index (chain topology)
        float q = dofs[2 * d], qd = dofs[2 * d + 1];
        M3 Rj = mul(R[pl], load3(m.R_pj[d]));
        V3 ax = {m.axis[d][0], m.axis[d][1], m.axis[d][2]};
        p[d + 1] = p[pl] + mul(R[pl], V3{m.p_pj[d][0], m.p_pj[d][1], m.p_pj[d][2]});
        V3 axb = mul(Rj, ax);
        R[d + 1] = mul(Rj, rodrigues(ax, q));
        S[d] = {axb, cross(p[d + 1], axb)};
        Sv vj = qd * S[d];
        vel[d + 1] = vel[pl] + vj;
        cb[d] = crm(vel[d + 1], vj);
        V3 cl = {m.com[d + 1][0], m.com[d + 1][1], m.com[d + 1][2]};
        M3 Ic = mul(mul(R[d + 1], load3(m.inertia[d + 1])), M3{{{R[d + 1].m[0][0], R[d + 1].m[1][0], R[d + 1].m[2][0]},
                                                                 {R[d + 1].m[0][1], R[d + 1].m[1][1], R[d + 1].m[2][1]},
                                                                 {R[d + 1].m[0][2], R[d + 1].m[1][2], R[d + 1].m[2][2]}}});
        IA[d + 1] = rigid_inertia(m.mass[d + 1], p[d + 1] + mul(R[d + 1], cl), Ic);
        pA[d + 1] = crf(vel[d + 1], mul(IA[d + 1], vel[d + 1]));
    }
    for (int d = A - 1; d >= 0; --d) {                                  // inward pass
        int l = d + 1, pl = par[d];
        U[d] = mul(IA[l], S[d]);
        D[d] = sdot(S[d], U[d]) + c.armature;                            // asset.armature (legged_robot.py:703)
        u[d] = (tau[d] - m.joint_damping[d] * dofs[2 * d + 1]) - sdot(S[d], pA[l]);
        float Uv[6] = {U[d].w.x, U[d].w.y, U[d].w.z, U[d].v.x, U[d].v.y, U[d].v.z};
        M6 Ia;
        float invD = 1.0f / D[d];
        for (int a = 0; a < 6; ++a)
            for (int b = 0; b < 6; ++b) Ia.m[a][b] = IA[l].m[a][b] - Uv[a] * Uv[b] * invD;
        Sv pa = pA[l] + mul(Ia, cb[d]) + (u[d] * invD) * U[d];
        for (int a = 0; a < 6; ++a)
            for (int b = 0; b < 6; ++b) IA[pl].m[a][b] += Ia.m[a][b];
        pA[pl] = pA[pl] + pa;
    }
    M6 I0inv;
    const bool ok = spd_inverse6(IA[0], I0inv);
    if (!ok) { for (int a = 0; a < 6; ++a) for (int b = 0; b < 6; ++b) I0inv.m[a][b] = 0.f; }
    acc[0] = -1.0f * mul(I0inv, pA[0]);
    for (int d = 0; d < A; ++d) {                                       // outward pass
        Sv ap = acc[par[d]] + cb[d];
        qdd[d] = (u[d] - sdot(U[d], ap)) / D[d];
        acc[d + 1] = ap + qdd[d] * S[d];
    }
    // free velocities (gravity = uniform spatial acceleration [0; g_b] on every link)
    Sv velf[LG_MAX_DOF + 1];
    float qdf[LG_MAX_DOF];
    for (int l = 0; l < NL; ++l) velf[l] = vel[l] + dt * (acc[l] + Sv{{0, 0, 0}, gb});
    for (int d = 0; d < A; ++d) qdf[d] = dofs[2 * d + 1] + dt * qdd[d];

    // impulse propagation through the articulated-body factors: fimp = spatial impulse per link
    // (timp: optional generalised impulse per joint, used by the joint-limit constraints)
    auto propagate = [&](const Sv *fimp, Sv *dvel, float *dqd, const float *timp = nullptr) {
        Sv pAi[LG_MAX_DOF + 1];
        float ui[LG_MAX_DOF];
        for (int l = 0; l < NL; ++l) pAi[l] = -1.0f * fimp[l];
        for (int d = A - 1; d >= 0; --d) {
            ui[d] = (timp ? timp[d] : 0.0f) - sdot(S[d], pAi[d + 1]);
            pAi[par[d]] = pAi[par[d]] + pAi[d + 1] + (ui[d] / D[d]) * U[d];
        }
        dvel[0] = -1.0f * mul(I0inv, pAi[0]);
        for (int d = 0; d < A; ++d) {
            dqd[d] = (ui[d] - sdot(U[d], dvel[par[d]])) / D[d];
            dvel[d + 1] = dvel[par[d]] + dqd[d] * S[d];
        }
    };

    // ---- contact detection
    Contact ct[LG_MAX_SPHERES];
    int nc = 0;
    int group_count[LG_MAX_DOF + 1] = {};
    const float mu = 0.5f * (e.friction[i] + c.ground_friction);
    const float *mat = &e.material[(size_t)i * 4];                       // restitution, compliance, thickness (lg_cfg.material_rand)
    for (int k = 0; k < m.num_spheres; ++k) {
        int l = m.sph_link[k] + 1;
        V3 cbk = p[l] + mul(R[l], V3{m.sph_center[k][0], m.sph_center[k][1], m.sph_center[k][2]});
        V3 cw = xw + mul(Rb, cbk);
        Ground g = ground_at(e, cw.x, cw.y);
        float gap = (cw.z - g.h) * g.n.z - m.sph_radius[k];
        gap -= c.material_rand ? mat[2] : c.rest_offset;                 // shape thickness (asset option, or the env's draw): the robot rests that far off the surface
        if (gap >= c.contact_offset) continue;
        Contact &C = ct[nc++];
        C.link = l;
        C.body = m.sph_body[k];
        C.group = (l == 0) ? 0 : 1 + (l - 1) / m.joints_per_leg;
        group_count[C.group]++;
        C.n = mulT(Rb, g.n);
        C.P = cbk - m.sph_radius[k] * C.n;
        V3 ref = std::fabs(C.n.x) < 0.9f ? V3{1, 0, 0} : V3{0, 1, 0};
        V3 t1 = cross(C.n, ref);
        C.t1 = (1.0f / std::sqrt(dot(t1, t1))) * t1;
        C.t2 = cross(C.n, C.t1);
        C.vtarget = gap >= 0.0f ? -gap / dt : std::min(-gap * c.contact_erp / dt, c.max_depenetration_velocity);
        C.mu = mu;
        C.lam[0] = C.lam[1] = C.lam[2] = 0.0f;
        V3 dirs[3] = {C.n, C.t1, C.t2};
        for (int a = 0; a < 3; ++a) {                                  // W column a = response to a unit impulse along dirs[a]
            Sv fimp[LG_MAX_DOF + 1] = {};
            fimp[l] = {cross(C.P, dirs[a]), dirs[a]};
            Sv dv[LG_MAX_DOF + 1];
            float dq[LG_MAX_DOF];
            propagate(fimp, dv, dq);
            V3 dvP = dv[l].v + cross(dv[l].w, C.P);
            for (int b = 0; b < 3; ++b) C.W[b][a] = dot(dirs[b], dvP);
        }
        if (c.material_rand) C.W[0][0] += mat[1] * (1.0f / dt) * (1.0f / dt);   // compliance (m/N) as constraint-force mixing on the normal row
        for (int a = 0; a < 3; ++a) C.iW[a] = C.W[a][a] > 1e-9f ? 1.0f / C.W[a][a] : 0.0f;
    }
    for (int k = 0; k < nc; ++k) ct[k].relax = 1.0f / (float)group_count[ct[k].group];

    // ---- joint position limits (URDF lower/upper; equal = none) as unilateral constraints on the joint rate:
    // active when the free motion would carry the joint past the limit within this step.  gap rate = -sgn * qd must
    // be >= vtarget (reach the limit exactly; when already beyond it, come back with the contacts' ERP and speed cap).
    // W = response of qd to a unit joint impulse through the same articulated-body factors.
    struct Limit { int d; float sgn, vtarget, iW, lam, relax; };
    Limit lm[LG_MAX_DOF];
    int nlim = 0;
    int lim_count[LG_MAX_DOF + 1] = {};                                // active limits per leg chain (Jacobi relaxation, as for contacts)
    // The joint velocity limit rides in the same rows (a bound on the joint rate enforced by a joint-space impulse: equal and opposite
    // on child and parent, so momentum is conserved -- the clamp at integration alone leaves the base the reaction of the rate it takes away).
    for (int d = 0; d < A; ++d) {
        const float lo = m.q_lower[d], hi = m.q_upper[d], vlim = m.vel_limit[d];
        const float q = dofs[2 * d];
        float sgn = 0.0f, gap = 0.0f;
        if (hi > lo) {
            if (q + dt * qdf[d] > hi) { sgn = 1.0f; gap = hi - q; }
            else if (q + dt * qdf[d] < lo) { sgn = -1.0f; gap = q - lo; }
        }
        float vtarget = gap >= 0.0f ? -gap / dt : std::min(-gap * c.contact_erp / dt, c.max_depenetration_velocity);
        if (vlim > 0.0f) {
            if (sgn == 0.0f) {
                if (qdf[d] > vlim) { sgn = 1.0f; vtarget = -vlim; }
                else if (qdf[d] < -vlim) { sgn = -1.0f; vtarget = -vlim; }
            } else vtarget = std::max(vtarget, -vlim);
        }
        if (sgn == 0.0f) continue;
        float timp[LG_MAX_DOF] = {};
        timp[d] = 1.0f;
        Sv fz[LG_MAX_DOF + 1] = {};
        Sv dv[LG_MAX_DOF + 1];
        float dq[LG_MAX_DOF];
        propagate(fz, dv, dq, timp);
        Limit &Lm = lm[nlim++];
        Lm.d = d; Lm.sgn = sgn; Lm.lam = 0.0f;
        Lm.iW = dq[d] > 1e-9f ? 1.0f / dq[d] : 0.0f;
        Lm.vtarget = vtarget;
        lim_count[d / m.joints_per_leg]++;
    }
    for (int k = 0; k < nlim; ++k) lm[k].relax = 1.0f / (float)lim_count[lm[k].d / m.joints_per_leg];

    // ---- projected Jacobi sweeps
    for (int it = 0; it < c.solver_iterations && nc + nlim > 0; ++it) {
        Sv fimp[LG_MAX_DOF + 1] = {};
        float timp[LG_MAX_DOF] = {};
        for (int k = 0; k < nlim; ++k) {
            Limit &Lm = lm[k];
            const float vc = -Lm.sgn * qdf[Lm.d];
            const float ln = std::max(0.0f, Lm.lam - Lm.relax * (vc - Lm.vtarget) * Lm.iW);
            timp[Lm.d] += -Lm.sgn * (ln - Lm.lam);
            Lm.lam = ln;
        }
        for (int k = 0; k < nc; ++k) {
            Contact &C = ct[k];
            V3 vP = velf[C.link].v + cross(velf[C.link].w, C.P);
            float vc[3] = {dot(C.n, vP), dot(C.t1, vP), dot(C.t2, vP)};
            float old[3] = {C.lam[0], C.lam[1], C.lam[2]};
            if (c.material_rand && it == 0 && vc[0] < -c.bounce_threshold)     // restitution: leave with e x the approach speed
                C.vtarget = std::max(C.vtarget, -0.5f * (mat[0] + c.ground_restitution) * vc[0]);
            float ln = std::max(0.0f, old[0] - C.relax * (vc[0] - C.vtarget) * C.iW[0]);
            float dn = ln - old[0];
            vc[1] += C.W[1][0] * dn;
            vc[2] += C.W[2][0] * dn;
            float l1 = old[1] - C.relax * vc[1] * C.iW[1];
            vc[2] += C.W[2][1] * (l1 - old[1]);
            float l2 = old[2] - C.relax * vc[2] * C.iW[2];
            float lim = C.mu * ln, mag = std::sqrt(l1 * l1 + l2 * l2);
            if (mag > lim) { float s = lim / std::max(mag, 1e-12f); l1 *= s; l2 *= s; }
            C.lam[0] = ln; C.lam[1] = l1; C.lam[2] = l2;
            V3 dl = (ln - old[0]) * C.n + (l1 - old[1]) * C.t1 + (l2 - old[2]) * C.t2;
            fimp[C.link] = fimp[C.link] + Sv{cross(C.P, dl), dl};
        }
        Sv dv[LG_MAX_DOF + 1];
        float dq[LG_MAX_DOF];
        propagate(fimp, dv, dq, timp);
        for (int l = 0; l < NL; ++l) velf[l] = velf[l] + dv[l];
        for (int d = 0; d < A; ++d) qdf[d] += dq[d];
    }

    // ---- integrate (semi-implicit Euler) and publish
    for (int k = 0; k < nc; ++k) {
        const Contact &C = ct[k];
        V3 lw = mul(Rb, C.lam[0] * C.n + C.lam[1] * C.t1 + C.lam[2] * C.t2);
        float s = cf_weight / dt;
        cf_accum[3 * C.body + 0] += s * lw.x;
        cf_accum[3 * C.body + 1] += s * lw.y;
        cf_accum[3 * C.body + 2] += s * lw.z;
    }
    // Same rule as the HIP kernels.  A runaway body: PhysX clamps its velocities at the asset options max_linear_velocity /
    // max_angular_velocity (legged_robot.py:701-702) and carries on.  The guard is for non-finite state only: keep the pose, bring
    // the env to rest, report the fault (the post-step then terminates and resets it).
    float chk = dot(velf[0].w, velf[0].w) + dot(velf[0].v, velf[0].v);
    for (int d = 0; d < A; ++d) chk += qdf[d] * qdf[d] * 1e-4f;
    if (!ok || !(chk < 3.0e38f)) {
        for (int d = 0; d < A; ++d) dofs[2 * d + 1] = 0.0f;
        for (int k = 7; k < 13; ++k) root[k] = 0.0f;
        return 1;
    }
    int code = 0;
    for (int d = 0; d < A; ++d) {
        float v = qdf[d];
        if (m.q_upper[d] > m.q_lower[d]) {
            // hard stop behind the limit constraints (a few Jacobi sweeps may leave a residual when several limits of one
            // chain are active at once): the joint never ends the step further out than max(limit, where it started)
            const float q0 = dofs[2 * d];
            const float qn = std::min(std::max(q0 + dt * v, std::min(m.q_lower[d], q0)), std::max(m.q_upper[d], q0));
            v = (qn - q0) / dt;
        }
        if (m.vel_limit[d] > 0.0f) v = std::min(std::max(v, -m.vel_limit[d]), m.vel_limit[d]);
        dofs[2 * d + 1] = v;
        dofs[2 * d] += dt * v;
    }
    V3 wn = velf[0].w;
    V3 vn = velf[0].v + dt * cross(wb, vb);                            // classical velocity of the base origin
    {   // the base's velocities as they are published, clamped at the asset's maxima
        const float w2 = dot(wn, wn), v2 = dot(vn, vn);
        if (c.max_angular_velocity > 0.0f && w2 > c.max_angular_velocity * c.max_angular_velocity) { wn = (c.max_angular_velocity / std::sqrt(w2)) * wn; code = 2; }
        if (c.max_linear_velocity > 0.0f && v2 > c.max_linear_velocity * c.max_linear_velocity) { vn = (c.max_linear_velocity / std::sqrt(v2)) * vn; code = 2; }
    }
    V3 vw = mul(Rb, vn), ww = mul(Rb, wn);
    root[0] += dt * vw.x; root[1] += dt * vw.y; root[2] += dt * vw.z;
    root[7] = vw.x; root[8] = vw.y; root[9] = vw.z;
    root[10] = ww.x; root[11] = ww.y; root[12] = ww.z;
    float ang = std::sqrt(dot(wn, wn)) * dt;                            // q <- q * exp(w_body dt)
    float sh, ch = std::cos(0.5f * ang);
    V3 ax;
    if (ang > 1e-8f) { sh = std::sin(0.5f * ang); ax = (dt / ang) * wn; } else { sh = 0.5f * dt; ax = wn; }
    float dq[4] = {sh * ax.x, sh * ax.y, sh * ax.z, ch};
    float *q = root + 3;
    float qn[4] = {q[3] * dq[0] + q[0] * dq[3] + q[1] * dq[2] - q[2] * dq[1],
                   q[3] * dq[1] - q[0] * dq[2] + q[1] * dq[3] + q[2] * dq[0],
                   q[3] * dq[2] + q[0] * dq[1] - q[1] * dq[0] + q[2] * dq[3],
                   q[3] * dq[3] - q[0] * dq[0] - q[1] * dq[1] - q[2] * dq[2]};
    float nrm = 1.0f / std::sqrt(qn[0] * qn[0] + qn[1] * qn[1] + qn[2] * qn[2] + qn[3] * qn[3]);
    for (int k = 0; k < 4; ++k) q[k] = qn[k] * nrm;
    return code;
}

void simulate(Env &e) {
    const int N = e.N, B = e.B;
    const int ns = std::max(1, e.cfg.phys_substeps);
    const float dt = e.cfg.sim_dt / (float)ns;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; ++i) {
        float *cf = &e.contact[(size_t)i * B * 3];
        for (int k = 0; k < 3 * B; ++k) cf[k] = 0.0f;
        for (int s = 0; s < ns; ++s) {
            const int code = simulate_env(e, i, dt, cf, 1.0f / (float)ns);
            if (code & 1) e.fault[i] = 1;
            if (code & 2) {
#pragma omp atomic
                e.clamp_count += 1;
            }
        }
    }
}

}  // namespace lgo
