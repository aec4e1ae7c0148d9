// Pair-lane articulated-body physics for gfx950: TWO lanes per (environment, leg).
//
// Same algorithm, same LDS records and same results (up to fp32 summation order) as physics_lane in lg_physics.h; the
// difference is the lane map.  The control loop lasts as long as the dependent instruction chain of the wave that runs the
// physics (it fills about half of its VALU issue slots: profiles/r03_substeps_pmc.json, r03_substeps_clock.json), and most of
// that chain is 6-vector / 6x6 arithmetic.  Here the two lanes of a
// pair split every spatial quantity by rows: lane h = 0 holds the angular half (w rows), lane h = 1 the linear half
// (v rows) of every spatial vector, and the matching three rows of every 6x6 matrix, stored as two 3x3 blocks relative to
// the lane's own role:  mm multiplies the lane's own half of an operand, mo the partner's half.  With that convention a
// 6x6 product, a rank-one update and a spatial dot product are role-free code:
//      (I x).mine = mm * x.mine + mo * x.other          x.other = DPP quad_perm [1,0,3,2] of x.mine
//      <a, b>     = dot3(a.mine, b.mine) + the partner's (one DPP add)
// Only the spatial cross products and the rigid-body inertia need a role select.  The 3x3 kinematics run redundantly in both
// lanes (identical inputs, identical results).  Scalar work per collision sphere does not (round 4): contact detection and the
// contact law of the sweeps are DEALT between the two lanes -- lane h detects slots 2i + h and takes every other active contact of
// the leg through the sweeps, the halves of the link velocity and of the impulse swapped by DPP -- because a launch lasts as long
// as its workgroup with the most contacts per leg (profiles/r04_substeps_spread.txt).  Lanes of an environment: index = leg * 2 + h, i.e. 2L consecutive lanes; leg sums
// are DPP butterflies over lane xor 2 (and xor 4 for L = 4).  A block's 64 (env, leg) pairs occupy waves 0 and 1, which
// sit on different SIMDs of the CU and run concurrently.
#pragma once
#include "lg_physics.h"

__device__ __forceinline__ float psum(float x) { return x + quad_xor1(x); }                  // sum over the pair
__device__ __forceinline__ V3 px3(V3 a) { return {quad_xor1(a.x), quad_xor1(a.y), quad_xor1(a.z)}; }
__device__ __forceinline__ V3 sel3(bool h, V3 a, V3 b) { return {h ? a.x : b.x, h ? a.y : b.y, h ? a.z : b.z}; }
__device__ __forceinline__ float lane_xor4(float x) {
    int t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141 /*row_half_mirror: i -> 7 - i*/, 0xF, 0xF, true);
    t = __builtin_amdgcn_update_dpp(0, t, 0x1B /*quad_perm [3,2,1,0]: i -> i ^ 3*/, 0xF, 0xF, true);
    return __builtin_bit_cast(float, t);
}
template <int L>
__device__ __forceinline__ float pleg_sum(float x) {     // sum over the L legs of one environment, same role
    static_assert(L == 2 || L == 4, "2L lanes of an environment sit in one group of 4 or 8 lanes");
    x += quad_xor2(x);
    if (L == 4) x += lane_xor4(x);
    return x;
}
template <int L>
__device__ __forceinline__ V3 pleg_sum(V3 a) { return {pleg_sum<L>(a.x), pleg_sum<L>(a.y), pleg_sum<L>(a.z)}; }

struct H6 { M3 mm, mo; };                                // this lane's three rows of a spatial 6x6
__device__ __forceinline__ V3 hmul(const H6 &I, V3 xm, V3 xo) { return mul(I.mm, xm) + mul(I.mo, xo); }
__device__ __forceinline__ V3 hmul(const H6 &I, V3 xm) { return hmul(I, xm, px3(xm)); }
__device__ __forceinline__ float pdot(V3 a, V3 b) { return psum(dot(a, b)); }
// rows of the rigid-body inertia [[Ic + m(cc 1 - c c^T), m cx], [m cx^T, m 1]] about the frame origin
__device__ __forceinline__ H6 rigid_inertia_h(bool h, float m, V3 c, const M3 &Ic) {
    H6 I;
    const float cc = dot(c, c), sm = h ? -m : m;
    const float cv[3] = {c.x, c.y, c.z};
    const float hx[3][3] = {{0.f, -sm * c.z, sm * c.y}, {sm * c.z, 0.f, -sm * c.x}, {-sm * c.y, sm * c.x, 0.f}};
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float top = Ic.m[i][j] + m * ((i == j ? cc : 0.0f) - cv[i] * cv[j]);
            const float bot = (i == j) ? m : 0.0f;
            I.mm.m[i][j] = h ? bot : top;
            I.mo.m[i][j] = hx[i][j];
        }
    return I;
}
// my half of crf(a, f) = {a.w x f.w + a.v x f.v, a.w x f.v}
__device__ __forceinline__ V3 crf_h(bool h, V3 am, V3 ao, V3 fm, V3 fo) {
    const V3 t = cross(sel3(h, ao, am), fm);             // a.w x f.mine in both roles
    const V3 t2 = cross(ao, fo);                         // role 0: a.v x f.v
    return {t.x + (h ? 0.f : t2.x), t.y + (h ? 0.f : t2.y), t.z + (h ? 0.f : t2.z)};
}
// my half of crm(a, b) = {a.w x b.w, a.w x b.v + a.v x b.w}
__device__ __forceinline__ V3 crm_h(bool h, V3 am, V3 ao, V3 bm, V3 bo) {
    const V3 t = cross(sel3(h, ao, am), bm);             // a.w x b.mine
    const V3 t2 = cross(am, bo);                         // role 1: a.v x b.w
    return {t.x + (h ? t2.x : 0.f), t.y + (h ? t2.y : 0.f), t.z + (h ? t2.z : 0.f)};
}

// Inverse of the SPD 6x6 whose rows are spread over the pair; returns this lane's rows of the inverse in H6 form.
// Each lane gathers the matrix in ITS OWN block order [mine, other] (a symmetric permutation of the other lane's), factors
// it and solves for the first three columns only: by symmetry these are the lane's rows, already split as [mm | mo].
// One instruction stream, no role selects; the two lanes' halves come from differently ordered factorisations of the same
// matrix and agree to rounding.
__device__ __forceinline__ bool spd_inverse_h(const H6 &A, H6 &Ainv) {
    float a[6][6];                                       // lower triangle only
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            a[k][j] = A.mm.m[k][j];
            a[3 + k][j] = quad_xor1(A.mo.m[k][j]);       // partner's rows, the columns of my half
            a[3 + k][3 + j] = quad_xor1(A.mm.m[k][j]);
        }
    float Lm[6][6], Li[6];
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            float s = a[i][j];
#pragma unroll
            for (int k = 0; k < j; ++k) s -= Lm[i][k] * Lm[j][k];
            if (i == j) { ok = ok && (s > 0.0f); Li[i] = rsqrtf(fmaxf(s, 1e-30f)); Lm[i][i] = s * Li[i]; }
            else Lm[i][j] = s * Li[j];
        }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        float y[6], x[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            float s = (i == r) ? 1.0f : 0.0f;
#pragma unroll
            for (int k = r; k < i; ++k) s -= Lm[i][k] * y[k];
            y[i] = (i < r) ? 0.f : s * Li[i];
        }
#pragma unroll
        for (int i = 5; i >= 0; --i) {
            float s = y[i];
#pragma unroll
            for (int k = i + 1; k < 6; ++k) s -= Lm[k][i] * x[k];
            x[i] = s * Li[i];
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) { Ainv.mm.m[r][j] = x[j]; Ainv.mo.m[r][j] = x[3 + j]; }
    }
    const float okf = ok ? 1.0f : 0.0f;
    return fminf(okf, quad_xor1(okf)) > 0.5f;            // uniform over the pair
}

// Section timing of the control loop (tools/substeps_sections.py): built only with -DLG_PROF_SUBSTEPS into a separate
// library; s_memtime deltas accumulated per section in SGPRs.
#ifdef LG_PROF_SUBSTEPS
struct SubProf { unsigned long long acc[16], last; };
#define PSTAMP(pr, k) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); (pr).acc[k] += t_ - (pr).last; (pr).last = t_; } while (0)
#else
struct SubProf {};
#define PSTAMP(pr, k) do { } while (0)
#endif

// One contact's record as the sweeps use it, and one projected update of its three impulses (normal, then the two tangents against the
// normal's change, then the friction cone): returns the impulse change in base coordinates.
struct CRec { V3 Pc, nb, t1, t2; float iwn, w10, w20, iw1, w21, iw2, vt, ln, l1, l2; };
__device__ __forceinline__ V3 contact_update(CRec &r, V3 vP, float relax, float mu, bool bounce, float rest /* 0.5 (e_shape + e_ground) */,
                                             float bounce_threshold) {
    float vc0 = dot(r.nb, vP), vc1 = dot(r.t1, vP), vc2 = dot(r.t2, vP);
    if (bounce && vc0 < -bounce_threshold) r.vt = fmaxf(r.vt, -rest * vc0);      // restitution: leave with e x the approach speed
    const float oln = r.ln, ol1 = r.l1, ol2 = r.l2;
    const float ln = fmaxf(0.0f, oln - relax * (vc0 - r.vt) * r.iwn);
    const float dn = ln - oln;
    vc1 += r.w10 * dn;
    vc2 += r.w20 * dn;
    float l1 = ol1 - relax * vc1 * r.iw1;
    vc2 += r.w21 * (l1 - ol1);
    float l2 = ol2 - relax * vc2 * r.iw2;
    const float lim = mu * ln, mag = sqrtf(l1 * l1 + l2 * l2);
    if (mag > lim) { const float sc = lim * frcp(fmaxf(mag, 1e-12f)); l1 *= sc; l2 *= sc; }
    r.ln = ln; r.l1 = l1; r.l2 = l2;
    return (ln - oln) * r.nb + (l1 - ol1) * r.t1 + (l2 - ol2) * r.t2;
}

#define LG_LKP_NF 12     // floats per link record shared by the pair: R 9, p 3
#define LG_LKH_NF 6      // floats per link record per lane: my half of vel 3, of the velocity-product term 3

// One physics step of length dt for lane (env, leg, h).  root, q, qd, tau, friction, dmass are replicated in both lanes
// of the pair and come back identical.  pcol = column of the pair (0..63) in the contact / limit / shared link records,
// lcol = column of the lane (0..127) in the per-lane link records.
template <int L, int J>
__device__ __forceinline__ int physics_pair(const PhysCfg &c /* phys_cfg(P): registers */, int leg, bool h, int pcol, int lcol, float dt,
                                             float *root, float *q, float *qd, const float *tau, float friction, float dmass,
                                             const float *__restrict__ mat /* LDS: this env's restitution, compliance, thickness */,
                                             V3 *fslot, V3 &fbase, float *__restrict__ cst, float *__restrict__ lkp,
                                             float *__restrict__ lkh, const float *__restrict__ ltab, float *__restrict__ lmt,
                                             SubProf &pr, bool want_forces = true /* wave-uniform: fslot / fbase are read by the caller */) {
    const float *__restrict__ lt = ltab + leg * LG_LT_STRIDE;
    const M3 Rb = quat_to_mat(root + 3);
    const V3 xw = {root[0], root[1], root[2]};
    const V3 vb = mulT(Rb, V3{root[7], root[8], root[9]}), wb = mulT(Rb, V3{root[10], root[11], root[12]});
    const V3 gb = mulT(Rb, V3{c.gravity[0], c.gravity[1], c.gravity[2]});
    const V3 vel0 = sel3(h, vb, wb), vel0o = sel3(h, wb, vb);
    const V3 zero3 = {0.f, 0.f, 0.f};
    const int nslots = c.n_leg_slots;
    const unsigned long long link_pk = c.slot_link_pk;      // sphere slot -> link, 4 bits per slot

#define LKP(j, f) lkp[((j) * LG_LKP_NF + (f)) * 64 + pcol]
#define CF(si, f) cst[((si) * LG_CT_NF + (f)) * 64 + pcol]
#define LKH(j, f) lkh[((j) * LG_LKH_NF + (f)) * 128 + lcol]
    V3 S[J], U[J];                                       // my halves
    float iD[J], u[J];
    const float inv_dt = frcp(dt);
    {   // outward kinematics (3x3 part in both lanes, spatial part by halves)
        M3 Rpar = {{{1.f, 0.f, 0.f}, {0.f, 1.f, 0.f}, {0.f, 0.f, 1.f}}};
        V3 ppar = {0.f, 0.f, 0.f};
        V3 vpar = vel0, vparo = vel0o;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const float *jt = lt + LG_LT_JOINT * j;
            M3 Rj = mul(Rpar, load3(jt));
            V3 ax = ld3(jt + 12);
            V3 pj = ppar + mul(Rpar, ld3(jt + 9));
            V3 axb = mul(Rj, ax);
            M3 Rlj = mul(Rj, rodrigues(ax, q[j]));
            const V3 Sw = axb, Sv_ = cross(pj, axb);
            S[j] = sel3(h, Sv_, Sw);
            const V3 So = sel3(h, Sw, Sv_);
            const V3 vj = qd[j] * S[j], vjo = qd[j] * So;
            const V3 velj = vpar + vj, veljo = vparo + vjo;
            const V3 cbj = crm_h(h, velj, veljo, vj, vjo);
#pragma unroll
            for (int e = 0; e < 9; ++e) LKP(j, e) = Rlj.m[e / 3][e % 3];
            LKP(j, 9) = pj.x; LKP(j, 10) = pj.y; LKP(j, 11) = pj.z;
            LKH(j, 0) = velj.x; LKH(j, 1) = velj.y; LKH(j, 2) = velj.z;
            LKH(j, 3) = cbj.x; LKH(j, 4) = cbj.y; LKH(j, 5) = cbj.z;
            // base-frame centres of this link's collision spheres, while R and p are in registers (wave-uniform slot -> link
            // table): the detection loop below then needs three LDS reads per slot instead of the link tile
#pragma unroll
            for (int sl = 0; sl < LG_MAX_LEG_SLOTS; ++sl)
                if (sl < nslots && (int)((link_pk >> (4 * sl)) & 15ull) == j) {
                    const V3 cs = pj + mul(Rlj, ld3(lt + LG_LT_SLOTS + 4 * sl));
                    CF(sl, 0) = cs.x; CF(sl, 1) = cs.y; CF(sl, 2) = cs.z;
                }
            Rpar = Rlj; ppar = pj; vpar = velj; vparo = veljo;
        }
    }
    PSTAMP(pr, 2);
    // inward pass along the chain
    H6 Ia_run;
    V3 pa_run = zero3;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) { Ia_run.mm.m[a][b] = 0.f; Ia_run.mo.m[a][b] = 0.f; }
#pragma unroll
    for (int j = J - 1; j >= 0; --j) {
        const float *jt = lt + LG_LT_JOINT * j;
        M3 Rlj;
#pragma unroll
        for (int e = 0; e < 9; ++e) Rlj.m[e / 3][e % 3] = LKP(j, e);
        const V3 pj = {LKP(j, 9), LKP(j, 10), LKP(j, 11)};
        const V3 velj = {LKH(j, 0), LKH(j, 1), LKH(j, 2)};
        const V3 cbj = {LKH(j, 3), LKH(j, 4), LKH(j, 5)};
        const V3 veljo = px3(velj);
        M3 Ic = mulBT(mul(Rlj, load3(jt + 15)), Rlj);
        H6 IA = rigid_inertia_h(h, jt[27], pj + mul(Rlj, ld3(jt + 24)), Ic);
        const V3 Iv = hmul(IA, velj, veljo);
        V3 pA = crf_h(h, velj, veljo, Iv, px3(Iv));
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) { IA.mm.m[a][b] += Ia_run.mm.m[a][b]; IA.mo.m[a][b] += Ia_run.mo.m[a][b]; }
        pA = pA + pa_run;
        U[j] = hmul(IA, S[j]);
        const float Dj = pdot(S[j], U[j]) + c.armature;
        u[j] = (tau[j] - jt[28] * qd[j]) - pdot(S[j], pA);
        const float invD = frcp(Dj);
        iD[j] = invD;
        const V3 Uo = px3(U[j]);
        const float Um[3] = {U[j].x, U[j].y, U[j].z}, Uov[3] = {Uo.x, Uo.y, Uo.z};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float ua = Um[a] * invD;
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                Ia_run.mm.m[a][b] = IA.mm.m[a][b] - ua * Um[b];
                Ia_run.mo.m[a][b] = IA.mo.m[a][b] - ua * Uov[b];
            }
        }
        pa_run = pA + hmul(Ia_run, cbj) + (u[j] * invD) * U[j];
    }
    PSTAMP(pr, 3);
    // floating base
    H6 I0 = rigid_inertia_h(h, c.base_mass + dmass, ld3(c.base_com), load3(c.base_inertia));
    const V3 I0v = hmul(I0, vel0, vel0o);
    const V3 pA0 = crf_h(h, vel0, vel0o, I0v, px3(I0v)) + pleg_sum<L>(pa_run);
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            I0.mm.m[a][b] += pleg_sum<L>(Ia_run.mm.m[a][b]);
            I0.mo.m[a][b] += pleg_sum<L>(Ia_run.mo.m[a][b]);
        }
    H6 I0inv;
    const bool ok = spd_inverse_h(I0, I0inv);
    const V3 a0 = -1.0f * hmul(I0inv, pA0);
    PSTAMP(pr, 4);
    // outward accelerations -> free velocities
    V3 velf[J], velf0;
    float qdf[J];
    {
        const V3 grav = sel3(h, gb, zero3);
        velf0 = vel0 + dt * (a0 + grav);
        V3 apar = a0;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const V3 velj = {LKH(j, 0), LKH(j, 1), LKH(j, 2)};
            const V3 cbj = {LKH(j, 3), LKH(j, 4), LKH(j, 5)};
            V3 ap = apar + cbj;
            float qdd = (u[j] - pdot(U[j], ap)) * iD[j];
            V3 acc = ap + qdd * S[j];
            velf[j] = velj + dt * (acc + grav);
            qdf[j] = qd[j] + dt * qdd;
            apar = acc;
        }
    }

    PSTAMP(pr, 5);
    // ---- contact detection + W per slot.  The records are shared by the pair in one LDS column, and since round 4 a lane reads records
    // its PARTNER wrote (detection and the sweeps' contact law are dealt between the lanes; W and the force output read every record):
    // both lanes are lanes of one wave, whose LDS instructions execute in program order, and a record's address is computed alike in
    // both, so the compiler keeps write -> read order as well.
    const float mu = 0.5f * (friction + c.ground_friction);
    // Detection is scalar work per sphere, so the pair does not share it either: lane h takes slots 2i + h (the table rows of the leg's
    // spheres and of the base spheres dealt to this leg are contiguous, LG_LT_SLOTS + 4 si), two passes, both written out -- the first
    // places the spheres in the world and fetches the heightfield samples under them (all in flight together), the second evaluates the
    // gaps and writes the records of the slots in contact to the pair's shared column.  Half the instructions of every lane visiting
    // all eight slots (round 4, until here: 10 k of the launch's 150 k cycles).
    static_assert(LG_NUM_SLOTS % 2 == 0, "slots are dealt to the two lanes of a pair");
    const int nbase_it = (c.n_base_spheres + L - 1) / L;
    V3 cbks[LG_NUM_SLOTS / 2], cws[LG_NUM_SLOTS / 2];
    float rads[LG_NUM_SLOTS / 2];
    GroundTap taps[LG_NUM_SLOTS / 2];
#pragma unroll
    for (int i = 0; i < LG_NUM_SLOTS / 2; ++i) {
        const int si = 2 * i + (int)h;
        const bool is_base = si >= LG_MAX_LEG_SLOTS;
        const float *tb = lt + LG_LT_SLOTS + 4 * si;
        const float *cp = is_base ? tb : &CF(si, 0);            // base spheres: the table's centre; leg spheres: from the kinematics pass
        const int cs = is_base ? 1 : 64;
        cbks[i] = {cp[0], cp[cs], cp[2 * cs]};
        rads[i] = tb[3];
        cws[i] = xw + mul(Rb, cbks[i]);
        taps[i] = ground_fetch(c, c.height_samples, cws[i].x, cws[i].y);
    }
    unsigned amine = 0u;
#pragma unroll
    for (int i = 0; i < LG_NUM_SLOTS / 2; ++i) {
        const int si = 2 * i + (int)h;
        const bool is_base = si >= LG_MAX_LEG_SLOTS;
        const int ub = si - LG_MAX_LEG_SLOTS;
        const bool exists = is_base ? (ub < nbase_it && leg + ub * L < c.n_base_spheres) : si < nslots;
        const Ground g = ground_finish(c, taps[i]);
        float gap = (cws[i].z - g.h) * g.n.z - rads[i];
        gap -= c.material_rand ? mat[2] : c.rest_offset;    // shape thickness (asset option, or the env's draw): the robot rests that far off the surface
        if (exists && gap < c.contact_offset) {             // records of inactive slots are never read for a result
            const V3 nb = mulT(Rb, g.n), Pc = cbks[i] - rads[i] * nb;
            const float vtarget = gap >= 0.0f ? -gap * inv_dt : fminf(-gap * c.contact_erp * inv_dt, c.max_depenetration_velocity);
            CF(si, 0) = Pc.x; CF(si, 1) = Pc.y; CF(si, 2) = Pc.z;
            CF(si, 3) = nb.x; CF(si, 4) = nb.y; CF(si, 5) = nb.z;
            CF(si, 12) = vtarget;
            CF(si, 13) = 0.f; CF(si, 14) = 0.f; CF(si, 15) = 0.f;
            amine |= 1u << si;
        }
    }
    // active slots of the pair, and the ones THIS lane takes through the sweeps: every other one in ascending order (bit i of px = parity of
    // the active slots below i)
    const unsigned amask = amine | (unsigned)__builtin_amdgcn_update_dpp(0, (int)amine, 0xB1 /*quad_perm [1,0,3,2]*/, 0xF, 0xF, true);
    unsigned px = amask << 1;
    px ^= px << 1; px ^= px << 2; px ^= px << 4;
    const unsigned mym = amask & (h ? px : ~px);
    PSTAMP(pr, 6);
    for (unsigned rem = amask; __any(rem != 0u); rem &= rem - 1u) {
        const bool valid = rem != 0u;
        const int si = valid ? __ffs(rem) - 1 : 0;
        const int jl = (si >= LG_MAX_LEG_SLOTS) ? -1 : (int)((link_pk >> (4 * si)) & 15ull);
        const V3 Pc = {CF(si, 0), CF(si, 1), CF(si, 2)}, nb = {CF(si, 3), CF(si, 4), CF(si, 5)};
        V3 t1, t2;
        tangents(nb, t1, t2);
        V3 dirs[3] = {nb, t1, t2};
        // W = G^T Phi G (G = the three unit impulses at the contact, Phi = the chain's response at the link) from the INWARD pass alone:
        // with the innovations u_a[k] = -S_k . p_a,k and the force p_a,0 that reaches the base, the articulated-body factorisation
        // M^-1 = (I - H psi K)^T D^-1 (I - H psi K) gives  W_ab = sum_k u_a[k] u_b[k] / D_k + p_a,0 . I0^-1 p_b,0  -- no outward pass, no
        // point-velocity reconstruction (a third fewer instructions than propagating each impulse out again; the one-lane map and the
        // oracle keep the literal form, and the parity tests compare the two).  Joints outside the contact's chain (k > jl) drop out
        // through a zero factor, not a branch: as `if (k <= jl) { ... }` the three impulses compiled into 31 basic blocks (exec-mask
        // regions) that the scheduler could not interleave.
        float Wc[3][3], uu[3][J];
        V3 p0[3], y0[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            V3 pAi = sel3(h, dirs[a], cross(Pc, dirs[a]));
#pragma unroll
            for (int k = J - 1; k >= 0; --k) {
                const float d = pdot(S[k], pAi);
                uu[a][k] = k <= jl ? d : 0.f;
                pAi = pAi - (uu[a][k] * iD[k]) * U[k];
            }
            p0[a] = pAi;
            y0[a] = hmul(I0inv, pAi);
        }
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = a; b < 3; ++b) {
                float w = dot(p0[b], y0[a]);                     // my half; the pair's sum below
#pragma unroll
                for (int k = 0; k < J; ++k) w += 0.5f * (uu[a][k] * iD[k]) * uu[b][k];      // identical in both lanes: half each
                Wc[b][a] = psum(w);
            }
        if (valid) {
            CF(si, 16) = t1.x; CF(si, 17) = t1.y; CF(si, 18) = t1.z;   // first tangent: the sweeps and the force output rebuild t2 = n x t1 only
            if (c.material_rand) Wc[0][0] += mat[1] * inv_dt * inv_dt;      // compliance (m/N) as constraint-force mixing on the normal row
            CF(si, 6) = Wc[0][0] > 1e-9f ? frcp(Wc[0][0]) : 0.f; CF(si, 7) = Wc[1][0]; CF(si, 8) = Wc[2][0];
            CF(si, 9) = Wc[1][1] > 1e-9f ? frcp(Wc[1][1]) : 0.f; CF(si, 10) = Wc[2][1];
            CF(si, 11) = Wc[2][2] > 1e-9f ? frcp(Wc[2][2]) : 0.f;
        }
    }
    const int n_base_active = (int)pleg_sum<L>((float)__popc(amask >> LG_MAX_LEG_SLOTS));
    const int n_leg_active = __popc(amask & ((1u << LG_MAX_LEG_SLOTS) - 1u));
    const float rl = frcp((float)max(n_leg_active, 1)), rb = frcp((float)max(n_base_active, 1));

    PSTAMP(pr, 7);
    // ---- joint position limits, and the joint velocity limit as the same kind of row (a bound on the joint rate enforced by a JOINT-SPACE
    // impulse, i.e. equal and opposite on child and parent: momentum is conserved).  Until round 4 the velocity limit was only the clamp
    // at integration, which takes the child's excess rate away but leaves the base the reaction it was given for it -- a saturated
    // joint pushed by a large torque then pumps angular momentum into the base substep after substep (profiles/r04_diag_faults.txt:
    // base spins of hundreds of rad/s under actions of +-20 and more, what round 3's fault guard was resetting).
#define LM(j, f) lmt[((j) * 4 + (f)) * 64 + pcol]
    unsigned lmask = 0u;
    float lsgn[J], lvt[J];
    float jlo[J], jhi[J], jvl[J];                       // the joints' limits, read once per call (also used by the integration below)
#pragma unroll
    for (int j = 0; j < J; ++j) { jlo[j] = lt[LG_LT_JOINT * j + 30]; jhi[j] = lt[LG_LT_JOINT * j + 31]; jvl[j] = lt[LG_LT_JOINT * j + 29]; }
#pragma unroll
    for (int j = 0; j < J; ++j) {                       // which rows exist: all joints first, so that the common case (none) is one test
        const float lo = jlo[j], hi = jhi[j], vlim = jvl[j];
        // selects, not branches: as nested ifs the three joints' tests compiled into 36 exec-mask regions (1.4 k cycles per substep with no limit in play)
        const float qn = q[j] + dt * qdf[j];
        const bool haslim = hi > lo, up = haslim && qn > hi, dn = haslim && !up && qn < lo;
        float sgn = up ? 1.0f : dn ? -1.0f : 0.f;
        const float gap = up ? hi - q[j] : dn ? q[j] - lo : 0.f;
        // the outward rate the row allows: stop exactly at the position limit (or come back from beyond it), and never above the velocity limit
        float vtarget = gap >= 0.0f ? -gap * inv_dt : fminf(-gap * c.contact_erp * inv_dt, c.max_depenetration_velocity);
        const bool hasv = vlim > 0.0f, free = sgn == 0.f, vup = hasv && free && qdf[j] > vlim, vdn = hasv && free && qdf[j] < -vlim;
        vtarget = (vup || vdn) ? -vlim : (hasv && !free) ? fmaxf(vtarget, -vlim) : vtarget;
        sgn = vup ? 1.0f : vdn ? -1.0f : sgn;
        lsgn[j] = sgn; lvt[j] = vtarget;
        if (sgn != 0.f) lmask |= 1u << j;
    }
    if (__any(lmask != 0u)) {
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const bool act = (lmask >> j) & 1u;
            if (!__any(act)) continue;
            V3 pAi = zero3;
            float ui[J];
#pragma unroll
            for (int k = J - 1; k >= 0; --k) {
                if (k > j) ui[k] = 0.f;
                else {
                    ui[k] = (k == j ? 1.0f : 0.f) - pdot(S[k], pAi);
                    pAi = pAi + (ui[k] * iD[k]) * U[k];
                }
            }
            V3 dv = -1.0f * hmul(I0inv, pAi);
            float Wj = 0.f;
#pragma unroll
            for (int k = 0; k < J; ++k)
                if (k <= j) {
                    const float dq = (ui[k] - pdot(U[k], dv)) * iD[k];
                    dv = dv + dq * S[k];
                    if (k == j) Wj = dq;
                }
            if (act) {
                LM(j, 0) = lsgn[j];
                LM(j, 1) = lvt[j];
                LM(j, 2) = Wj > 1e-9f ? frcp(Wj) : 0.f;
                LM(j, 3) = 0.f;
            }
        }
    }
    const float rlim = frcp((float)max(__popc(lmask), 1));

    PSTAMP(pr, 8);
    // ---- projected Jacobi sweeps
    if (__any((amask | lmask) != 0u)) {
        // The scalar contact law needs no spatial algebra, so the pair does not share it: each lane takes every other active contact of the
        // (env, leg) (mym, dealt at detection) -- a leg with n contacts costs ceil(n / 2) trips, and a launch lasts as long as its workgroup
        // with the most contacts per leg (profiles/r04_substeps_spread.txt: fallen robots, 3-8 contacts per leg, set the launch's duration).
        // Per trip the lanes swap what the other needs: the partner's half of the velocity of the link MY contact sits on, and the partner's half of
        // the impulse I found.  DPP moves run outside the divergent part (they would read disabled partners inside it).
        // The lane's first two contacts -- a robot on its feet has one per leg -- stay in registers for all iterations: no record reads, no
        // impulse writes, no bit scan in the trips nearly every sweep consists of; the second trip is skipped when no lane of the wave has one.
        constexpr int NRC = J <= 3 ? 2 : 1;                     // contacts per lane held in registers (two lanes: up to four per leg); one for the
                                                                // six-joint legs, whose chain state already fills the register file (Cassie: 113 -> 116 us with two)
        bool ra[NRC], rbase[NRC];
        int rsi[NRC], rjl[NRC], rjlp[NRC];
        float rrelax[NRC];
        CRec rr[NRC];
        unsigned mrest = mym;
#pragma unroll
        for (int t = 0; t < NRC; ++t) {
            ra[t] = mrest != 0u;
            rsi[t] = ra[t] ? __ffs(mrest) - 1 : 0;
            rbase[t] = rsi[t] >= LG_MAX_LEG_SLOTS;
            rjl[t] = rbase[t] ? -1 : (int)((link_pk >> (4 * rsi[t])) & 15ull);
            rjlp[t] = __builtin_amdgcn_update_dpp(0, rjl[t], 0xB1 /*quad_perm [1,0,3,2]*/, 0xF, 0xF, true);
            rrelax[t] = rbase[t] ? rb : rl;
            const int si = rsi[t];
            CRec &r = rr[t];
            r.Pc = {CF(si, 0), CF(si, 1), CF(si, 2)}; r.nb = {CF(si, 3), CF(si, 4), CF(si, 5)};
            r.t1 = {CF(si, 16), CF(si, 17), CF(si, 18)}; r.t2 = cross(r.nb, r.t1);
            r.iwn = CF(si, 6); r.w10 = CF(si, 7); r.w20 = CF(si, 8); r.iw1 = CF(si, 9); r.w21 = CF(si, 10); r.iw2 = CF(si, 11);
            r.vt = CF(si, 12); r.ln = 0.f; r.l1 = 0.f; r.l2 = 0.f;
            mrest &= mrest - 1u;
        }
        const float rest = c.material_rand ? 0.5f * (mat[0] + c.ground_restitution) : 0.f;
        for (int it = 0; it < c.solver_iterations; ++it) {
            V3 fimp[J], fb = zero3;
#pragma unroll
            for (int k = 0; k < J; ++k) fimp[k] = zero3;
            const bool bounce = c.material_rand && it == 0;
#pragma unroll
            for (int t = 0; t < NRC; ++t) {                     // the lane's first contacts, from registers
                if (t > 0 && !__any(ra[t])) continue;
                V3 vm = velf0, vq = velf0;                      // my half of my contact's link, and of the partner's
#pragma unroll
                for (int k = 0; k < J; ++k) {
                    if (rjl[t] == k) vm = velf[k];
                    if (rjlp[t] == k) vq = velf[k];
                }
                const V3 vo = px3(vq);                          // the partner's half of MY contact's link
                const V3 wl = sel3(h, vo, vm), vlin = sel3(h, vm, vo);
                const V3 vP = vlin + cross(wl, rr[t].Pc);
                V3 fmine = zero3, fsend = zero3;                // my half of the impulse of my contact, and the partner's half of it
                if (ra[t]) {
                    const V3 dl = contact_update(rr[t], vP, rrelax[t], mu, bounce, rest, c.bounce_threshold);
                    const V3 tq = cross(rr[t].Pc, dl);
                    fmine = sel3(h, dl, tq);
                    fsend = sel3(h, tq, dl);
                }
                const V3 fget = px3(fsend);                     // my half of the partner's contact's impulse (zero if it has none)
                if (rbase[t]) fb = fb + fmine;
                if (rjlp[t] < 0) fb = fb + fget;
#pragma unroll
                for (int k = 0; k < J; ++k) {
                    if (rjl[t] == k) fimp[k] = fimp[k] + fmine;
                    if (rjlp[t] == k) fimp[k] = fimp[k] + fget;
                }
            }
            for (unsigned rem = mrest; __any(rem != 0u); rem &= rem - 1u) {      // further contacts: records in the pair's LDS column
                const bool active = rem != 0u;
                const int si = active ? __ffs(rem) - 1 : 0;
                const bool is_base = si >= LG_MAX_LEG_SLOTS;
                const int jl = is_base ? -1 : (int)((link_pk >> (4 * si)) & 15ull);
                const int jlp = __builtin_amdgcn_update_dpp(0, jl, 0xB1 /*quad_perm [1,0,3,2]*/, 0xF, 0xF, true);     // the link of the partner's contact
                V3 vm = velf0, vq = velf0;
#pragma unroll
                for (int k = 0; k < J; ++k) {
                    if (jl == k) vm = velf[k];
                    if (jlp == k) vq = velf[k];
                }
                const V3 vo = px3(vq);
                const V3 wl = sel3(h, vo, vm), vlin = sel3(h, vm, vo);
                CRec r;
                r.Pc = {CF(si, 0), CF(si, 1), CF(si, 2)}; r.nb = {CF(si, 3), CF(si, 4), CF(si, 5)};
                const V3 vP = vlin + cross(wl, r.Pc);
                V3 fmine = zero3, fsend = zero3;
                if (active) {
                    r.t1 = {CF(si, 16), CF(si, 17), CF(si, 18)}; r.t2 = cross(r.nb, r.t1);
                    r.iwn = CF(si, 6); r.w10 = CF(si, 7); r.w20 = CF(si, 8); r.iw1 = CF(si, 9); r.w21 = CF(si, 10); r.iw2 = CF(si, 11);
                    r.vt = CF(si, 12); r.ln = CF(si, 13); r.l1 = CF(si, 14); r.l2 = CF(si, 15);
                    const V3 dl = contact_update(r, vP, is_base ? rb : rl, mu, bounce, rest, c.bounce_threshold);
                    if (bounce) CF(si, 12) = r.vt;
                    CF(si, 13) = r.ln; CF(si, 14) = r.l1; CF(si, 15) = r.l2;
                    const V3 tq = cross(r.Pc, dl);
                    fmine = sel3(h, dl, tq);
                    fsend = sel3(h, tq, dl);
                }
                const V3 fget = px3(fsend);
                if (is_base) fb = fb + fmine;
                if (jlp < 0) fb = fb + fget;
#pragma unroll
                for (int k = 0; k < J; ++k) {
                    if (jl == k) fimp[k] = fimp[k] + fmine;
                    if (jlp == k) fimp[k] = fimp[k] + fget;
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
            V3 run = zero3;
#pragma unroll
            for (int k = J - 1; k >= 0; --k) {
                V3 cur = run - fimp[k];
                ui[k] = timp[k] - pdot(S[k], cur);
                run = cur + (ui[k] * iD[k]) * U[k];
            }
            V3 pAi0 = pleg_sum<L>(run - fb);
            V3 dv = -1.0f * hmul(I0inv, pAi0);
            velf0 = velf0 + dv;
#pragma unroll
            for (int k = 0; k < J; ++k) {
                float dq = (ui[k] - pdot(U[k], dv)) * iD[k];
                dv = dv + dq * S[k];
                velf[k] = velf[k] + dv;
                qdf[k] += dq;
            }
        }
#pragma unroll
        for (int t = 0; t < NRC; ++t)
            if (ra[t]) { CF(rsi[t], 13) = rr[t].ln; CF(rsi[t], 14) = rr[t].l1; CF(rsi[t], 15) = rr[t].l2; }      // for the force output below
    }

    PSTAMP(pr, 9);
    // ---- contact forces out (world frame, N)
    fbase = {0.f, 0.f, 0.f};
#pragma unroll
    for (int si = 0; si < LG_NUM_SLOTS; ++si) {
        V3 f = {0.f, 0.f, 0.f};
        if (want_forces && ((amask >> si) & 1u)) {
            const V3 nb = {CF(si, 3), CF(si, 4), CF(si, 5)};
            const V3 t1 = {CF(si, 16), CF(si, 17), CF(si, 18)}, t2 = cross(nb, t1);
            f = inv_dt * mul(Rb, CF(si, 13) * nb + CF(si, 14) * t1 + CF(si, 15) * t2);
        }
        if (si >= LG_MAX_LEG_SLOTS) fbase = fbase + f; else fslot[si] = f;
    }
#undef CF
#undef LKP
#undef LKH
#undef LM
    // ---- what PhysX does with a runaway body (asset options max_linear_velocity / max_angular_velocity, legged_robot.py:701-702):
    // it clamps the velocity and carries on.  The guard is for non-finite state only: such an env keeps its pose, is brought to
    // rest and is reported, and the post-step terminates and resets it.
    float chk = pdot(velf0, velf0);
#pragma unroll
    for (int j = 0; j < J; ++j) chk += qdf[j] * qdf[j] * 1e-4f;
    chk = pleg_sum<L>(chk);
    if (!ok || !(chk < 3.0e38f)) {
#pragma unroll
        for (int j = 0; j < J; ++j) qd[j] = 0.f;
#pragma unroll
        for (int k = 7; k < 13; ++k) root[k] = 0.f;
        return 1;
    }
    int code = 0;
    PSTAMP(pr, 10);
    // ---- integrate (both lanes, identical)
#pragma unroll
    for (int j = 0; j < J; ++j) {
        float v = qdf[j];
        {
            const float lo = jlo[j], hi = jhi[j];
            if (hi > lo) {
                const float qn = fminf(fmaxf(q[j] + dt * v, fminf(lo, q[j])), fmaxf(hi, q[j]));
                v = (qn - q[j]) * inv_dt;
            }
        }
        const float vl = jvl[j];
        if (vl > 0.0f) v = fminf(fmaxf(v, -vl), vl);
        qd[j] = v;
        q[j] += dt * v;
    }
    const V3 velf0o = px3(velf0);
    V3 wn = sel3(h, velf0o, velf0);
    V3 vn = sel3(h, velf0, velf0o) + dt * cross(wb, vb);
    {   // the base's velocities as they are published, clamped at the asset's maxima (both lanes hold both and clamp alike)
        const float w2 = dot(wn, wn), v2 = dot(vn, vn);
        if (c.max_angular_velocity > 0.0f && w2 > c.max_angular_velocity * c.max_angular_velocity) { wn = (c.max_angular_velocity * rsqrtf(w2)) * wn; code = 2; }
        if (c.max_linear_velocity > 0.0f && v2 > c.max_linear_velocity * c.max_linear_velocity) { vn = (c.max_linear_velocity * rsqrtf(v2)) * vn; code = 2; }
    }
    V3 vw = mul(Rb, vn), ww = mul(Rb, wn);
    root[0] += dt * vw.x; root[1] += dt * vw.y; root[2] += dt * vw.z;
    root[7] = vw.x; root[8] = vw.y; root[9] = vw.z;
    root[10] = ww.x; root[11] = ww.y; root[12] = ww.z;
    float ang = sqrtf(dot(wn, wn)) * dt;
    float sh, ch;
    V3 ax;
    if (__all(ang < 0.5f)) {                            // every robot of the wave turns less than half a radian this substep (28 deg: 100 rad/s
        // at the 200 Hz physics): sin and cos of the half angle from their series, truncation below 1e-12 -- the library calls carry an
        // argument reduction for any angle that costs ~400 cycles of every substep
        const float a = 0.5f * ang, a2 = a * a;
        ch = 1.0f + a2 * (-0.5f + a2 * (4.16666667e-2f + a2 * (-1.38888889e-3f + a2 * 2.48015873e-5f)));
        const float sinc = 1.0f + a2 * (-1.66666667e-1f + a2 * (8.33333333e-3f + a2 * (-1.98412698e-4f + a2 * 2.75573192e-6f)));
        sh = 0.5f * dt * sinc;                          // sin(a) / |wn|: the axis below stays unnormalised
        ax = wn;
    } else {
        ch = cosf(0.5f * ang);
        if (ang > 1e-8f) { sh = sinf(0.5f * ang); ax = (dt / ang) * wn; } else { sh = 0.5f * dt; ax = wn; }
    }
    float dq[4] = {sh * ax.x, sh * ax.y, sh * ax.z, ch};
    float *qq = root + 3;
    float qn[4] = {qq[3] * dq[0] + qq[0] * dq[3] + qq[1] * dq[2] - qq[2] * dq[1],
                   qq[3] * dq[1] - qq[0] * dq[2] + qq[1] * dq[3] + qq[2] * dq[0],
                   qq[3] * dq[2] + qq[0] * dq[1] - qq[1] * dq[0] + qq[2] * dq[3],
                   qq[3] * dq[3] - qq[0] * dq[0] - qq[1] * dq[1] - qq[2] * dq[2]};
    float nrm = rsqrtf(qn[0] * qn[0] + qn[1] * qn[1] + qn[2] * qn[2] + qn[3] * qn[3]);
#pragma unroll
    for (int k = 0; k < 4; ++k) qq[k] = qn[k] * nrm;
    PSTAMP(pr, 11);
    return code;
}
