// Environment step kernels for gfx950 (wave64).  Reference semantics: LeggedRobot.step /
// post_physics_step (legged_gym/envs/base/legged_robot.py:80-226, "LR"), Anymal._compute_torques
// (envs/anymal_c/anymal.py:71-81, "AN"), Cassie._reward_no_fly (envs/cassie/cassie.py:43-46, "CA").
#include "lg_device.h"
#include "lg_physics_pair.h"
#include "lg_traj.h"
#include "lg_finalize.h"

// ------------------------------------------------------------------------------------------------
// LR:86-87  actions = clip(actions, +-clip_actions)
__global__ void k_set_actions(const DevParams *__restrict__ P, const float *__restrict__ a_in) {
    const int n = P->cfg.num_envs * P->cfg.num_actions;
    const float c = P->cfg.clip_actions;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        P->buf.actions[i] = fminf(fmaxf(a_in[i], -c), c);
}

// ------------------------------------------------------------------------------------------------
// Actuator net, wide form: 8 lanes per (env, joint) row, lane k owns hidden unit k of both LSTM layers.  The 113 weights
// lane k needs are staged once per block into one 16-byte aligned LDS record per k (stride 116 floats: the eight records'
// b128 reads fall on disjoint banks) and read into VGPRs once per substep for all rows of the thread.  The 8 hidden values
// of a row are exchanged with DPP only (quad rotations + row_half_mirror, lstm_acc below), no LDS round trip; the image stores each
// lane's weights in the order the lane meets the units, so the eight lanes add their eight products in eight different orders.
// All NR rows of a thread advance stage by stage, which gives the scheduler NR independent chains to hide the exp / rcp latencies with.
__device__ __forceinline__ float fsigm(float x) { return frcp(1.0f + __expf(-x)); }
__device__ __forceinline__ float ftanh(float x) { return 2.0f * frcp(1.0f + __expf(-2.0f * x)) - 1.0f; }
template <int Q>
__device__ __forceinline__ float quad_bcast(float x) {                   // lane (i & ~3) + Q of every quad
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), Q * 0x55, 0xF, 0xF, true));
}
// g[r][gate] += sum_u W[gate][u] h_r[u] over the 8 units of the row.  The row's hidden values are rotated inside their quads (r = 0..3:
// lane j then holds unit (j & 4) + ((j + r) & 3)); lane k multiplies what it holds itself and what lane 7 - k holds (row_half_mirror, a DPP
// operand of the multiply-add: no instruction of its own) by the two weight quadruples the image stores for (k, r) -- 3 moves and 32
// multiply-adds per row and matrix (round 3: quad broadcasts and a two-move swap per unit, 56 instructions).
template <int R>
__device__ __forceinline__ float quad_rot(float x) {                     // lane j <- lane (j & ~3) + ((j + R) & 3)
    constexpr int ctl = ((0 + R) & 3) | (((1 + R) & 3) << 2) | (((2 + R) & 3) << 4) | (((3 + R) & 3) << 6);
    return R == 0 ? x : __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), ctl, 0xF, 0xF, true));
}
__device__ __forceinline__ float half_mirror(float x) {                  // lane j <- lane 7 - j of its group of eight
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xF, 0xF, true));
}
template <int R, int NR>
__device__ __forceinline__ void lstm_acc_r(float (&g)[NR][4], const float (&h)[NR], const float *__restrict__ wm) {
    const float4 wo = *reinterpret_cast<const float4 *>(wm + R * 8), wx = *reinterpret_cast<const float4 *>(wm + R * 8 + 4);
    const float wov[4] = {wo.x, wo.y, wo.z, wo.w}, wxv[4] = {wx.x, wx.y, wx.z, wx.w};
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const float t = quad_rot<R>(h[r]), x = half_mirror(t);
#pragma unroll
        for (int gt = 0; gt < 4; ++gt) g[r][gt] = fmaf(wxv[gt], x, fmaf(wov[gt], t, g[r][gt]));
    }
}
template <int NR>
__device__ __forceinline__ void lstm_acc(float (&g)[NR][4], const float (&h)[NR], const float *__restrict__ wm) {
    lstm_acc_r<0, NR>(g, h, wm);
    lstm_acc_r<1, NR>(g, h, wm);
    lstm_acc_r<2, NR>(g, h, wm);
    lstm_acc_r<3, NR>(g, h, wm);
}
// One actuator-net update of the NR rows of this thread (lane k = hidden unit k of both layers); wr = the LDS record of k.
// y[r] = the row's output sum, in every lane of the row.
template <int NR>
__device__ __forceinline__ void lstm8_rows(const float *__restrict__ wr, int k, const float (&x0)[NR], const float (&x1)[NR], float (&h0)[NR],
                                           float (&c0)[NR], float (&h1)[NR], float (&c1)[NR], float (&y)[NR]) {
    const float4 b0 = *reinterpret_cast<const float4 *>(wr), b1 = *reinterpret_cast<const float4 *>(wr + 4);
    const float4 wa = *reinterpret_cast<const float4 *>(wr + 8), wb = *reinterpret_cast<const float4 *>(wr + 12);
    const float bias0[4] = {b0.x, b0.y, b0.z, b0.w}, bias1[4] = {b1.x, b1.y, b1.z, b1.w};
    const float wi[4][2] = {{wa.x, wa.y}, {wa.z, wa.w}, {wb.x, wb.y}, {wb.z, wb.w}};
    float g[NR][4];
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int gt = 0; gt < 4; ++gt) g[r][gt] = bias0[gt] + wi[gt][0] * x0[r] + wi[gt][1] * x1[r];
    lstm_acc<NR>(g, h0, wr + 16);
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        c0[r] = fsigm(g[r][1]) * c0[r] + fsigm(g[r][0]) * ftanh(g[r][2]);
        h0[r] = fsigm(g[r][3]) * ftanh(c0[r]);
    }
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int gt = 0; gt < 4; ++gt) g[r][gt] = bias1[gt];
    lstm_acc<NR>(g, h0, wr + 48);
    lstm_acc<NR>(g, h1, wr + 80);
    const float lw = wr[112];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        c1[r] = fsigm(g[r][1]) * c1[r] + fsigm(g[r][0]) * ftanh(g[r][2]);
        h1[r] = fsigm(g[r][3]) * ftanh(c1[r]);
        float t = lw * h1[r];
        t += quad_xor1(t);
        t += quad_xor2(t);
        t += lane_xor4(t);
        y[r] = t;
    }
}

// ------------------------------------------------------------------------------------------------
// The control loop of LeggedRobot.step (LR:86-96): clip the actions, then `iters` x {torque law, physics substep} with
// the robot state (root, q, qd), the actuator-net state (h, c of both layers) and the model constants resident in
// registers / LDS for the whole loop -- read once and written once per env step instead of once per substep, and 1 launch
// instead of 1 + 2 x decimation.  Block = 4 waves = 64/L environments.  Waves 0 and 1 run the physics, two lanes per
// (env, leg) (lg_physics_pair.h; PAIR = false: wave 0 alone, one lane per (env, leg), lg_physics.h).  All 256 lanes run the
// actuator net (8 lanes per (env, joint) row, 2J rounds per substep, the state of each round in VGPRs); PD laws are evaluated
// by the physics lanes for their own joints.  q, qd and tau cross between the two lane maps through LDS.
//
// The operator-level entry points run THE SAME KERNEL with one stage switched off: lg_compute_torques = torque stage only
// (LG_RUN_TORQUES, one iteration), lg_simulate = physics stage only with the torques read from the buffer (LG_RUN_PHYSICS).
// One instance of the torque code and one of the physics serve all three, so lg_step equals the launch-per-substep
// sequence bit for bit (fp32 loads/stores between launches are exact; tests/test_hip_env.py asserts equality).
//
// Measured and not kept: post_physics_step in the tail of this launch, each workgroup for its own environments (same device code,
// bit-identical): neutral at 4096 envs (every workgroup's post-step is a latency chain as long as the whole k_post_step launch)
// and it doubles this kernel's LDS footprint (75 -> 132 KB; Cassie 99 -> 156 KB), which takes away the second resident workgroup
// per CU that larger env counts use.
#define LG_RUN_TORQUES 1
#define LG_RUN_PHYSICS 2
// OCC = waves per SIMD the kernel is compiled for.  1: the whole register file (292 VGPRs for the quadruped), the shape every launch of up
// to one workgroup per CU takes; 2: 256 VGPRs (152 B of scratch), so that two workgroups of a larger env count share a CU instead of
// running one after the other (8192 envs: 110 us against 143; at 4096 the spills cost 6 %: profiles/r04_env_count_sweep.txt).
template <int L, int J, bool LSTM, bool PAIR, int NW = 4, int OCC = 1>
__global__ void __launch_bounds__(64 * NW, OCC) k_substeps(const DevParams *__restrict__ P, const float *__restrict__ a_in, int mode, int iters) {
    // NW waves per block.  NW = 4: two physics waves (pair-lane map) + two that only run the actuator net, 64/L envs.  NW = 2 (pair-lane
    // map only): one physics wave + one actuator-net wave, 32/L envs -- two such blocks per CU, and no physics wave ever waits at a
    // substep barrier for the other one's contacts (LG_SUBSTEPS_NW).
    static_assert(NW == 4 || (NW == 2 && PAIR), "block shapes: 4 waves, or 2 waves with the pair-lane physics");
    constexpr int NT = 64 * NW, RPP = NT / 8;                   // threads; actuator-net rows per pass (8 lanes per row)
    constexpr int PW = PAIR ? NW / 2 : 1, LPE = PAIR ? 2 * L : L;    // physics waves of the block, physics lanes per env
    constexpr int A = L * J, EPW = 64 * PW / LPE, ROWS = EPW * A, NR = ROWS * 8 / NT;
    const lg_cfg &c = P->cfg;
    const lg_model &m = P->model;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int N = c.num_envs, B = c.num_bodies;
    const int env0 = blockIdx.x * EPW;
    const int nrow = min(ROWS, (N - env0) * A);              // live rows of this block
    const bool do_tau = mode & LG_RUN_TORQUES, do_phys = mode & LG_RUN_PHYSICS;
    SubProf pr;
#ifdef LG_PROF_SUBSTEPS
    for (int k = 0; k < 16; ++k) pr.acc[k] = 0;
    pr.last = __builtin_amdgcn_s_memtime();
    const unsigned long long prof_rt0 = __builtin_amdgcn_s_memrealtime();      // 100 MHz: in-kernel clock = ticks / realtime x 0.1 GHz
#endif
    __shared__ float s_ct[LG_NUM_SLOTS * LG_CT_NF * 64];
    __shared__ float s_lk[J * LG_LK_NF * 64];
    __shared__ float s_lt[L * LG_LT_STRIDE];
    __shared__ float s_lm[J * 4 * 64];
    __shared__ __attribute__((aligned(16))) float s_w[LSTM ? LG_LSTM_LDS : 4];
    __shared__ float s_act[ROWS], s_q[ROWS], s_qd[ROWS], s_tau[ROWS];
    __shared__ float s_cf[EPW * LG_MAX_BODIES * 3];          // net contact force per (env, body) of the last control substep
    __shared__ float s_mat[EPW * 4];                         // per-env shape material (restitution, compliance, thickness): read only
    if (c.material_rand)                                     // when the cfg randomises them (lg_cfg.material_rand)
        for (int t = tid; t < EPW * 4; t += NT) s_mat[t] = P->buf.material[(size_t)min(env0 + t / 4, N - 1) * 4 + (t & 3)];
    for (int t = tid; t < EPW * LG_MAX_BODIES * 3; t += NT) s_cf[t] = 0.f;
    for (int t = tid; t < L * LG_LT_STRIDE; t += NT) s_lt[t] = (&P->leg_tab[0][0])[t];
    if (LSTM)
        for (int t = tid; t < LG_LSTM_LDS; t += NT) s_w[t] = P->lstm_img[t];
    const size_t row0 = (size_t)env0 * A;
    for (int t = tid; t < ROWS; t += NT) {
        const bool in = t < nrow;
        const size_t r = row0 + (in ? t : 0);
        const float a = fminf(fmaxf(a_in[r], -c.clip_actions), c.clip_actions);      // LR:86-87 (idempotent on clipped input)
        if (in) P->buf.actions[r] = a;
        const float2 st = reinterpret_cast<const float2 *>(P->buf.dof_state)[r];
        s_act[t] = a; s_q[t] = st.x; s_qd[t] = st.y;
        s_tau[t] = P->buf.torques[r];
    }
    // actuator-net state of this thread's NR rows
    float h0[NR], c0[NR], h1[NR], c1[NR];
    const size_t ls = (size_t)N * A * 8;
    if (LSTM && do_tau) {
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int rl = r * RPP + (tid >> 3);
            const size_t idx = (row0 + (rl < nrow ? rl : 0)) * 8 + (tid & 7);
            h0[r] = P->buf.lstm_h[idx]; c0[r] = P->buf.lstm_c[idx]; h1[r] = P->buf.lstm_h[ls + idx]; c1[r] = P->buf.lstm_c[ls + idx];
        }
    }
    // physics lanes: wave 0, one lane per (env, leg) -- or waves 0 and 1, two lanes per (env, leg) (lg_physics_pair.h);
    // of a pair, lane h = 0 does the stores
    const int ptid = tid & (64 * PW - 1);
    const bool phys = wave < PW, hrole = PAIR && (ptid & 1), writer = !hrole;
    const int pe = ptid / LPE;                               // env of this lane within the block
    int env = env0 + pe;
    const int leg = PAIR ? (ptid >> 1) % L : ptid % L;
    const bool live = env < N;
    if (!live) env = N - 1;
    const int rl0 = pe * A + leg * J;                        // first row of this lane's joints in the block
    float root[13], q[J], qd[J], tau[J];
    float fr = 0.f, dm = 0.f;
    if (phys) {
        const float *rp = P->buf.root_states + (size_t)env * 13;
#pragma unroll
        for (int k = 0; k < 13; ++k) root[k] = rp[k];
        fr = P->buf.friction[env]; dm = P->buf.base_mass_delta[env];
    }
    // per-launch constants of the loop, fetched once and BEFORE the barrier, with everything else this launch reads from memory: a global
    // load inside the substep loop is a round trip to the L2 on the critical path of every substep (the compiler does not hoist them past
    // the LDS traffic: six serialised loads per actuator-net pass, seven in the contact-force accumulation -- found in the ISA, round 4)
    const PhysCfg pk = phys_cfg(P);            // the physics' launch constants, in registers for the whole control loop
    const float action_scale = c.action_scale;
    const int control_type = c.control_type;
    const float sim_dt = c.sim_dt;
    float qtgt[NR];                                                              // AN:72  actions * scale + default_dof_pos
    if (LSTM && do_tau) {
#pragma unroll
        for (int r = 0; r < NR; ++r) qtgt[r] = c.default_dof_pos[(r * RPP + (tid >> 3)) % A];
    }
    float kp[J], kd[J], tlim[J], ptgt[J], lqd[J];                                // LR:389-413: gains, limits and the target of this lane's joints
    if (!LSTM && do_tau && phys) {
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int d = leg * J + j;
            kp[j] = c.p_gains[d]; kd[j] = c.d_gains[d]; tlim[j] = c.torque_limits[d];
            ptgt[j] = control_type == 0 ? c.default_dof_pos[d] : 0.f;
            lqd[j] = control_type == 1 ? P->buf.last_dof_vel[(size_t)env * A + d] : 0.f;
        }
    }
    int slot_body[LG_MAX_LEG_SLOTS], base_body0 = 0;
    if (do_phys && phys) {
#pragma unroll
        for (int k = 0; k < LG_MAX_LEG_SLOTS; ++k) slot_body[k] = k < pk.n_leg_slots ? P->slot_body[k][leg] : 0;
        base_body0 = P->base_body[0];
    }
    __syncthreads();
    if (phys) {
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int t = live ? rl0 + j : 0;
            q[j] = s_q[t]; qd[j] = s_qd[t]; tau[j] = s_tau[t];
        }
    }
    const int ns = c.phys_substeps > 1 ? c.phys_substeps : 1;
    const float dt = c.sim_dt / (float)ns, wgt = 1.0f / (float)ns;
    float *cf = s_cf + pe * B * 3;
    if (LSTM && do_tau) {
#pragma unroll
        for (int r = 0; r < NR; ++r) qtgt[r] = s_act[r * RPP + (tid >> 3)] * action_scale + qtgt[r];
    }
    if (!LSTM && do_tau && phys) {
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const float as = s_act[live ? rl0 + j : 0] * action_scale;
            ptgt[j] = control_type == 0 ? as + ptgt[j] : as;
        }
    }
    PSTAMP(pr, 0);
    for (int sub = 0; sub < iters; ++sub) {
        // ---- torques
        if (do_tau) {
            if (LSTM) {                                                      // AN:71-81
                float x0[NR], x1[NR], y[NR];
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    const int rl = r * RPP + (tid >> 3);
                    x0[r] = (qtgt[r] - s_q[rl]) * s_w[0];
                    x1[r] = s_qd[rl] * s_w[1];
                }
                lstm8_rows<NR>(s_w + 4 + (tid & 7) * LG_LSTM_REC, tid & 7, x0, x1, h0, c0, h1, c1, y);
                if ((tid & 7) == 0) {
#pragma unroll
                    for (int r = 0; r < NR; ++r) s_tau[r * RPP + (tid >> 3)] = s_w[2] * (y[r] + s_w[3]);
                }
                __syncthreads();
                if (phys) {
#pragma unroll
                    for (int j = 0; j < J; ++j) tau[j] = s_tau[live ? rl0 + j : 0];
                }
            } else if (phys) {                                               // LR:389-413
#pragma unroll
                for (int j = 0; j < J; ++j) {
                    float t;
                    if (control_type == 0) t = kp[j] * (ptgt[j] - q[j]) - kd[j] * qd[j];
                    else if (control_type == 1) t = kp[j] * (ptgt[j] - qd[j]) - kd[j] * (qd[j] - lqd[j]) / sim_dt;
                    else t = ptgt[j];
                    tau[j] = clampf(t, -tlim[j], tlim[j]);
                    if (live && writer) s_tau[rl0 + j] = tau[j];
                }
            }
        }
        PSTAMP(pr, 1);
        // ---- physics
        if (do_phys && phys) {
            const bool last = sub == iters - 1;
            for (int s = 0; s < ns; ++s) {
                V3 fslot[LG_MAX_LEG_SLOTS], fbase;
                int fault;                                          // bit 0: non-finite solve (reset), bit 1: base velocity clamped
                V3 fb;
                if constexpr (PAIR) {
                    fault = physics_pair<L, J>(pk, leg, hrole, ptid >> 1, ptid, dt, root, q, qd, tau, fr, dm, s_mat + 4 * pe, fslot, fbase, s_ct, s_lk,
                                               s_lk + J * LG_LKP_NF * 64, s_lt, s_lm, pr, last);
                    fb = pleg_sum<L>(fbase);
                } else {
                    fault = physics_lane<L, J>(P, leg, dt, root, q, qd, tau, fr, dm, s_mat + 4 * pe, fslot, fbase, s_ct, s_lk, s_lt, s_lm);
                    fb = {leg_sum<L>(fbase.x), leg_sum<L>(fbase.y), leg_sum<L>(fbase.z)};
                }
                if ((fault & 1) && live && leg == 0 && writer) P->fault[env] = 1;
                if ((fault & 2) && live && leg == 0 && writer) atomicAdd(P->clamp_count, 1);
                if (last && live && writer) {
#pragma unroll
                    for (int k = 0; k < LG_MAX_LEG_SLOTS; ++k)
                        if (k < pk.n_leg_slots) {
                            float *o = cf + 3 * slot_body[k];
                            o[0] += wgt * fslot[k].x; o[1] += wgt * fslot[k].y; o[2] += wgt * fslot[k].z;
                        }
                    if (leg == 0 && pk.n_base_spheres > 0) {
                        float *o = cf + 3 * base_body0;
                        o[0] += wgt * fb.x; o[1] += wgt * fb.y; o[2] += wgt * fb.z;
                    }
                }
            }
            if (live && writer) {
#pragma unroll
                for (int j = 0; j < J; ++j) { s_q[rl0 + j] = q[j]; s_qd[rl0 + j] = qd[j]; }
            }
        }
        PSTAMP(pr, 12);
        __syncthreads();
        PSTAMP(pr, 13);
    }
    // ---- write back (once per launch)
    if (do_phys)
        for (int t = tid; t < min(EPW, N - env0) * B * 3; t += NT) P->buf.contact_forces[(size_t)env0 * B * 3 + t] = s_cf[t];
    for (int t = tid; t < nrow; t += NT) {
        if (do_phys) reinterpret_cast<float2 *>(P->buf.dof_state)[row0 + t] = make_float2(s_q[t], s_qd[t]);
        if (do_tau) P->buf.torques[row0 + t] = s_tau[t];
    }
    if (LSTM && do_tau) {
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int rl = r * RPP + (tid >> 3);
            if (rl < nrow) {
                const size_t idx = (row0 + rl) * 8 + (tid & 7);
                P->buf.lstm_h[idx] = h0[r]; P->buf.lstm_c[idx] = c0[r]; P->buf.lstm_h[ls + idx] = h1[r]; P->buf.lstm_c[ls + idx] = c1[r];
            }
        }
    }
    if (do_phys && phys && live && leg == 0 && writer) {
        float *wp = P->buf.root_states + (size_t)env * 13;
#pragma unroll
        for (int k = 0; k < 13; ++k) wp[k] = root[k];
    }
#ifdef LG_PROF_SUBSTEPS
    PSTAMP(pr, 14);
    pr.acc[15] = __builtin_amdgcn_s_memrealtime() - prof_rt0;
#ifdef LG_PROF_SPAN        // every workgroup's first and last tick of the chip-wide 100 MHz clock (tools/substeps_span.py): dispatch skew and tail
    if (tid == 0 && blockIdx.x < 256) { P->dbg_cycles[blockIdx.x * 2] = prof_rt0; P->dbg_cycles[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime(); }
#else
    if ((tid & 63) == 0 && wave < 2 && blockIdx.x < 16)
        for (int k = 0; k < 16; ++k) P->dbg_cycles[(blockIdx.x * 2 + wave) * 16 + k] = pr.acc[k];
#endif
#endif
}

// ------------------------------------------------------------------------------------------------
// post_physics_step (LR:106-137) + observation clip (LR:100-103), one workgroup per tile of envs.
__device__ __forceinline__ float uni(const DevParams *P, int env, int slot, int64_t counter, int inject) {
    if (inject) return P->buf.inject_uniforms[(size_t)env * P->K + slot];
    return philox_uniform(P->cfg.seed, (uint32_t)(P->cfg.env_offset + env), (uint64_t)counter, (uint32_t)slot);
}

// lo / hi: the command ranges in force -- cfg.cmd_* for a reset env, cb.cmd_* inside the step callback (lg_device.h StageCb)
__device__ void resample_commands(const DevParams *P, int i, int slot0, int64_t counter, int inject, const float *lo, const float *hi) {   // LR:365-387
    const lg_cfg &c = P->cfg;
    float *cmd = P->buf.commands + (size_t)i * 4;
    float c0 = (hi[0] - lo[0]) * uni(P, i, slot0 + 0, counter, inject) + lo[0];
    float c1 = (hi[1] - lo[1]) * uni(P, i, slot0 + 1, counter, inject) + lo[1];
    float u2 = uni(P, i, slot0 + 2, counter, inject);
    if (c.heading_command) cmd[3] = (hi[3] - lo[3]) * u2 + lo[3];
    else cmd[2] = (hi[2] - lo[2]) * u2 + lo[2];
    float keep = sqrtf(c0 * c0 + c1 * c1) > 0.2f ? 1.0f : 0.0f;
    cmd[0] = c0 * keep;
    cmd[1] = c1 * keep;
}

__device__ __forceinline__ float fnorm3(const float *f) { return sqrtf(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]); }

struct RewardCtx {
    const float *js;    // the eight joint-summed terms of this env (k_post_step phase A1), JS_* below
    float *air;         // feet_air_time / last_contacts rows of this env (LDS copies: the stateful feet_air_time term updates them)
    uint8_t *lc;
    V3 blv, bav, pg;
    const float *fr;    // the same three frame vectors as 9 floats in LDS: what signal_at indexes (a select chain over the V3
                        // members made the compiler keep this whole struct in scratch)
    const float *cmd, *cf, *tau, *act, *lact, *lqd;
    const float *dof;   // interleaved q, qd
    float root_z;
    bool reset, time_out;
};

enum { JS_TORQUES = 0, JS_DOF_VEL, JS_DOF_ACC, JS_ACTION_RATE, JS_DOF_POS_LIMITS, JS_DOF_VEL_LIMITS, JS_TORQUE_LIMITS, JS_STAND_STILL, JS_N };
__device__ float reward_term(const DevParams *P, int i, int k, const RewardCtx &x) {   // LR:918-1015, CA:43-46
    const lg_cfg &c = P->cfg;
    const int A = c.num_actions, F = c.num_feet, H = c.num_height_points;
    float s = 0.0f;
    switch (k) {
    case LG_REW_LIN_VEL_Z: return x.blv.z * x.blv.z;
    case LG_REW_ANG_VEL_XY: return x.bav.x * x.bav.x + x.bav.y * x.bav.y;
    case LG_REW_ORIENTATION: return x.pg.x * x.pg.x + x.pg.y * x.pg.y;
    case LG_REW_BASE_HEIGHT: {
        const float *hh = P->buf.measured_heights + (size_t)i * H;
        for (int h = 0; h < H; ++h) s += x.root_z - hh[h];
        float bh = s / (float)H;
        return (bh - c.base_height_target) * (bh - c.base_height_target);
    }
    case LG_REW_TORQUES: return x.js[JS_TORQUES];
    case LG_REW_DOF_VEL: return x.js[JS_DOF_VEL];
    case LG_REW_DOF_ACC: return x.js[JS_DOF_ACC];
    case LG_REW_ACTION_RATE: return x.js[JS_ACTION_RATE];
    case LG_REW_COLLISION:
        for (int b = 0; b < c.num_pen; ++b) s += fnorm3(x.cf + 3 * c.pen_idx[b]) > 0.1f ? 1.0f : 0.0f;
        return s;
    case LG_REW_TERMINATION: return (x.reset && !x.time_out) ? 1.0f : 0.0f;
    case LG_REW_DOF_POS_LIMITS: return x.js[JS_DOF_POS_LIMITS];
    case LG_REW_DOF_VEL_LIMITS: return x.js[JS_DOF_VEL_LIMITS];
    case LG_REW_TORQUE_LIMITS: return x.js[JS_TORQUE_LIMITS];
    case LG_REW_TRACKING_LIN_VEL: {
        float dx = x.cmd[0] - x.blv.x, dy = x.cmd[1] - x.blv.y;
        return expf(-(dx * dx + dy * dy) / c.tracking_sigma);
    }
    case LG_REW_TRACKING_ANG_VEL: {
        float d = x.cmd[2] - x.bav.z;
        return expf(-(d * d) / c.tracking_sigma);
    }
    case LG_REW_FEET_AIR_TIME: {
        float *air = x.air;                                            // staged in LDS by the caller, written back after phase A
        uint8_t *lc = x.lc;
        for (int f = 0; f < F; ++f) {
            bool contact = x.cf[3 * c.feet_idx[f] + 2] > 1.0f;
            bool filt = contact || lc[f];
            lc[f] = contact;
            float a = air[f];
            float first = (a > 0.0f && filt) ? 1.0f : 0.0f;
            a += c.dt;
            s += (a - 0.5f) * first;
            air[f] = filt ? 0.0f : a;
        }
        if (c.feet_air_time_ungated) return s;                        // trajectory env: no command gate (LT:1071-1080)
        float cn = sqrtf(x.cmd[0] * x.cmd[0] + x.cmd[1] * x.cmd[1]);
        return s * (cn > 0.1f ? 1.0f : 0.0f);
    }
    case LG_REW_STUMBLE: {
        bool any = false;
        for (int f = 0; f < F; ++f) {
            const float *ff = x.cf + 3 * c.feet_idx[f];
            any |= sqrtf(ff[0] * ff[0] + ff[1] * ff[1]) > 5.0f * fabsf(ff[2]);
        }
        return any ? 1.0f : 0.0f;
    }
    case LG_REW_STAND_STILL: {
        float cn = sqrtf(x.cmd[0] * x.cmd[0] + x.cmd[1] * x.cmd[1]);
        return x.js[JS_STAND_STILL] * (cn < 0.1f ? 1.0f : 0.0f);
    }
    case LG_REW_FEET_CONTACT_FORCES:
        for (int f = 0; f < F; ++f) s += fmaxf(fnorm3(x.cf + 3 * c.feet_idx[f]) - c.max_contact_force, 0.0f);
        return s;
    case LG_REW_NO_FLY: {
        int n = 0;
        for (int f = 0; f < F; ++f) n += x.cf[3 * c.feet_idx[f] + 2] > 0.1f ? 1 : 0;
        return n == 1 ? 1.0f : 0.0f;
    }
    }
    return 0.0f;
}

// per-env signals an extra reward term may read (legged_hip.h lg_signal)
__device__ float signal_at(const DevParams *P, int i, int sig, int k, const RewardCtx &x) {
    const lg_cfg &c = P->cfg;
    switch (sig) {
    case LG_SIG_BASE_LIN_VEL: return x.fr[k];
    case LG_SIG_BASE_ANG_VEL: return x.fr[3 + k];
    case LG_SIG_PROJ_GRAVITY: return x.fr[6 + k];
    case LG_SIG_COMMANDS: return x.cmd[k];
    case LG_SIG_ROOT_POS: return P->buf.root_states[(size_t)i * 13 + k];
    case LG_SIG_TRAJ0: return P->buf.trajectory[(size_t)i * (c.traj.enabled ? c.traj.N : 1) * 2 + k];
    case LG_SIG_PREV_ERROR: return P->buf.prev_error[(size_t)i * 2 + k];
    case LG_SIG_DOF_POS_REL: return x.dof[2 * k] - c.default_dof_pos[k];
    case LG_SIG_DOF_VEL: return x.dof[2 * k + 1];
    case LG_SIG_TORQUES: return x.tau[k];
    case LG_SIG_ACTIONS: return x.act[k];
    case LG_SIG_LAST_ACTIONS: return x.lact[k];
    }
    return 0.0f;
}
// generic extra terms (legged_hip.h lg_xterm_kind): what a subclass of the reference writes as a _reward_<name> method
// (LR:605-629); tracking_rom LT:1060-1069 and differential_error LT:1100-1110 are declared through it
__device__ float xterm_value(const DevParams *P, int i, const lg_xterm &t, const RewardCtx &x) {
    float s = 0.0f;
    switch (t.kind) {
    case LG_XT_EXP_NEG_WSQ_ERR:
        for (int k = 0; k < t.n; ++k) {
            const float d = signal_at(P, i, t.sig_a, t.off_a + k, x) - signal_at(P, i, t.sig_b, t.off_b + k, x);
            s += d * d * t.w[k];
        }
        return expf(-s / t.p[0]);
    case LG_XT_WSQ:
        for (int k = 0; k < t.n; ++k) { const float a = signal_at(P, i, t.sig_a, t.off_a + k, x); s += t.w[k] * a * a; }
        return s;
    case LG_XT_SLOPED_ERR_CHANGE: {
        float pn = 0.0f;
        for (int k = 0; k < t.n; ++k) {
            const float d = signal_at(P, i, t.sig_a, t.off_a + k, x) - signal_at(P, i, t.sig_b, t.off_b + k, x);
            const float te = d * d, pc = signal_at(P, i, t.sig_c, t.off_c + k, x);
            s += te * te;
            pn += pc * pc;
        }
        const float diff = sqrtf(s) - sqrtf(pn);
        return (diff < 0.0f ? t.p[0] : t.p[1]) * diff;
    }
    }
    return 0.0f;
}

// reset_traj (LT:222-229) + the stale-trajectory error of LT:199 for one env; root already holds the post-reset pose
// Inlined into their callers: as separate functions they cost k_post_step a call frame in scratch (saved registers on the
// phase-A latency chain) and the callee's register interface -- post-step 17.3 -> 14.5 us flat, 29.2 -> 25.1 us rough terrain
// (profiles/r03_ab.txt).
__device__ __forceinline__ void reset_trajectory(const DevParams *P, int i, int64_t counter, int inject, float *__restrict__ win) {
    const lg_cfg &c = P->cfg;
    const int A = c.num_actions;
    const float *r = P->buf.root_states + (size_t)i * 13;
    float zx = r[0], zy = r[1];
    if (c.traj.randomize_rom_distance && uni(P, i, LG_TSLOT_ROMD(A), counter, inject) > c.traj.zero_rom_dist_llh) {
        zx += (c.traj.max_rom_dist[0] - (-c.traj.max_rom_dist[0])) * uni(P, i, LG_TSLOT_ROMD(A) + 1, counter, inject) + (-c.traj.max_rom_dist[0]);
        zy += (c.traj.max_rom_dist[1] - (-c.traj.max_rom_dist[1])) * uni(P, i, LG_TSLOT_ROMD(A) + 2, counter, inject) + (-c.traj.max_rom_dist[1]);
    }
    tg_reset(P, i, zx, zy, counter, inject, win);
    for (int k = 0; k < 2; ++k) {
        const float d = P->buf.trajectory[(size_t)i * c.traj.N * 2 + k] - r[k];
        P->buf.prev_error[(size_t)i * 2 + k] = d * d;
    }
}

// LR:147-187 (+:415-454, :463-486, AN:56-60) for one env
// win: LG_TG_WIN floats of LDS for this lane (trajectory env: the generator's window while it is rebuilt)
__device__ void reset_env(const DevParams *P, int i, int64_t counter, int inject, int init_done, float *__restrict__ win) {
    const lg_cfg &c = P->cfg;
    const int A = c.num_actions, N = c.num_envs;
    float *r = P->buf.root_states + (size_t)i * 13;
    float *org = P->buf.env_origins + (size_t)i * 3;
    float *cmd = P->buf.commands + (size_t)i * 4;
    if (c.curriculum && init_done) {
        float dx = r[0] - org[0], dy = r[1] - org[1];
        float dist = sqrtf(dx * dx + dy * dy);
        bool up = dist > c.terrain_env_length / 2.0f;
        float cn = sqrtf(cmd[0] * cmd[0] + cmd[1] * cmd[1]);
        bool down = (dist < cn * c.episode_length_s * 0.5f) && !up;
        int64_t lvl = P->buf.terrain_levels[i] + (up ? 1 : 0) - (down ? 1 : 0);
        if (lvl >= c.max_terrain_level) {
            if (inject) lvl = P->buf.inject_levels[i];
            else {
                lvl = (int64_t)(uni(P, i, LG_SLOT_LEVEL, counter, 0) * c.max_terrain_level);
                if (lvl > c.max_terrain_level - 1) lvl = c.max_terrain_level - 1;
            }
        } else if (lvl < 0) lvl = 0;
        P->buf.terrain_levels[i] = lvl;
        const float *to = P->terrain_origins + ((size_t)lvl * c.terrain_num_cols + P->buf.terrain_types[i]) * 3;
        org[0] = to[0]; org[1] = to[1]; org[2] = to[2];
    }
    const bool tj = c.traj.enabled;
    const int s_dof = tj ? LG_TSLOT_DOF : LG_SLOT_DOF, s_xy = tj ? LG_TSLOT_XY(A) : LG_SLOT_XY(A), s_vel = tj ? LG_TSLOT_VEL(A) : LG_SLOT_VEL(A);
    float2 *dof = reinterpret_cast<float2 *>(P->buf.dof_state) + (size_t)i * A;
    for (int j = 0; j < A; ++j) {
        float u = uni(P, i, s_dof + j, counter, inject);
        dof[j] = make_float2(c.default_dof_pos[j] * ((1.5f - 0.5f) * u + 0.5f), 0.0f);
    }
    for (int k = 0; k < 13; ++k) r[k] = c.base_init_state[k];
    for (int k = 0; k < 3; ++k) r[k] += org[k];
    if (c.custom_origins)
        for (int k = 0; k < 2; ++k) r[k] += (1.0f - (-1.0f)) * uni(P, i, s_xy + k, counter, inject) + (-1.0f);
    for (int k = 0; k < 6; ++k) r[7 + k] = (0.5f - (-0.5f)) * uni(P, i, s_vel + k, counter, inject) + (-0.5f);
    if (tj) reset_trajectory(P, i, counter, inject, win);
    else resample_commands(P, i, LG_SLOT_RCMD(A), counter, inject, c.cmd_lo, c.cmd_hi);
    for (int j = 0; j < A; ++j) { P->buf.last_actions[(size_t)i * A + j] = 0.0f; P->buf.last_dof_vel[(size_t)i * A + j] = 0.0f; }
    for (int f = 0; f < c.num_feet; ++f) P->buf.feet_air_time[(size_t)i * c.num_feet + f] = 0.0f;
    P->buf.episode_length[i] = 0;
    P->buf.reset[i] = 1;
    P->reset_mark[i] = 1;
    if (c.use_actuator_net) {
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int l = 0; l < 2; ++l)
            for (int j = 0; j < A; ++j) {
                size_t idx = ((size_t)l * N * A + (size_t)i * A + j) * 8;
                float4 *hp = reinterpret_cast<float4 *>(P->buf.lstm_h + idx), *cp = reinterpret_cast<float4 *>(P->buf.lstm_c + idx);
                hp[0] = z; hp[1] = z; cp[0] = z; cp[1] = z;
            }
    }
}

// reset_env for ONE environment carried out by the whole workgroup (the masked in-kernel reset of k_post_step: an env
// that resets would otherwise keep one lane busy for ~2.5 k instructions -- 20+ Philox draws, ~150 stores -- while the
// block's other lanes wait, and the kernel lasts as long as its slowest block).  Same arithmetic, same Philox slots, so
// the result is bit-identical to reset_env; only who computes what changes.  Call from all threads (contains barriers).
// WAVE = true: the same by ONE wave, no barriers (its lanes run in step; a workgroup-scope fence orders the global stores of one
// lane before the loads of another) -- the waves of k_post_step each take one of the tile's resetting envs, so a workgroup with
// two or three resets (Cassie under a random policy: 55 resets per step) lasts as long as one with a single reset.  The caller
// places a barrier after its last reset.
template <bool WAVE = false>
__device__ __forceinline__ void reset_env_coop(const DevParams *P, int i, int64_t counter, int inject, int init_done, float *__restrict__ win) {
    const lg_cfg &c = P->cfg;
    const int A = c.num_actions, N = c.num_envs, F = c.num_feet;
    const int tid = WAVE ? (int)(threadIdx.x & 63) : (int)threadIdx.x, nthr = WAVE ? 64 : (int)blockDim.x;
    auto sync = [&]() __attribute__((always_inline)) {
        if constexpr (WAVE) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        else __syncthreads();
    };
    float *r = P->buf.root_states + (size_t)i * 13;
    float *org = P->buf.env_origins + (size_t)i * 3;
    if (tid == 0 && c.curriculum && init_done) {                  // needs the pre-reset pose and commands
        float *cmd = P->buf.commands + (size_t)i * 4;
        float dx = r[0] - org[0], dy = r[1] - org[1];
        float dist = sqrtf(dx * dx + dy * dy);
        bool up = dist > c.terrain_env_length / 2.0f;
        float cn = sqrtf(cmd[0] * cmd[0] + cmd[1] * cmd[1]);
        bool down = (dist < cn * c.episode_length_s * 0.5f) && !up;
        int64_t lvl = P->buf.terrain_levels[i] + (up ? 1 : 0) - (down ? 1 : 0);
        if (lvl >= c.max_terrain_level) {
            if (inject) lvl = P->buf.inject_levels[i];
            else {
                lvl = (int64_t)(uni(P, i, LG_SLOT_LEVEL, counter, 0) * c.max_terrain_level);
                if (lvl > c.max_terrain_level - 1) lvl = c.max_terrain_level - 1;
            }
        } else if (lvl < 0) lvl = 0;
        P->buf.terrain_levels[i] = lvl;
        const float *to = P->terrain_origins + ((size_t)lvl * c.terrain_num_cols + P->buf.terrain_types[i]) * 3;
        org[0] = to[0]; org[1] = to[1]; org[2] = to[2];
    }
    sync();
    const bool tj = c.traj.enabled;
    const int s_dof = tj ? LG_TSLOT_DOF : LG_SLOT_DOF, s_xy = tj ? LG_TSLOT_XY(A) : LG_SLOT_XY(A), s_vel = tj ? LG_TSLOT_VEL(A) : LG_SLOT_VEL(A);
    for (int t = tid; t < A + 14 + F; t += nthr) {                 // one role per lane
        if (t < A) {                                               // joint j = t
            const float u = uni(P, i, s_dof + t, counter, inject);
            reinterpret_cast<float2 *>(P->buf.dof_state)[(size_t)i * A + t] = make_float2(c.default_dof_pos[t] * ((1.5f - 0.5f) * u + 0.5f), 0.0f);
            P->buf.last_actions[(size_t)i * A + t] = 0.0f;
            P->buf.last_dof_vel[(size_t)i * A + t] = 0.0f;
        } else if (t < A + 13) {                                   // root component k
            const int k = t - A;
            float v = c.base_init_state[k];
            if (k < 3) v += org[k];
            if (k < 2 && c.custom_origins) v += (1.0f - (-1.0f)) * uni(P, i, s_xy + k, counter, inject) + (-1.0f);
            if (k >= 7) v = (0.5f - (-0.5f)) * uni(P, i, s_vel + (k - 7), counter, inject) + (-0.5f);
            r[k] = v;
        } else if (t == A + 13) {
            if (!tj) resample_commands(P, i, LG_SLOT_RCMD(A), counter, inject, c.cmd_lo, c.cmd_hi);
            P->buf.episode_length[i] = 0;
            P->buf.reset[i] = 1;
            P->reset_mark[i] = 1;
        } else {
            P->buf.feet_air_time[(size_t)i * F + (t - A - 14)] = 0.0f;
        }
    }
    if (c.use_actuator_net) {                                      // h, c of both layers: 2 x A rows of 8 floats each
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int q = tid; q < 2 * A * 2; q += nthr) {        // q = (layer, joint, half)
            const int l = q / (2 * A), j = (q / 2) % A, hf = q & 1;
            const size_t idx = ((size_t)l * N * A + (size_t)i * A + j) * 8 + 4 * hf;
            *reinterpret_cast<float4 *>(P->buf.lstm_h + idx) = z;
            *reinterpret_cast<float4 *>(P->buf.lstm_c + idx) = z;
        }
    }
    sync();
    if (tj) {                                                      // needs the new root pose: after the barrier, one lane (a serial 10-step
        if (tid == 0) reset_trajectory(P, i, counter, inject, win);   // ROM integration; resets are rare)
        sync();
    }
}


// LR:877-915 (+ utils/math.py:38-42): one lane per (env, height point)
__device__ __forceinline__ float height_sample(const DevParams *P, const float *r, int h) {
    const lg_cfg &c = P->cfg;
    float qz = r[5], qw = r[6];
    float n = fmaxf(sqrtf(qz * qz + qw * qw), 1e-9f);
    float qy[4] = {0.0f, 0.0f, qz / n, qw / n};
    V3 w = quat_apply(qy, V3{P->height_points[2 * h], P->height_points[2 * h + 1], 0.0f});
    float x = (w.x + r[0] + c.border_size) / c.hf_hscale;
    float y = (w.y + r[1] + c.border_size) / c.hf_hscale;
    int px = (int)x, py = (int)y;                                 // .long(): truncation toward zero
    px = min(max(px, 0), c.hf_rows - 2);
    py = min(max(py, 0), c.hf_cols - 2);
    const int16_t *hs = P->height_samples + (size_t)px * c.hf_cols + py;
    int16_t h1 = hs[0], h2 = hs[c.hf_cols], h3 = hs[1];
    int16_t mn = h1 < h2 ? h1 : h2;
    mn = mn < h3 ? mn : h3;
    return (float)mn * c.hf_vscale;
}

// The body works on one tile of TILE environments starting at env0 and touches no other environment's state.
template <int TILE>
__device__ __forceinline__ void post_step_tile(const DevParams *__restrict__ P, const int env0, int64_t counter, int inject, int init_done,
                                               int push_now) {
    const lg_cfg &c = P->cfg;
    const int N = c.num_envs, A = c.num_actions, B = c.num_bodies, O = c.num_obs, H = c.num_height_points;
    const int nE = min(TILE, N - env0);
    const int tid = threadIdx.x;
    const int tile_id = env0 / TILE;
    __shared__ float s_acc[LG_NUM_TERMS];
    __shared__ int s_cnt, s_flt;
    __shared__ int s_list[TILE];                               // envs of this tile that reset this step
    if (tid < LG_NUM_TERMS) s_acc[tid] = 0.0f;
    if (tid == 0) { s_cnt = 0; s_flt = 0; }
#define STAMP(k) do { if (tid == 0 && tile_id < 64) P->dbg_cycles[tile_id * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
    STAMP(0);

    // ---- phase H: height scan of the pre-reset pose (LR:356-357)
    // What every scan point of an env shares -- the normalised yaw quaternion and the base position -- is computed once per env
    // into LDS and the table of scan points is staged there too: a point then costs its three height-sample gathers and nothing
    // else from memory (height_sample re-reads the root state and its table entry and repeats the sqrt / two divisions per point;
    // same arithmetic on the same values, so the heights are bit-identical).  The heights stay in LDS for phase O.
    constexpr int HP_LDS = 192;                                   // scan points staged (ANYmal 17 x 11 = 187, Cassie 121)
    __shared__ float s_hp[2 * HP_LDS];
    __shared__ float s_hr[TILE][4];
    __shared__ float s_mh[TILE * HP_LDS];
    const bool hstage = c.measure_heights && c.terrain_type == 1 && H <= HP_LDS;
    if (c.measure_heights) {
        if (hstage) {
            for (int t = tid; t < 2 * H; t += LG_TILE_THREADS) s_hp[t] = P->height_points[t];
            if (tid < nE) {
                const float *r = P->buf.root_states + (size_t)(env0 + tid) * 13;
                const float qz = r[5], qw = r[6];
                const float n = fmaxf(sqrtf(qz * qz + qw * qw), 1e-9f);
                s_hr[tid][0] = qz / n; s_hr[tid][1] = qw / n; s_hr[tid][2] = r[0]; s_hr[tid][3] = r[1];
            }
            __syncthreads();
        }
        // HU scan points per lane and trip: their 3 HU height-sample gathers are independent and in flight together (one lane
        // per point and trip exposed three dependent-latency round trips per trip to a lone wave)
        constexpr int HU = 4;                                     // scan points per lane and trip (8: slower, 25.7 k vs 21.3 k cycles at 187 points)
        for (int idx0 = tid; idx0 < nE * H; idx0 += HU * LG_TILE_THREADS) {
            float v[HU];
#pragma unroll
            for (int u = 0; u < HU; ++u) {
                const int idx = min(idx0 + u * LG_TILE_THREADS, nE * H - 1);
                const int el = idx / H, h = idx % H;
                if (hstage) {                                     // LR:877-915, the body of height_sample on the staged values
                    const float qy[4] = {0.0f, 0.0f, s_hr[el][0], s_hr[el][1]};
                    const V3 w = quat_apply(qy, V3{s_hp[2 * h], s_hp[2 * h + 1], 0.0f});
                    const float x = (w.x + s_hr[el][2] + c.border_size) / c.hf_hscale;
                    const float y = (w.y + s_hr[el][3] + c.border_size) / c.hf_hscale;
                    int px = (int)x, py = (int)y;
                    px = min(max(px, 0), c.hf_rows - 2);
                    py = min(max(py, 0), c.hf_cols - 2);
                    const int16_t *hs = P->height_samples + (size_t)px * c.hf_cols + py;
                    const int16_t h1 = hs[0], h2 = hs[c.hf_cols], h3 = hs[1];
                    int16_t mn = h1 < h2 ? h1 : h2;
                    mn = mn < h3 ? mn : h3;
                    v[u] = (float)mn * c.hf_vscale;
                } else {
                    v[u] = c.terrain_type == 1 ? height_sample(P, P->buf.root_states + (size_t)(env0 + el) * 13, h) : 0.0f;
                }
            }
#pragma unroll
            for (int u = 0; u < HU; ++u) {
                const int idx = idx0 + u * LG_TILE_THREADS;
                if (idx < nE * H) {
                    P->buf.measured_heights[(size_t)(env0 + idx / H) * H + idx % H] = v[u];
                    if (hstage) s_mh[(idx / H) * HP_LDS + idx % H] = v[u];
                }
            }
        }
    }
    // contact forces and the feet state of the tile, staged once (coalesced) for the per-env lane of phase A
    __shared__ float s_cf[TILE * LG_MAX_BODIES * 3];
    __shared__ float s_air[TILE * LG_MAX_FEET];
    __shared__ uint8_t s_lc[TILE * LG_MAX_FEET];
    {
        const int F = c.num_feet, nb = B * 3;
        const float *gcf = P->buf.contact_forces + (size_t)env0 * nb;
        for (int idx = tid; idx < nE * nb; idx += LG_TILE_THREADS) s_cf[(idx / nb) * (LG_MAX_BODIES * 3) + idx % nb] = gcf[idx];
        for (int idx = tid; idx < nE * F; idx += LG_TILE_THREADS) {
            s_air[(idx / F) * LG_MAX_FEET + idx % F] = P->buf.feet_air_time[(size_t)env0 * F + idx];
            s_lc[(idx / F) * LG_MAX_FEET + idx % F] = P->buf.last_contacts[(size_t)env0 * F + idx];
        }
    }
    __syncthreads();
    STAMP(1);

    // ---- phase A (LR:111-129,139-145,189-206).  16 lanes per env (the workgroup is 16 envs x 16 lanes, one DPP row each):
    //   A1  every joint-summed reward ingredient on a lane per joint, reduced over the row (a lone lane per env walked each
    //       sum joint by joint, one exposed global-load latency per iteration);
    //   A2  lane 0 of the row: frames, commands / trajectory generator, pushes, termination and the remaining terms; all
    //       loads first, term values into LDS, stores last -- a store to a state buffer in the middle makes every later
    //       load wait for it (the buffers may alias as far as the compiler knows);
    //   A3  all lanes: episode sums (+ the logging sums of the envs that reset).
    static_assert(TILE * 16 <= LG_TILE_THREADS && LG_TILE_THREADS % 64 == 0, "phase A: 16 lanes per environment on the first TILE rows of lanes");
    static_assert(LG_TILE_THREADS / 64 <= TILE, "phase R: one generator window (s_win row) per wave");
    __shared__ float s_tv[TILE][LG_NUM_TERMS];
    __shared__ float s_fr[TILE][9];
    __shared__ float s_win[TILE][LG_TG_WIN];                          // trajectory env: the generator window of each env's phase-A lane
    __shared__ float s_js[TILE][JS_N];
    __shared__ uint8_t s_rst[TILE];
    const int e16 = tid >> 4, l16 = tid & 15;
    if (e16 < TILE) {                                                 // (wave-uniform: TILE rows of 16 lanes are whole waves)
        const int i = env0 + min(e16, nE - 1);
        float pj[JS_N];
#pragma unroll
        for (int k = 0; k < JS_N; ++k) pj[k] = 0.0f;
        if (l16 < A) {
            const size_t ij = (size_t)i * A + l16;
            const float2 st = reinterpret_cast<const float2 *>(P->buf.dof_state)[ij];
            const float tau = P->buf.torques[ij], act = P->buf.actions[ij], lact = P->buf.last_actions[ij], lqd = P->buf.last_dof_vel[ij];
            const float q = st.x, qd = st.y, acc = (lqd - qd) / c.dt, da = lact - act;
            pj[JS_TORQUES] = tau * tau;
            pj[JS_DOF_VEL] = qd * qd;
            pj[JS_DOF_ACC] = acc * acc;
            pj[JS_ACTION_RATE] = da * da;
            pj[JS_DOF_POS_LIMITS] = -fminf(q - c.dof_pos_limits[l16][0], 0.0f) + fmaxf(q - c.dof_pos_limits[l16][1], 0.0f);
            pj[JS_DOF_VEL_LIMITS] = clampf(fabsf(qd) - c.dof_vel_limits[l16] * c.soft_dof_vel_limit, 0.0f, 1.0f);
            pj[JS_TORQUE_LIMITS] = fmaxf(fabsf(tau) - c.torque_limits[l16] * c.soft_torque_limit, 0.0f);
            pj[JS_STAND_STILL] = fabsf(q - c.default_dof_pos[l16]);
        }
#pragma unroll
        for (int k = 0; k < JS_N; ++k) pj[k] = row16_sum(pj[k]);
        if (l16 == 0) {
#pragma unroll
            for (int k = 0; k < JS_N; ++k) s_js[e16][k] = pj[k];
            s_rst[e16] = 0;
        }
    }
    if (l16 == 0 && e16 < nE) {
        const int i = env0 + e16;
        float *r = P->buf.root_states + (size_t)i * 13;
        float *cmd = P->buf.commands + (size_t)i * 4;
        const float *cf = s_cf + e16 * (LG_MAX_BODIES * 3);
        const int64_t ep = P->buf.episode_length[i] + 1;                   // LR:114
        RewardCtx x;
        x.js = s_js[e16];
        x.air = s_air + e16 * LG_MAX_FEET;
        x.lc = s_lc + e16 * LG_MAX_FEET;
        x.blv = quat_rotate_inverse(r + 3, V3{r[7], r[8], r[9]});        // LR:118-121
        x.bav = quat_rotate_inverse(r + 3, V3{r[10], r[11], r[12]});
        x.pg = quat_rotate_inverse(r + 3, V3{0.0f, 0.0f, -1.0f});
        {
            float *f = s_fr[e16];
            f[0] = x.blv.x; f[1] = x.blv.y; f[2] = x.blv.z; f[3] = x.bav.x; f[4] = x.bav.y; f[5] = x.bav.z;
            f[6] = x.pg.x; f[7] = x.pg.y; f[8] = x.pg.z;
            x.fr = f;
        }
        bool rst = false;                                                   // LR:139-145 (contact forces do not change below)
        for (int b = 0; b < c.num_term; ++b) rst |= fnorm3(cf + 3 * c.term_idx[b]) > 1.0f;
        const bool flt = P->fault[i] != 0;                                  // physics fault guard (lg_physics.h)
        const bool to = ep > c.max_episode_length;
        rst = rst || flt || to;
        if (c.traj.enabled) tg_callback_step(P, i, counter, inject, s_win[e16]);   // LT:405-417
        else if (ep % c.resample_steps == 0) resample_commands(P, i, LG_SLOT_CMD, counter, inject, P->cb.cmd_lo, P->cb.cmd_hi);   // LR:348-350
        if (c.heading_command && !c.traj.enabled) {                         // LR:351-354, math.py:45-48
            V3 fwd = quat_apply(r + 3, V3{1.0f, 0.0f, 0.0f});
            float heading = atan2f(fwd.y, fwd.x);
            const float two_pi = 6.283185307179586f;
            float ang = fmodf(cmd[3] - heading, two_pi);
            if (ang < 0.0f) ang += two_pi;
            if (ang > 3.141592653589793f) ang -= two_pi;
            cmd[2] = clampf(0.5f * ang, -1.0f, 1.0f);
        }
        if (c.traj.enabled) {                                               // LT:150-160,483-486: per-env push timers
            float tm = P->buf.push_timer[i] - c.dt;
            if (tm <= 0.0f) {
                const float mv = c.traj.max_push_vel_xy;
                r[7] = (mv - (-mv)) * uni(P, i, LG_TSLOT_PUSH, counter, inject) + (-mv);
                r[8] = (mv - (-mv)) * uni(P, i, LG_TSLOT_PUSH + 1, counter, inject) + (-mv);
                tm = (c.traj.push_t_hi - c.traj.push_t_lo) * uni(P, i, LG_TSLOT_TIMER, counter, inject) + c.traj.push_t_lo;
            }
            P->buf.push_timer[i] = tm;
        } else if (push_now) {                                              // LR:358-359,456-461 (the period check is the host's: lg_stage.push_time)
            const float mv = P->cb.max_push_vel;
            r[7] = (mv - (-mv)) * uni(P, i, LG_SLOT_PUSH, counter, inject) + (-mv);
            r[8] = (mv - (-mv)) * uni(P, i, LG_SLOT_PUSH + 1, counter, inject) + (-mv);
        }
        x.cmd = cmd; x.cf = cf; x.root_z = r[2]; x.reset = rst; x.time_out = to;
        x.dof = P->buf.dof_state + (size_t)i * A * 2;
        x.tau = P->buf.torques + (size_t)i * A; x.act = P->buf.actions + (size_t)i * A;
        x.lact = P->buf.last_actions + (size_t)i * A; x.lqd = P->buf.last_dof_vel + (size_t)i * A;
        float rew = 0.0f;                                                   // LR:189-206
        for (int o = 0; o < c.num_terms; ++o) {                             // active builtin and extra terms, alphabetical (LR:605-629)
            const int k = c.term_order[o];
            float v;
            if (k < LG_NUM_REWARDS) v = reward_term(P, i, k, x) * c.rew_scale[k];
            else v = xterm_value(P, i, c.xterms[k - LG_NUM_REWARDS], x) * c.xterms[k - LG_NUM_REWARDS].scale;
            rew += v;
            s_tv[e16][k] = v;
        }
        if (c.only_positive_rewards) rew = fmaxf(rew, 0.0f);
        if (c.rew_scale[LG_REW_TERMINATION] != 0.0f) {
            float v = reward_term(P, i, LG_REW_TERMINATION, x) * c.rew_scale[LG_REW_TERMINATION];
            rew += v;
            s_tv[e16][LG_REW_TERMINATION] = v;
        }
        // ---- stores of this phase
        P->buf.episode_length[i] = ep;
        float *o3 = P->buf.base_lin_vel + 3 * (size_t)i;
        o3[0] = x.blv.x; o3[1] = x.blv.y; o3[2] = x.blv.z;
        o3 = P->buf.base_ang_vel + 3 * (size_t)i;
        o3[0] = x.bav.x; o3[1] = x.bav.y; o3[2] = x.bav.z;
        o3 = P->buf.projected_gravity + 3 * (size_t)i;
        o3[0] = x.pg.x; o3[1] = x.pg.y; o3[2] = x.pg.z;
        P->buf.time_out[i] = to;
        P->buf.reset[i] = rst;
        P->buf.rew[i] = rew;
        if (flt) { P->fault[i] = 0; atomicAdd(&s_flt, 1); }
        if (rst) { s_list[atomicAdd(&s_cnt, 1)] = i; s_rst[e16] = 1; }    // LR:147-187
    }
    __syncthreads();
    if (c.rew_scale[LG_REW_FEET_AIR_TIME] != 0.0f)                        // the term's state, back to HBM (a reset below clears its rows)
        for (int idx = tid; idx < nE * c.num_feet; idx += LG_TILE_THREADS) {
            P->buf.feet_air_time[(size_t)env0 * c.num_feet + idx] = s_air[(idx / c.num_feet) * LG_MAX_FEET + idx % c.num_feet];
            P->buf.last_contacts[(size_t)env0 * c.num_feet + idx] = s_lc[(idx / c.num_feet) * LG_MAX_FEET + idx % c.num_feet];
        }
    // A3: episode_sums += term value (LR:196-197,205); the sums of resetting envs go to the logging accumulators and are cleared
    for (int idx = tid; idx < nE * (c.num_terms + 1); idx += LG_TILE_THREADS) {
        const int e2 = idx / (c.num_terms + 1), o = idx % (c.num_terms + 1);
        if (o == c.num_terms && c.rew_scale[LG_REW_TERMINATION] == 0.0f) continue;
        const int k = o < c.num_terms ? c.term_order[o] : LG_REW_TERMINATION;
        float *es = P->buf.episode_sums + (size_t)k * N + env0 + e2;
        const float v = *es + s_tv[e2][k];
        if (s_rst[e2]) { atomicAdd(&s_acc[k], v); *es = 0.0f; }
        else *es = v;
    }
    __syncthreads();
    STAMP(2);
    for (int q = tid >> 6; q < s_cnt; q += LG_TILE_THREADS / 64)      // one wave per resetting env (wave-uniform trip count)
        reset_env_coop<true>(P, s_list[q], counter, inject, init_done, s_win[tid >> 6]);
    if (s_cnt > 0) __syncthreads();                                   // workgroup-uniform: phase O reads what the resets wrote
    STAMP(3);
    if (s_cnt > 0) {
        if (tid < LG_NUM_TERMS && term_scale(c, tid) != 0.0f) atomicAdd(P->ep_accum + tid, s_acc[tid]);
        if (tid == 0) { atomicAdd(P->reset_count, s_cnt); *P->any_reset_step = counter; }
        if (tid == 0 && s_flt > 0) atomicAdd(P->fault_count, s_flt);
    }

    // ---- phase O: observations (LR:208-226), clip (LR:100-103), bookkeeping (LR:132-134)
    const int ob = c.traj.enabled ? 9 + 2 * c.traj.N : 12;              // first joint entry: after commands[:3] | the trajectory block
    const int s_noise = c.traj.enabled ? LG_TSLOT_NOISE(A) : LG_SLOT_NOISE(A);
    // Segment by segment (frame block | joint positions | joint rates | actions | heights), four entries per lane and trip with
    // their operand loads issued together: one generic loop over all entries took a different branch -- and exposed one more
    // global-load latency to the lone wave -- for every kind of entry its 64 lanes happened to hold.
    // Noise: slot s_noise + k of the env's Philox stream per entry.  One Philox evaluation yields the uniforms of four consecutive
    // slots (philox_uniform4), so the noiseless entries are staged in LDS by the segment loops and a second pass walks the stream
    // block by block: a quarter of the evaluations of one per entry (rough terrain: 235 entries x 16 envs per workgroup).
    constexpr int OBS_LDS = 256;                                        // widest observation row staged (rough terrain: 235)
    __shared__ float s_ob[TILE * OBS_LDS];
    __shared__ float s_nv[OBS_LDS];                                     // the noise scale vector: one global round trip for all trips
    const bool stage = c.add_noise && O <= OBS_LDS;
    if (stage)
        for (int k = tid; k < O; k += LG_TILE_THREADS) s_nv[k] = P->noise_vec[k];   // (the barrier before the noise pass orders it)
    auto emit = [&](int i, int k, float v) {
        if (stage) { s_ob[(i - env0) * OBS_LDS + k] = v; return; }
        if (c.add_noise) v += (2.0f * uni(P, i, s_noise + k, counter, inject) - 1.0f) * P->noise_vec[k];
        P->buf.obs[(size_t)i * O + k] = clampf(v, -c.clip_obs, c.clip_obs);
    };
    for (int idx = tid; idx < nE * ob; idx += LG_TILE_THREADS) {           // lin vel, ang vel, gravity, commands | trajectory
        const int i = env0 + idx / ob, k = idx % ob;
        const float *r = P->buf.root_states + (size_t)i * 13;
        float v;
        if (k < 9) {
            const float *src = k < 3 ? P->buf.base_lin_vel : k < 6 ? P->buf.base_ang_vel : P->buf.projected_gravity;
            v = src[3 * (size_t)i + k % 3] * (k < 3 ? c.obs_scale_lin_vel : k < 6 ? c.obs_scale_ang_vel : 1.0f);
        } else if (c.traj.enabled) {
            v = (P->buf.trajectory[(size_t)i * 2 * c.traj.N + (k - 9)] - r[(k - 9) & 1]) * c.traj.obs_scale[(k - 9) & 1];   // LT:280-288
        } else {
            v = P->buf.commands[(size_t)i * 4 + k - 9] * (k == 11 ? c.obs_scale_ang_vel : c.obs_scale_lin_vel);
        }
        emit(i, k, v);
    }
    for (int idx = tid; idx < nE * A; idx += LG_TILE_THREADS) {            // joint positions, rates, actions: one lane per (env, joint)
        const int i = env0 + idx / A, j = idx % A;
        const float2 st = reinterpret_cast<const float2 *>(P->buf.dof_state)[(size_t)i * A + j];
        const float act = P->buf.actions[(size_t)i * A + j];
        emit(i, ob + j, (st.x - c.default_dof_pos[j]) * c.obs_scale_dof_pos);
        emit(i, ob + A + j, st.y * c.obs_scale_dof_vel);
        emit(i, ob + 2 * A + j, act);
    }
    STAMP(5);
    if (c.measure_heights)
        for (int idx0 = tid; idx0 < nE * H; idx0 += 4 * LG_TILE_THREADS) { // height entries, four in flight
            float hv[4], rz[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = min(idx0 + u * LG_TILE_THREADS, nE * H - 1);
                hv[u] = hstage ? s_mh[(idx / H) * HP_LDS + idx % H] : P->buf.measured_heights[(size_t)(env0 + idx / H) * H + idx % H];
                rz[u] = P->buf.root_states[(size_t)(env0 + idx / H) * 13 + 2];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = idx0 + u * LG_TILE_THREADS;
                if (idx < nE * H) emit(env0 + idx / H, ob + 3 * A + idx % H, clampf(rz[u] - 0.5f - hv[u], -1.0f, 1.0f) * c.obs_scale_height);
            }
        }
    STAMP(6);
    if (stage) {
        __syncthreads();
        const int b0 = s_noise >> 2, nb = ((s_noise + O + 3) >> 2) - b0;  // Philox blocks that hold the noise slots
        for (int idx = tid; idx < nE * nb; idx += LG_TILE_THREADS) {   // (two blocks per lane and trip: no faster -- the Philox rounds
            const int e = idx / nb, b = b0 + idx % nb, i = env0 + e;   //  are quarter-rate multiplies, throughput-bound on the SIMD)
            float u4[4] = {0.f, 0.f, 0.f, 0.f};
            if (!inject) philox_uniform4(c.seed, (uint32_t)(c.env_offset + i), (uint64_t)counter, (uint32_t)b, u4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = 4 * b + r - s_noise;
                if (k < 0 || k >= O) continue;
                const float u = inject ? P->buf.inject_uniforms[(size_t)i * P->K + 4 * b + r] : u4[r];
                float v = s_ob[e * OBS_LDS + k];
                v += (2.0f * u - 1.0f) * s_nv[k];
                P->buf.obs[(size_t)i * O + k] = clampf(v, -c.clip_obs, c.clip_obs);
            }
        }
    }
    STAMP(7);
    for (int idx = tid; idx < nE * A; idx += LG_TILE_THREADS) {
        const size_t ij = (size_t)env0 * A + idx;
        P->buf.last_actions[ij] = P->buf.actions[ij];
        P->buf.last_dof_vel[ij] = P->buf.dof_state[ij * 2 + 1];
    }
    for (int idx = tid; idx < nE * 6; idx += LG_TILE_THREADS) {
        const int i = env0 + idx / 6, k = idx % 6;
        P->buf.last_root_vel[(size_t)i * 6 + k] = P->buf.root_states[(size_t)i * 13 + 7 + k];
    }
    STAMP(4);
#undef STAMP
}

// Measured and not kept: __attribute__((amdgpu_waves_per_eu(1, 2))) (a 256-VGPR budget: the launch is one wave per SIMD at 4096
// envs).  It removes the last scratch bytes but the allocator then parks the trajectory window in LDS (+32 KB) and the kernel
// schedules for occupancy 2: 26.4 vs 14.5 us flat, 38.4 vs 25.1 us rough (profiles/r03_ab.txt).
template <int TILE>
__global__ void __launch_bounds__(LG_TILE_THREADS) k_post_step(const DevParams *__restrict__ P, int64_t counter, int inject,
                                                               int init_done, int push_now) {
    post_step_tile<TILE>(P, (int)blockIdx.x * TILE, counter, inject, init_done, push_now);
}

// A curriculum stage (legged_hip.h lg_stage) into the device constants.  what & 1: the values read after the step callback
// (cfg); what & 2: the values the callback reads (cb).  One lane; stream-ordered with the step kernels.
__global__ void k_set_stage(DevParams *P, lg_stage s, int what) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    lg_cfg &c = P->cfg;
    if (what & 1) {
        for (int k = 0; k < 4; ++k) { c.cmd_lo[k] = s.cmd_lo[k]; c.cmd_hi[k] = s.cmd_hi[k]; }
        c.max_push_vel = s.max_push_vel;
        for (int k = 0; k < LG_NUM_REWARDS; ++k) c.rew_scale[k] = s.rew_scale[k];
        for (int k = 0; k < LG_MAX_XTERMS; ++k) { c.xterms[k].scale = s.xterm_scale[k]; c.xterms[k].p[0] = s.xterm_p0[k]; }
        for (int k = 0; k < 2; ++k) { c.traj.v_min[k] = s.traj_v_min[k]; c.traj.v_max[k] = s.traj_v_max[k]; c.traj.max_rom_dist[k] = s.traj_max_rom_dist[k]; }
        c.traj.t_low = s.traj_t_low; c.traj.t_high = s.traj_t_high;
    }
    if (what & 2) {
        StageCb &b = P->cb;
        for (int k = 0; k < 4; ++k) { b.cmd_lo[k] = s.cmd_lo[k]; b.cmd_hi[k] = s.cmd_hi[k]; }
        b.max_push_vel = s.max_push_vel;
        for (int k = 0; k < 2; ++k) { b.v_min[k] = s.traj_v_min[k]; b.v_max[k] = s.traj_v_max[k]; }
        b.t_low = s.traj_t_low; b.t_high = s.traj_t_high;
    }
}

__global__ void __launch_bounds__(256) k_finalize(const DevParams *__restrict__ P, int accumulate) { finalize_body(P, accumulate); }

// Trajectory env, after the post-step (or lg_reset_ids) and before k_finalize: on a step where some env reset, the reference's
// generator re-checks the hold time of EVERY env in its reset loop (lg_traj.h, tg_late_resample).  One lane per env.
__global__ void __launch_bounds__(256) k_traj_late(const DevParams *__restrict__ P, int64_t counter, int inject) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P->cfg.num_envs || *P->reset_count <= 0) return;
    if (!P->reset_mark[i]) tg_late_resample(P, i, counter, inject);
}

// reset_idx(env_ids) for a caller-given subset (LR:147-187): one workgroup per id.  The episode-sum means of the
// subset go through the same accumulators as the in-step resets and k_finalize publishes them.
__global__ void __launch_bounds__(LG_TILE_THREADS) k_reset_ids(const DevParams *__restrict__ P, const int32_t *__restrict__ ids, int n,
                                                                int64_t counter, int inject, int init_done) {
    const lg_cfg &c = P->cfg;
    const int N = c.num_envs, tid = threadIdx.x;
    __shared__ float s_win[LG_TG_WIN];
    for (int q = blockIdx.x; q < n; q += gridDim.x) {         // workgroup-uniform
        const int i = ids[q];
        if (i < 0 || i >= N) continue;
        if (tid < LG_NUM_TERMS && term_scale(c, tid) != 0.0f) {
            atomicAdd(P->ep_accum + tid, P->buf.episode_sums[(size_t)tid * N + i]);
            P->buf.episode_sums[(size_t)tid * N + i] = 0.0f;
        }
        if (tid == 0) { atomicAdd(P->reset_count, 1); *P->any_reset_step = counter; }
        reset_env_coop(P, i, counter, inject, init_done, s_win);
    }
}

// reset_idx(arange(N)) (base_task.py:113): no logging
__global__ void __launch_bounds__(64) k_reset_all(const DevParams *__restrict__ P, int64_t counter, int inject, int init_done) {
    __shared__ float s_win[64][LG_TG_WIN];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P->cfg.num_envs) return;
    for (int k = 0; k < LG_NUM_TERMS; ++k) P->buf.episode_sums[(size_t)k * P->cfg.num_envs + i] = 0.0f;
    reset_env(P, i, counter, inject, init_done, s_win[threadIdx.x]);
    P->reset_mark[i] = 0;
}

// ------------------------------------------------------------------------------------------------ launchers
extern "C" void lgk_set_actions(const DevParams *P, const float *a, int n, hipStream_t s) {
    int blocks = (n + 255) / 256;
    hipLaunchKernelGGL(k_set_actions, dim3(blocks > 1024 ? 1024 : blocks), dim3(256), 0, s, P, a);
}
// The one-lane-per-leg physics (lg_physics.h) stays selectable for A/B runs and for the equivalence test of the two lane maps:
// LG_PHYS_PAIR=0 in the environment, or lg_debug_set_phys_pair() at run time.  Default: the pair-lane map.
static int g_phys_pair = -1;
extern "C" void lgk_debug_set_phys_pair(int v) { g_phys_pair = v ? 1 : 0; }
static int phys_pair_enabled() {
    if (g_phys_pair < 0) { const char *e = getenv("LG_PHYS_PAIR"); g_phys_pair = e ? (atoi(e) ? 1 : 0) : 1; }
    return g_phys_pair;
}
// LG_SUBSTEPS_NW (pair-lane physics only): 4 = blocks of 4 waves and 64/L envs, 2 = blocks of 2 waves and 32/L envs; 0 / unset = by
// topology.  Measured per lg_step: quadruped 117.2 (4) vs 122.4 us (2: two blocks share a CU and its LDS / instruction cache for
// nothing, the chain per wave is the same); biped 148.7 (4: 128 blocks, half the CUs idle) vs 145.5 us (2: 256 blocks).  Results are
// bit-identical either way (tests/test_hip_env.py::test_control_loop_block_shapes_are_bit_identical).
static int g_substeps_nw = -1;
static int g_substeps_occ = 0;        // 0: by grid size; 1 / 2: force the build for that many waves per SIMD (tests)
extern "C" void lgk_debug_set_substeps_occ(int v) { g_substeps_occ = v == 1 || v == 2 ? v : 0; }
extern "C" void lgk_debug_set_substeps_nw(int v) { g_substeps_nw = v == 2 ? 2 : v == 4 ? 4 : 0; }
template <int L, int J, bool LSTM>
static void launch_substeps(int N, const DevParams *P, const float *a_in, int mode, int iters, hipStream_t s) {
    if (g_substeps_nw < 0) { const char *e = getenv("LG_SUBSTEPS_NW"); g_substeps_nw = e ? (atoi(e) == 2 ? 2 : atoi(e) == 4 ? 4 : 0) : 0; }
    const int epw4 = 64 / L, epw2 = 32 / L;
    const int nw = g_substeps_nw ? g_substeps_nw : (L == 2 ? 2 : 4);
    if (phys_pair_enabled() && nw == 2)
        hipLaunchKernelGGL((k_substeps<L, J, LSTM, true, 2>), dim3((N + epw2 - 1) / epw2), dim3(128), 0, s, P, a_in, mode, iters);
    else if (phys_pair_enabled()) {
        static int cus = 0;
        if (!cus) { int dev = 0; hipDeviceProp_t pr; cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256; }
        const int grid = (N + epw4 - 1) / epw4;
        if (L == 4 && (g_substeps_occ ? g_substeps_occ == 2 : grid > cus))   // more workgroups than CUs: the 256-register build, two per CU (bit-identical results)
            hipLaunchKernelGGL((k_substeps<L, J, LSTM, true, 4, (L == 4 ? 2 : 1)>), dim3(grid), dim3(256), 0, s, P, a_in, mode, iters);
        else
            hipLaunchKernelGGL((k_substeps<L, J, LSTM, true>), dim3(grid), dim3(256), 0, s, P, a_in, mode, iters);
    }
    else hipLaunchKernelGGL((k_substeps<L, J, LSTM, false>), dim3((N + epw4 - 1) / epw4), dim3(256), 0, s, P, a_in, mode, iters);
}
extern "C" int lgk_substeps(const DevParams *P, const float *a_in, int N, int L, int J, int lstm, int mode, int iters, hipStream_t s) {
    if (L == 4 && J == 3 && lstm) launch_substeps<4, 3, true>(N, P, a_in, mode, iters, s);
    else if (L == 4 && J == 3) launch_substeps<4, 3, false>(N, P, a_in, mode, iters, s);
    else if (L == 2 && J == 6 && !lstm) launch_substeps<2, 6, false>(N, P, a_in, mode, iters, s);
    else return -1;
    return 0;
}
extern "C" void lgk_set_stage(DevParams *P, const lg_stage *st, int what, hipStream_t s) {
    hipLaunchKernelGGL(k_set_stage, dim3(1), dim3(64), 0, s, P, *st, what);
}
extern "C" void lgk_finalize(const DevParams *P, int accumulate, hipStream_t s) {
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(256), 0, s, P, accumulate);
}
// finalize = 0: the caller defers the single-workgroup epilogue (lg_ctx.defer_finalize)
extern "C" void lgk_post_step(const DevParams *P, int N, int64_t counter, int inject, int init_done, int traj, int push_now, int finalize,
                              hipStream_t s) {
    constexpr int TILE = 16;
    hipLaunchKernelGGL((k_post_step<TILE>), dim3((N + TILE - 1) / TILE), dim3(LG_TILE_THREADS), 0, s, P, counter, inject, init_done, push_now);
    if (traj) hipLaunchKernelGGL(k_traj_late, dim3((N + 255) / 256), dim3(256), 0, s, P, counter, inject);
    if (finalize) hipLaunchKernelGGL(k_finalize, dim3(1), dim3(256), 0, s, P, 1);
}
extern "C" void lgk_reset_all(const DevParams *P, int N, int64_t counter, int inject, int init_done, hipStream_t s) {
    hipLaunchKernelGGL(k_reset_all, dim3((N + 63) / 64), dim3(64), 0, s, P, counter, inject, init_done);
}
extern "C" void lgk_reset_ids(const DevParams *P, const int32_t *ids, int n, int N, int64_t counter, int inject, int init_done, int traj,
                              hipStream_t s) {
    hipLaunchKernelGGL(k_reset_ids, dim3(n < 1024 ? n : 1024), dim3(LG_TILE_THREADS), 0, s, P, ids, n, counter, inject, init_done);
    if (traj) hipLaunchKernelGGL(k_traj_late, dim3((N + 255) / 256), dim3(256), 0, s, P, counter, inject);
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(256), 0, s, P, 0);
}
