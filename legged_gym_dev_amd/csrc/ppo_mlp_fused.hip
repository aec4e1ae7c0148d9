// Rollout forward of both MLPs (PPO.act: actor means + critic value for every env, rsl_rl ActorCritic.act /
// evaluate; call site lg_ppo_act) as ONE launch instead of one GEMM launch per layer: at 4096 rows the per-layer
// GEMMs are launch/latency-bound (4 dependent launches of ~13 us for 3 GFLOP).
//
// Workgroup = 32 rows of one net (grid.y: 0 actor, 1 critic), 4 waves.  The activations of the 32 rows never leave
// LDS (two fp32 images [32][K + 4]: K + 4 = 4 x odd keeps the 16-lane groups of ds_read_b128 on distinct 16-B
// slots).  Per layer every wave owns the column tiles n = 32 * (wave + 4 i): the A fragment (32 rows x 16 k) is read
// from LDS and split into its three bf16 terms in registers (same exact split as the GEMM staging), the B fragments
// come straight from the optimiser's bf16 weight planes in L2 (W [n][k]: 16 B per lane per plane, two k-steps
// prefetched), six v_mfma_f32_32x32x16_bf16 per fp32 product block as in k_gemm.  Bias + activation in registers,
// result back to LDS for the next layer; only the head outputs (means, value) go to HBM.
#include "ppo_device.h"
#include "ppo_mlp_args.h"
#include "lg_finalize.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));


__device__ __forceinline__ uint32_t mlp_cvt_pk_bf16(float a, float b) {
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// 8 consecutive floats -> the three bf16x8 terms (h, m, l), x = h + m + l exactly (see k_gemm's split2)
__device__ __forceinline__ void mlp_split8(const float (&x)[8], bf16x8 &h, bf16x8 &m, bf16x8 &l) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    uint32_t hh[4], mm[4], ll[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float x0 = x[2 * i], x1 = x[2 * i + 1];
        hh[i] = mlp_cvt_pk_bf16(x0, x1);
        f32x2 r = f32x2{x0, x1} - f32x2{__uint_as_float(hh[i] << 16), __uint_as_float(hh[i] & 0xffff0000u)};
        mm[i] = mlp_cvt_pk_bf16(r.x, r.y);
        r -= f32x2{__uint_as_float(mm[i] << 16), __uint_as_float(mm[i] & 0xffff0000u)};
        ll[i] = mlp_cvt_pk_bf16(r.x, r.y);
    }
    h = __builtin_bit_cast(bf16x8, make_uint4(hh[0], hh[1], hh[2], hh[3]));
    m = __builtin_bit_cast(bf16x8, make_uint4(mm[0], mm[1], mm[2], mm[3]));
    l = __builtin_bit_cast(bf16x8, make_uint4(ll[0], ll[1], ll[2], ll[3]));
}
__device__ __forceinline__ float mlp_act(int code, float v) {
    switch (code) {
    case 1: return v > 0.f ? v : __expf(v) - 1.0f;
    case 2: return v > 0.f ? 1.0507009873554804934193349852946f * v
                           : 1.0507009873554804934193349852946f * 1.6732632423543772848170429916717f * (__expf(v) - 1.0f);
    case 3: return fmaxf(v, 0.f);
    case 4: return v > 0.f ? v : 0.01f * v;
    case 5: return 2.0f * __frcp_rn(1.0f + __expf(-2.0f * v)) - 1.0f;
    case 6: return __frcp_rn(1.0f + __expf(-v));
    default: return v;
    }
}

#define MLP_ROWS 32
#ifndef MLP_NW
#define MLP_NW 4                 // waves per workgroup
#endif
#define MLP_NT (64 * MLP_NW)
#define MLP_KMAX 512
__host__ __device__ __forceinline__ constexpr int mlp_kpad(int K) { return (K + 15) & ~15; }
// Weight image of this kernel ("fragment order", built by k_mlp_frag_build from the fp32 parameters once per rollout): for
// layer (z, l), column tile t (32 outputs) and k-step ks (16 inputs) the three planes h, m, l are 1 KB blocks
//      frag_off[z][l] + ((t * K/16 + ks) * 3 + plane) * 512 + lane * 8 + j   =   plane of W[32 t + (lane & 31)][16 ks + 8 (lane >> 5) + j]
// i.e. exactly the B operand of v_mfma_f32_32x32x16_bf16 for that tile and k-step, lane by lane.  A wave's fragment load is
// one fully coalesced 1 KB read straight into the MFMA operand registers: 8 cache lines per instruction.  Reading the same
// fragments out of the row-major planes (W[n][k], one 16-byte piece of 32 different rows per instruction) costs the L1 32 tag
// lookups per instruction and made the launch address-bound: 68 us per act() at [512,256,128] against 61 us for the four
// per-layer GEMMs.  Output rows past N (the head) are zero in the image.
struct BFrag { bf16x8 p[3]; };
__device__ __forceinline__ BFrag mlp_load_frag(const uint16_t *__restrict__ wf, int nks, int tile, int ks, int lane) {
    BFrag f;
    const uint16_t *q = wf + ((size_t)(tile * nks + ks) * 3) * 512 + lane * 8;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) f.p[pl] = *reinterpret_cast<const bf16x8 *>(q + pl * 512);
    return f;
}

// One layer for one wave: CT column tiles (tile = wave + 4 (t0 + i)) advanced together along k, so the A fragment is read
// from LDS and split once per k-step.  k is walked in blocks of S k-steps (S CT = 4: twelve 1 KB loads per block and wave, two
// blocks in flight = 24 KB per wave, enough to cover the L2 round trip at the CU's fill rate).
template <int CT, int S, int ACT>
__device__ __forceinline__ void mlp_layer(const MlpArgs &g, int z, int l, const float *__restrict__ cur, float *__restrict__ nxt, int row0,
                                          int wave, int t0, int li, int lk) {
    // K: the layer's input width rounded up to whole k-steps (only an observation width can be off the grid: 235 rough terrain,
    // 169 Cassie, 65 trajectory task; the pad columns are zero in the LDS image and in the weight image)
    const int K = mlp_kpad(g.dims[z][l]), N = g.dims[z][l + 1], ld = K + 4, ldo = N + 4, nks = K / 16, nkb = (nks + S - 1) / S;
    const int lane = li + 32 * lk;
    const bool head = l == g.nl - 1;
    const uint16_t *__restrict__ wf = g.wfrag + g.frag_off[z][l];
    const float *__restrict__ arow = cur + li * ld + 8 * lk;
    int tile[CT], n[CT];
    f32x16 acc[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        tile[c] = wave + MLP_NW * (t0 + c);
        n[c] = tile[c] * 32 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    }
    BFrag bf[2][S][CT];
    auto load_block = [&](BFrag (&dst)[S][CT], int kb) {
#pragma unroll
        for (int q = 0; q < S; ++q)
#pragma unroll
            for (int c = 0; c < CT; ++c) dst[q][c] = mlp_load_frag(wf, nks, tile[c], min(S * kb + q, nks - 1), lane);
    };
    auto mul_block = [&](const BFrag (&src)[S][CT], int kb) {
#pragma unroll
        for (int q = 0; q < S; ++q) {
            if (S * kb + q >= nks) break;                              // tail of a K that is not a whole number of blocks (uniform)
            float x[8];
            *reinterpret_cast<float4 *>(x) = *reinterpret_cast<const float4 *>(arow + 16 * (S * kb + q));
            *reinterpret_cast<float4 *>(x + 4) = *reinterpret_cast<const float4 *>(arow + 16 * (S * kb + q) + 4);
            bf16x8 ah, am, al;
            mlp_split8(x, ah, am, al);
            // the six term products, smallest first; column tiles interleaved so that consecutive MFMAs never share an accumulator
            const bf16x8 *ax[3] = {&ah, &am, &al};
            constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PB[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int c = 0; c < CT; ++c)
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*ax[PA[t]], src[q][c].p[PB[t]], acc[c], 0, 0, 0);
        }
    };
    // sched_barrier: hipcc otherwise sinks the prefetch loads down to their uses and waits on each (vmcnt(0..5) all over
    // the loop), which exposes the L2 round trip once per k-step
    load_block(bf[0], 0);
    __builtin_amdgcn_sched_barrier(0);
    for (int kb = 0; kb < nkb; kb += 2) {
        load_block(bf[1], kb + 1 < nkb ? kb + 1 : 0);
        __builtin_amdgcn_sched_barrier(0);
        mul_block(bf[0], kb);
        __builtin_amdgcn_sched_barrier(0);
        if (kb + 1 < nkb) {
            load_block(bf[0], kb + 2 < nkb ? kb + 2 : 0);
            __builtin_amdgcn_sched_barrier(0);
            mul_block(bf[1], kb + 1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // acc[c][r]: col = lane&31 (n), row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        const float bv = g.params[g.b_off[z][l] + min(n[c], N - 1)];
        if (!head) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = acc[c][r] + bv;          // ACT 1: ELU compiled in (a per-element switch bloats the kernel past the I-cache)
                nxt[((r & 3) + 8 * (r >> 2) + 4 * lk) * ldo + n[c]] = ACT == 1 ? (v > 0.f ? v : __expf(v) - 1.0f) : mlp_act(g.act, v);
            }
        } else if (n[c] < N) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int lr = (r & 3) + 8 * (r >> 2) + 4 * lk, gr = row0 + lr;
                if (g.sample) nxt[lr * ldo + n[c]] = acc[c][r] + bv;     // stays in LDS for the sampling epilogue
                else if (gr < g.M) g.out[z][(size_t)gr * N + n[c]] = acc[c][r] + bv;
            }
        }
    }
}
// the column tiles of a wave, four / two / one at a time (S CT = 4)
template <int ACT>
__device__ __forceinline__ void mlp_layer_tiles(const MlpArgs &g, int z, int l, const float *cur, float *nxt, int row0, int wave, int mine,
                                                int li, int lk) {
    for (int t0 = 0; t0 < mine;) {
        const int c = mine - t0;
        if (c >= 4) { mlp_layer<4, 1, ACT>(g, z, l, cur, nxt, row0, wave, t0, li, lk); t0 += 4; }
        else if (c >= 2) { mlp_layer<2, 2, ACT>(g, z, l, cur, nxt, row0, wave, t0, li, lk); t0 += 2; }
        else { mlp_layer<1, 4, ACT>(g, z, l, cur, nxt, row0, wave, t0, li, lk); t0 += 1; }
    }
}

// fp32 parameters -> fragment-order image (one thread per 8 consecutive k of one output row and plane triple)
__global__ void __launch_bounds__(256) k_mlp_frag_build(MlpArgs g, uint16_t *__restrict__ wf) {
    for (int z = 0; z < 2; ++z)
        for (int l = 0; l < g.nl; ++l) {
            const int K = g.dims[z][l], N = g.dims[z][l + 1], nks = mlp_kpad(K) / 16, tiles = (N + 31) / 32;
            const float *__restrict__ W = g.params + g.w_off[z][l];
            uint16_t *__restrict__ dst = wf + g.frag_off[z][l];
            const int total = tiles * nks * 64;                       // (tile, ks, lane)
            for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
                const int lane = i & 63, ks = (i >> 6) % nks, tile = (i >> 6) / nks;
                const int n = tile * 32 + (lane & 31), k = 16 * ks + 8 * (lane >> 5);
                float x[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = (n < N && k + j < K) ? W[(size_t)n * K + min(k + j, K - 1)] : 0.f;
                bf16x8 h, m, lo;
                mlp_split8(x, h, m, lo);
                uint16_t *q = dst + ((size_t)(tile * nks + ks) * 3) * 512 + lane * 8;
                *reinterpret_cast<bf16x8 *>(q) = h;
                *reinterpret_cast<bf16x8 *>(q + 512) = m;
                *reinterpret_cast<bf16x8 *>(q + 1024) = lo;
            }
        }
}

// The previous policy step's epilogue, on workgroups beside the two MLPs (MlpArgs::pp).  Step s ran k_post_step; what follows it --
// the env's extras / counters and the learner's process_env_step -- depends on nothing this act computes and nothing here feeds
// the MLPs, so it shares the launch.  extras["time_outs"] is the reference's stale mask (legged_robot.py:156-157,186-187: refreshed
// only on a step with at least one reset): the env epilogue refreshes it exactly when any_reset_step == the step's counter, and in
// that case the process lanes read the fresh time_out mask instead, so they never read what the epilogue workgroup is writing.
__device__ __forceinline__ void rollout_epilogue(const MlpArgs &g, const PpoDev &P) {
    const DevParams *E = g.pp_env;
    const int nproc = (P.N + MLP_NT - 1) / MLP_NT;
    if ((int)blockIdx.x < nproc) {
        const int i = blockIdx.x * MLP_NT + threadIdx.x;
        if (i >= P.N) return;
        const bool fresh = *E->any_reset_step == g.pp_counter;
        const bool to = g.pp_use_tos && (fresh ? E->buf.time_out[i] : E->buf.extras_time_outs[i]) != 0;
        process_step_body(P, E->buf.rew[i], E->buf.reset[i] != 0, to, g.pp_t, i);
    } else if ((int)blockIdx.x == nproc && g.pp == 2) {
        finalize_body(E, 1);
    }
}

__global__ void __launch_bounds__(MLP_NT, 1) k_mlp_fwd(MlpArgs g, PpoDev P) {
    if (blockIdx.y == 2) { rollout_epilogue(g, P); return; }
    const int z = blockIdx.y;
    const int row0 = blockIdx.x * MLP_ROWS;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lk = lane >> 5;
    __shared__ __attribute__((aligned(16))) float s_a[MLP_ROWS * (MLP_KMAX + 4)];
    __shared__ __attribute__((aligned(16))) float s_b[MLP_ROWS * (MLP_KMAX + 4)];
    float *cur = s_a, *nxt = s_b;
#ifdef MLP_PROF                      // section timing (tools/mlp_sections.py): s_memtime deltas of wave 0 into P.noise
    unsigned long long t_last = __builtin_amdgcn_s_memtime();
    int t_k = 0;
#define MSTAMP() do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (tid == 0 && blockIdx.x < 32) P.noise[(blockIdx.y * 32 + blockIdx.x) * 8 + t_k] = (float)(t_ - t_last); t_last = t_; ++t_k; } while (0)
#else
#define MSTAMP() do { } while (0)
#endif
    {   // observations of the 32 rows -> LDS (rows past M read the last row: computed, never stored)
        const int K = g.dims[z][0], Kp = mlp_kpad(K), ld = Kp + 4;
        for (int i = tid; i < MLP_ROWS * (Kp - K); i += MLP_NT) cur[(i / (Kp - K)) * ld + K + i % (Kp - K)] = 0.f;   // pad columns
        for (int i = tid; i < MLP_ROWS * K; i += MLP_NT) {
            const int r = i / K, k = i - r * K;
            const int gr = min(row0 + r, g.M - 1);
            const float v = g.in[z][(size_t)gr * K + k];
            cur[r * ld + k] = v;
            // storage.add of the observations (rsl_rl RolloutStorage): the actor workgroup stores its rows, the critic's its own
            // when the critic has privileged observations
            if (g.sample && g.t >= 0 && row0 + r < g.M) {
                if (z == 0) P.st_obs[((size_t)g.t * g.M + gr) * K + k] = v;
                else if (P.st_critic_obs != P.st_obs) P.st_critic_obs[((size_t)g.t * g.M + gr) * K + k] = v;
            }
        }
    }
    __syncthreads();
    MSTAMP();
    for (int l = 0; l < g.nl; ++l) {
        const int ntiles = (g.dims[z][l + 1] + 31) / 32;
        // column tiles of this wave: wave, wave + 4, ... (workgroup-uniform count per wave up to rounding)
        const int mine = ntiles > wave ? (ntiles - wave + MLP_NW - 1) / MLP_NW : 0;
        if (g.act == 1) mlp_layer_tiles<1>(g, z, l, cur, nxt, row0, wave, mine, li, lk);
        else mlp_layer_tiles<-1>(g, z, l, cur, nxt, row0, wave, mine, li, lk);
        __syncthreads();
        MSTAMP();
        float *t = cur; cur = nxt; nxt = t;
    }
    if (!g.sample) return;
    // ---- PPO.act epilogue (the arithmetic of k_act_sample, term by term): cur = head outputs [32][N + 4]
    const int A = P.A, t = g.t, N = g.M, ldh = g.dims[z][g.nl] + 4;
    if (z == 0) {
        const float *std = P.params + P.off_std;
        if (blockIdx.x == 0 && tid < A && t == 0) P.st_sigma[tid] = std[tid];
        float *lpt = nxt;                                             // log-prob terms [32][A]
        for (int e = tid; e < MLP_ROWS * A; e += MLP_NT) {
            const int r = e / A, a = e - r * A, i = row0 + r;
            if (i >= N) continue;
            const float m = cur[r * ldh + a], s = std[a];
            const float zn = g.inject ? P.noise[(size_t)i * A + a] : philox_normal(P.seed, (uint32_t)(P.env_offset + i), (uint64_t)g.act_count, a);
            const float act = m + s * zn;
            lpt[r * A + a] = -((act - m) * (act - m)) / (2.0f * s * s) - logf(s) - 0.9189385332046727f;
            P.act_actions[(size_t)i * A + a] = act;
            P.act_mu[(size_t)i * A + a] = m;
            if (t >= 0) {
                P.st_actions[((size_t)t * N + i) * A + a] = act;
                P.st_mu[((size_t)t * N + i) * A + a] = m;
            }
        }
        __syncthreads();
        if (tid < MLP_ROWS && row0 + tid < N) {
            const int i = row0 + tid;
            float lp = 0.f;
            for (int a = 0; a < A; ++a) lp += lpt[tid * A + a];
            P.act_log_prob[i] = lp;
            if (t >= 0) P.st_log_prob[(size_t)t * N + i] = lp;
        }
    } else if (tid < MLP_ROWS && row0 + tid < N) {
        const int i = row0 + tid;
        const float v = cur[tid * ldh];
        P.act_values[i] = v;
        if (t >= 0) P.st_values[(size_t)t * N + i] = v;
    }
    MSTAMP();
}

// 0 when the fused kernel covers this network shape, -1 otherwise (caller runs the per-layer GEMMs)
extern "C" int ppok_mlp_supported(const MlpArgs *g) {
    for (int z = 0; z < 2; ++z)
        for (int l = 0; l < g->nl; ++l) {
            const int K = g->dims[z][l], N = g->dims[z][l + 1];
            if ((l > 0 && K % 16) || mlp_kpad(K) > MLP_KMAX || K < 1) return -1;
            if (l < g->nl - 1 && (N % 32 || N > MLP_KMAX)) return -1;
        }
    return 0;
}
// bf16 elements of the fragment-order image of layer (K inputs, N outputs)
extern "C" int64_t ppok_mlp_frag_elems(int K, int N) { return (int64_t)((N + 31) / 32) * 32 * mlp_kpad(K) * 3; }
extern "C" void ppok_mlp_frag_build(const MlpArgs *g, hipStream_t s) {
    hipLaunchKernelGGL(k_mlp_frag_build, dim3(256), dim3(256), 0, s, *g, const_cast<uint16_t *>(g->wfrag));
}
// P: the learner's device struct (sampling epilogue when g->sample; otherwise only passed through)
extern "C" int ppok_mlp_fwd(const MlpArgs *g, const PpoDev *P, int mask, hipStream_t s) {
    if (mask != 3 || ppok_mlp_supported(g) || ((uintptr_t)g->wfrag & 15)) return -1;
    if (g->sample && (g->M != P->N || g->dims[0][g->nl] != P->A || g->dims[1][g->nl] != 1 || P->A > LG_PPO_MAX_A)) return -1;
    static_assert(MLP_NT == 256, "finalize_body and the process lanes are written for 256 threads");
    if (g->pp && (P->N + MLP_NT - 1) / MLP_NT + 1 > (g->M + MLP_ROWS - 1) / MLP_ROWS) return -1;   // epilogue blocks must fit grid.x
    dim3 grid((g->M + MLP_ROWS - 1) / MLP_ROWS, g->pp ? 3 : 2);
    hipLaunchKernelGGL(k_mlp_fwd, grid, dim3(MLP_NT), 0, s, *g, *P);
    return 0;
}
