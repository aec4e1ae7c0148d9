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
#define MLP_KMAX 512
// B fragments of one k-step: planes h, m, l of W[n][k0 + 8 h .. + 8)
struct BFrag { bf16x8 p[3]; };
__device__ __forceinline__ BFrag mlp_load_b(const uint16_t *__restrict__ w, int64_t pl_stride, int K, int n, int k) {
    BFrag f;
    const uint16_t *q = w + (size_t)n * K + k;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) f.p[pl] = *reinterpret_cast<const bf16x8 *>(q + pl * pl_stride);
    return f;
}

// One layer for one wave: CT column tiles (n = 32 * (wave + 4 (t0 + i))) advanced together along k, so the A fragment is
// read and split once per k-step.  k is walked in blocks of S k-steps with the k index PERMUTED inside a block: lane half
// h owns the contiguous run k = 16 S kb + 8 S h .. + 8 S, its chunk s feeding MFMA k-step s (A uses the same map, the
// product does not care).  A lane's S loads of a block are then S consecutive 16-byte pieces of one weight row -- with
// S = 4 a whole 64-byte line, fetched once -- where the natural map touches every line in four widely spaced loads and
// thrashes the L1.  The next block's fragments are in flight while the current one is multiplied.
template <int CT, int S, int ACT>
__device__ __forceinline__ void mlp_layer(const MlpArgs &g, int z, int l, const float *__restrict__ cur, float *__restrict__ nxt, int row0,
                                          int wave, int t0, int li, int lk) {
    const int K = g.dims[z][l], N = g.dims[z][l + 1], ld = K + 4, ldo = N + 4, nkb = K / (16 * S);
    const bool head = l == g.nl - 1;
    const uint16_t *__restrict__ w = g.wpl + g.pl_off[z][l];
    const float *__restrict__ arow = cur + li * ld + 8 * S * lk;
    int n[CT], nc[CT];
    f32x16 acc[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        n[c] = (wave + 4 * (t0 + c)) * 32 + li;
        nc[c] = min(n[c], N - 1);                     // clamped: columns past N are computed on row N-1, never stored
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    }
    BFrag bf[2][S][CT];
    auto load_block = [&](BFrag (&dst)[S][CT], int kb) {
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int q = 0; q < S; ++q) dst[q][c] = mlp_load_b(w, g.pl_stride, K, nc[c], 16 * S * kb + 8 * S * lk + 8 * q);
    };
    auto mul_block = [&](const BFrag (&src)[S][CT], int kb) {
#pragma unroll
        for (int q = 0; q < S; ++q) {
            float x[8];
            *reinterpret_cast<float4 *>(x) = *reinterpret_cast<const float4 *>(arow + 16 * S * kb + 8 * q);
            *reinterpret_cast<float4 *>(x + 4) = *reinterpret_cast<const float4 *>(arow + 16 * S * kb + 8 * q + 4);
            bf16x8 ah, am, al;
            mlp_split8(x, ah, am, al);
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                const BFrag &b = src[q][c];
                f32x16 a = acc[c];
                a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, b.p[1], a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b.p[2], a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, b.p[0], a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b.p[1], a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, b.p[0], a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b.p[0], a, 0, 0, 0);
                acc[c] = a;
            }
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
        const float bv = g.params[g.b_off[z][l] + nc[c]];
        if (!head) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = acc[c][r] + bv;          // ACT 1: ELU compiled in (a per-element switch bloats the kernel past the I-cache)
                nxt[((r & 3) + 8 * (r >> 2) + 4 * lk) * ldo + n[c]] = ACT == 1 ? (v > 0.f ? v : __expf(v) - 1.0f) : mlp_act(g.act, v);
            }
        } else if (n[c] < N) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gr = row0 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (gr < g.M) g.out[z][(size_t)gr * N + n[c]] = acc[c][r] + bv;
            }
        }
    }
}
// S = 4 with at most two column tiles at a time (register budget of the two fragment sets), S = 1 for short or odd K
template <int S, int ACT>
__device__ __forceinline__ void mlp_layer_tiles(const MlpArgs &g, int z, int l, const float *cur, float *nxt, int row0, int wave, int mine,
                                                int li, int lk) {
    constexpr int CMAX = S == 4 ? 2 : 4;
    for (int t0 = 0; t0 < mine;) {
        const int c = min(CMAX, mine - t0);
        if (c == 4) { if constexpr (CMAX >= 4) mlp_layer<4, S, ACT>(g, z, l, cur, nxt, row0, wave, t0, li, lk); t0 += 4; }
        else if (c >= 2) { mlp_layer<2, S, ACT>(g, z, l, cur, nxt, row0, wave, t0, li, lk); t0 += 2; }
        else { mlp_layer<1, S, ACT>(g, z, l, cur, nxt, row0, wave, t0, li, lk); t0 += 1; }
    }
}

__global__ void __launch_bounds__(256, 1) k_mlp_fwd(MlpArgs g) {
    const int z = blockIdx.y;
    const int row0 = blockIdx.x * MLP_ROWS;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lk = lane >> 5;
    __shared__ __attribute__((aligned(16))) float s_a[MLP_ROWS * (MLP_KMAX + 4)];
    __shared__ __attribute__((aligned(16))) float s_b[MLP_ROWS * (MLP_KMAX + 4)];
    float *cur = s_a, *nxt = s_b;
    {   // observations of the 32 rows -> LDS (rows past M read the last row: computed, never stored)
        const int K = g.dims[z][0], ld = K + 4;
        for (int i = tid; i < MLP_ROWS * K; i += 256) {
            const int r = i / K, k = i - r * K;
            const int gr = min(row0 + r, g.M - 1);
            cur[r * ld + k] = g.in[z][(size_t)gr * K + k];
        }
    }
    __syncthreads();
    for (int l = 0; l < g.nl; ++l) {
        const int ntiles = (g.dims[z][l + 1] + 31) / 32;
        // column tiles of this wave: wave, wave + 4, ... (workgroup-uniform count per wave up to rounding)
        const int mine = ntiles > wave ? (ntiles - wave + 3) / 4 : 0;
        if (g.act == 1) {
            if (g.dims[z][l] % 64 == 0) mlp_layer_tiles<4, 1>(g, z, l, cur, nxt, row0, wave, mine, li, lk);
            else mlp_layer_tiles<1, 1>(g, z, l, cur, nxt, row0, wave, mine, li, lk);
        } else {
            if (g.dims[z][l] % 64 == 0) mlp_layer_tiles<4, -1>(g, z, l, cur, nxt, row0, wave, mine, li, lk);
            else mlp_layer_tiles<1, -1>(g, z, l, cur, nxt, row0, wave, mine, li, lk);
        }
        __syncthreads();
        float *t = cur; cur = nxt; nxt = t;
    }
}

// 0 when the fused kernel covers this network shape, -1 otherwise (caller runs the per-layer GEMMs)
extern "C" int ppok_mlp_fwd(const MlpArgs *g, int mask, hipStream_t s) {
    for (int z = 0; z < 2; ++z) {
        if (!(mask & (1 << z))) continue;
        for (int l = 0; l < g->nl; ++l) {
            const int K = g->dims[z][l], N = g->dims[z][l + 1];
            if (K % 16 || K > MLP_KMAX || K < 16 || (g->pl_off[z][l] & 7)) return -1;
            if (l < g->nl - 1 && (N % 32 || N > MLP_KMAX)) return -1;
        }
    }
    if (mask != 3 || (g->pl_stride & 7) || ((uintptr_t)g->wpl & 15)) return -1;
    dim3 grid((g->M + MLP_ROWS - 1) / MLP_ROWS, 2);
    hipLaunchKernelGGL(k_mlp_fwd, grid, dim3(256), 0, s, *g);
    return 0;
}
