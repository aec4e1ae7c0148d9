// PPO kernels for gfx950: fp32-MFMA GEMMs with fused epilogues (ActorCritic forward/backward),
// action sampling, GAE scan, PPO loss, grad-norm clip + Adam with the KL-adaptive learning rate
// kept on the device.  Semantics: rsl_rl v1.0.2 (SURVEY.md Appendix B -- third-party, not in the
// reference tree; call sites legged_gym/utils/task_registry.py:148-155).
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "ppo_device.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Hidden-layer activations of rsl_rl's get_activation (ActorCritic cfg `activation`, legged_robot_config.py:244):
// code 0 none, 1 elu, 2 selu, 3 relu, 4 lrelu (slope 0.01), 5 tanh, 6 sigmoid.  act_bwd is the derivative expressed
// through the activation's OUTPUT a (the only thing the forward pass keeps).
#define SELU_L 1.0507009873554804934193349852946f
#define SELU_LA (1.0507009873554804934193349852946f * 1.6732632423543772848170429916717f)
__device__ __forceinline__ float act_fwd(int code, float v) {
    switch (code) {
    case 1: return v > 0.f ? v : __expf(v) - 1.0f;
    case 2: return v > 0.f ? SELU_L * v : SELU_LA * (__expf(v) - 1.0f);
    case 3: return fmaxf(v, 0.f);
    case 4: return v > 0.f ? v : 0.01f * v;
    case 5: return 2.0f * __frcp_rn(1.0f + __expf(-2.0f * v)) - 1.0f;
    case 6: return __frcp_rn(1.0f + __expf(-v));
    default: return v;
    }
}
__device__ __forceinline__ float act_bwd(int code, float a) {
    switch (code) {
    case 1: return a > 0.f ? 1.0f : a + 1.0f;
    case 2: return a > 0.f ? SELU_L : a + SELU_LA;
    case 3: return a > 0.f ? 1.0f : 0.f;
    case 4: return a > 0.f ? 1.0f : 0.01f;
    case 5: return 1.0f - a * a;
    case 6: return a * (1.0f - a);
    default: return 1.0f;
    }
}

// XCD-aware workgroup -> tile map.  The 8 XCDs of an MI355X have private L2s and workgroups are dealt to them round-robin by
// linear workgroup id, so neighbouring ids -- the column tiles of one row block, the output tiles of one reduction slice --
// land on 8 different L2s and each fetches its own copy of the operand rows they share.  remap(l) regroups the ids so that
// the `inner` workgroups that share operand rows get the same XCD: XCD c owns outer indices c, c + 8, ...
static __device__ int g_xcd_remap = 1;
__device__ __forceinline__ void xcd_tile(int lin, int inner, int outer, int &o, int &i) {
    if (g_xcd_remap && (outer & 7) == 0) {
        const int c = lin & 7, j = lin >> 3;
        o = c + 8 * (j / inner);
        i = j % inner;
    } else {
        o = lin / inner;
        i = lin % inner;
    }
}

// ------------------------------------------------------------------------------------------------
// C[M,N] = opA(A) . opB(B), K = reduction length.  A_RC: A stored [m][k] (reduction contiguous),
// else [k][m].  B_RC: B stored [n][k], else [k][n].  4 waves as 2x2, each TM x TN tiles of 32x32
// computed with v_mfma_f32_32x32x2_f32 (exact fp32, A/B one VGPR per lane: lane l holds
// A[i = l&31][k = l>>5], B[k = l>>5][j = l&31]).
// LDS tiles keep the global layout (so the global->LDS copy is a straight 16-byte move) with a
// one-float row pad; fragments are ds_read_b32 with lanes on consecutive rows/columns, which is
// bank-conflict free in both layouts and far below the LDS rate at 64 cycles per fp32 MFMA.
// EPI 0: C = act(acc + bias[n]), act = g.elu code (forward, nn.Linear + activation; 0 on the head layer)
// EPI 1: C = acc * act'(aux[m][n]); colsum[n] += sum_m C   (input gradient + bias gradient below)
// EPI 2: C += acc via float atomics, reduction split over blockIdx.y   (weight gradient)
#define BK 32

template <bool RC, int ROWS, bool VEC, bool KSEQ = false, int NT = 256, bool FULL = false>
__device__ __forceinline__ void stage_load(const float *__restrict__ src, int ld, int row0, int red0, int nrows, int nred,
                                           float4 (&regs)[ROWS * BK / 4 / NT], unsigned &mask) {
    constexpr int NV = ROWS * BK / 4 / NT;
    const int tid = threadIdx.x & (NT - 1);      // index inside the group of NT threads that stages this tile (= threadIdx.x unless
                                                 // the workgroup holds two such groups: k_gemm_pp)
    const int row_lim = RC ? nrows : nred, col_lim = RC ? nred : nrows;
    mask = 0u;                          // bit v: regs[v] is in range (applied at stage_store, so that nothing
                                        // consumes the loaded data -- and waits on it -- before the MFMA block)
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int idx = tid + v * NT;
        int r, c;                       // r: index along the tile's non-contiguous dim, c: float4 along the contiguous one
        if (RC) { c = idx & (BK / 4 - 1); r = idx / (BK / 4); }
        else if (KSEQ) { c = tid & (ROWS / 4 - 1); r = NV * (tid / (ROWS / 4)) + v; }   // NV consecutive k per thread
        else { c = idx & (ROWS / 4 - 1); r = idx / (ROWS / 4); }
        const int grow = RC ? row0 + r : red0 + r;        // global row
        const int gcol = RC ? red0 + 4 * c : row0 + 4 * c;
        if (FULL) {                     // tile entirely in range (workgroup-uniform): no clamps, no mask
            regs[v] = *reinterpret_cast<const float4 *>(src + (size_t)grow * ld + gcol);
            mask = ~0u;
        } else if (VEC) {
            // branch-free: every lane loads 16 B from a clamped (always valid) address and zeroes it by
            // select when out of range -- a guarded load makes hipcc wait vmcnt(0) per element.
            // VEC implies col_lim % 4 == 0, so a float4 is entirely inside or entirely outside.
            const int rc_ = min(grow, row_lim - 1), cc_ = min(gcol, col_lim - 4);
            regs[v] = *reinterpret_cast<const float4 *>(src + (size_t)rc_ * ld + cc_);
            mask |= (grow < row_lim && gcol < col_lim) ? (1u << v) : 0u;
        } else {
            const int rc_ = min(grow, row_lim - 1);
            const float *p = src + (size_t)rc_ * ld;
            const bool rin = grow < row_lim;
            float e[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int cc_ = min(gcol + q, col_lim - 1);
                const float x = p[cc_];
                e[q] = (rin && gcol + q < col_lim) ? x : 0.f;
            }
            regs[v] = make_float4(e[0], e[1], e[2], e[3]);
            mask |= 1u << v;
        }
    }
}
template <bool RC, int ROWS>
__device__ __forceinline__ void stage_store(float *__restrict__ lds, const float4 (&regs_in)[ROWS * BK / 4 / 256], unsigned mask) {
    constexpr int NV = ROWS * BK / 4 / 256;
    const int tid = threadIdx.x;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        int idx = tid + v * 256;
        const bool in = (mask >> v) & 1u;
        float4 regs[1];
        regs[0] = make_float4(in ? regs_in[v].x : 0.f, in ? regs_in[v].y : 0.f, in ? regs_in[v].z : 0.f, in ? regs_in[v].w : 0.f);
        if (RC) {                       // lds[r][BK+1]
            int c = idx & (BK / 4 - 1), r = idx / (BK / 4);
            float *d = lds + r * (BK + 1) + 4 * c;
            d[0] = regs[0].x; d[1] = regs[0].y; d[2] = regs[0].z; d[3] = regs[0].w;
        } else {                        // lds[k][ROWS+4]
            int c = idx & (ROWS / 4 - 1), r = idx / (ROWS / 4);
            *reinterpret_cast<float4 *>(lds + r * (ROWS + 4) + 4 * c) = regs[0];
        }
    }
}
template <bool RC, int ROWS>
__device__ __forceinline__ float frag(const float *__restrict__ lds, int row, int k) {
    return RC ? lds[row * (BK + 1) + k] : lds[k * (ROWS + 4) + row];
}
template <bool RC, int ROWS>
constexpr int tile_floats() { return RC ? ROWS * (BK + 1) : BK * (ROWS + 4); }

template <bool A_RC, bool B_RC, int TM, int TN, bool VEC, bool DBUF>
__device__ __forceinline__ void gemm_mainloop(const float *__restrict__ A, const float *__restrict__ B, int lda, int ldb, int m0, int n0,
                                              int M, int N, int k_begin, int k_end, float *__restrict__ lds, int wm, int wn, int li, int lk,
                                              f32x16 (&acc)[TM][TN]) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    constexpr int AF = tile_floats<A_RC, BM>(), BF = tile_floats<B_RC, BN>();
    float4 ra[BM * BK / 4 / 256], rb[BN * BK / 4 / 256];
    unsigned ma, mb_;
    stage_load<A_RC, BM, VEC>(A, lda, m0, k_begin, M, k_end, ra, ma);
    stage_load<B_RC, BN, VEC>(B, ldb, n0, k_begin, N, k_end, rb, mb_);
    stage_store<A_RC, BM>(lds, ra, ma);
    stage_store<B_RC, BN>(lds + AF, rb, mb_);
    __syncthreads();
    int cur = 0;
    for (int k0 = k_begin; k0 < k_end; k0 += BK) {
        const bool more = k0 + BK < k_end;
        if (more) {
            stage_load<A_RC, BM, VEC>(A, lda, m0, k0 + BK, M, k_end, ra, ma);
            stage_load<B_RC, BN, VEC>(B, ldb, n0, k0 + BK, N, k_end, rb, mb_);
        }
        const float *as = lds + (DBUF ? cur : 0) * (AF + BF), *bs = as + AF;
        // fragments are fetched one group (GS k-steps) ahead of the MFMAs that consume them, so the
        // LDS latency is paid once per k-tile instead of once per k-step
        constexpr int GS = 4, NG = BK / 2 / GS;
        float av[2][GS][TM], bv[2][GS][TN];
#pragma unroll
        for (int s = 0; s < GS; ++s) {
#pragma unroll
            for (int a = 0; a < TM; ++a) av[0][s][a] = frag<A_RC, BM>(as, wm + 32 * a + li, 2 * s + lk);
#pragma unroll
            for (int b = 0; b < TN; ++b) bv[0][s][b] = frag<B_RC, BN>(bs, wn + 32 * b + li, 2 * s + lk);
        }
#pragma unroll
        for (int gq = 0; gq < NG; ++gq) {
            if (gq + 1 < NG) {
#pragma unroll
                for (int s = 0; s < GS; ++s) {
                    const int ks = 2 * ((gq + 1) * GS + s) + lk;
#pragma unroll
                    for (int a = 0; a < TM; ++a) av[(gq + 1) & 1][s][a] = frag<A_RC, BM>(as, wm + 32 * a + li, ks);
#pragma unroll
                    for (int b = 0; b < TN; ++b) bv[(gq + 1) & 1][s][b] = frag<B_RC, BN>(bs, wn + 32 * b + li, ks);
                }
            }
#pragma unroll
            for (int s = 0; s < GS; ++s)
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[gq & 1][s][a], bv[gq & 1][s][b], acc[a][b], 0, 0, 0);
        }
        if (!DBUF) __syncthreads();              // single buffer: every wave is done reading before it is refilled
        if (more) {
            stage_store<A_RC, BM>(lds + (DBUF ? (cur ^ 1) : 0) * (AF + BF), ra, ma);
            stage_store<B_RC, BN>(lds + (DBUF ? (cur ^ 1) : 0) * (AF + BF) + AF, rb, mb_);
        }
        __syncthreads();
        cur ^= 1;
    }

}


// ------------------------------------------------------------------------------------------------
// Split-bf16 ("bf16x6") mainloop: the same fp32 GEMM computed on the bf16 matrix cores, which on
// gfx950 run 16x the fp32-input MFMA rate.  Each fp32 operand is split exactly into three bf16
// terms x = h + m + l (round-to-nearest at each level: |m| <= 2^-8 |x|, |l| <= 2^-16 |x|, and the
// 24-bit significand is covered, so the split itself loses nothing).  The product a.b is summed
// from the six term products of weight >= 2^-16 (hh, hm, mh, hl, lh, mm), each EXACT in the fp32
// accumulator of v_mfma_f32_32x32x16_bf16; the three dropped ones (ml, lm, ll) are <= 2^-23 |ab|
// worst case and unbiased -- below one fp32 rounding of the product.  6 bf16 MFMAs of 32 cycles
// replace 8 fp32 MFMAs of 64 per 32x32x16 block: 2.67x the MFMA-bound rate at fp32 accuracy
// (tests/test_hip_ppo.py checks both paths against float64).
//
// LDS image per operand: 3 planes [physical row][32 bf16 + 16 B pad] (80-B rows: a 5-slot stride
// keeps the 16-lane groups of ds_read_b128 on distinct 16-B slots).  Logical row r lives at
// physical row (r&3)*(ROWS/4+4) + (r>>2): the operand whose reduction dimension is NOT contiguous
// in memory arrives as float4s along the rows, and this places the four rows of one float4 in
// four different bank phases while fragment reads (32 consecutive rows) stay conflict-free.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define X6_ROWB 80
#ifndef LG_GEMM_STEADY              // compile-time A/B: 1 = steady-state loop with unconditional prefetch (forward / input gradient)
#define LG_GEMM_STEADY 0
#endif
#ifndef LG_DW_STEADY                // the same for the weight-gradient kernel
#define LG_DW_STEADY 0
#endif
// Workgroup barrier of the GEMM mainloops: LDS traffic only.  __syncthreads() carries workgroup-scope fences, for which hipcc drains
// EVERY outstanding memory operation (s_waitcnt vmcnt(0)) -- including the global loads issued two k-tiles ahead, whose latency
// the prefetch distance exists to hide.  The mainloops exchange data through LDS alone: waiting for this wave's LDS operations
// and the barrier is all the ordering they need; the loaded registers are waited for where they are used (counted vmcnt).
// LG_GEMM_FULL_BARRIER (compile-time, A/B) restores __syncthreads().
__device__ __forceinline__ void lds_barrier() {
#ifdef LG_GEMM_FULL_BARRIER
    __syncthreads();
#else
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}

__device__ __forceinline__ uint32_t cvt_pk_bf16(float a, float b) {
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// (x0, x1) -> packed bf16 pairs of the three split terms
__device__ __forceinline__ void split2(float x0, float x1, uint32_t &h, uint32_t &m, uint32_t &l) {
#ifdef LG_EXP_NOSPLIT                                              // timing experiment only (make exp): what the split arithmetic costs
    h = __float_as_uint(x0); m = __float_as_uint(x1); l = h ^ m;
    return;
#endif
    typedef float f32x2 __attribute__((ext_vector_type(2)));      // v_pk_add_f32: both remainders in one instruction
    h = cvt_pk_bf16(x0, x1);
    f32x2 r = f32x2{x0, x1} - f32x2{__uint_as_float(h << 16), __uint_as_float(h & 0xffff0000u)};
    m = cvt_pk_bf16(r.x, r.y);
    r -= f32x2{__uint_as_float(m << 16), __uint_as_float(m & 0xffff0000u)};
    l = cvt_pk_bf16(r.x, r.y);
}
template <int ROWS>
__device__ __forceinline__ int x6_prow(int r) { return (r & 3) * (ROWS / 4 + 4) + (r >> 2); }
template <int ROWS>
constexpr int x6_plane_bytes() { return (ROWS + 16) * X6_ROWB; }

template <bool RC, int ROWS, int NT, bool FULL = false>
__device__ __forceinline__ void stage_store_x6(unsigned char *__restrict__ lds, const float4 (&regs)[ROWS * BK / 4 / NT], unsigned mask) {
    constexpr int NV = ROWS * BK / 4 / NT;
    constexpr int PL = x6_plane_bytes<ROWS>();
    const int tid = threadIdx.x & (NT - 1);
    if constexpr (RC) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int idx = tid + v * NT;
            const int c = idx & (BK / 4 - 1), r = idx / (BK / 4);
            const bool in = FULL || ((mask >> v) & 1u);
            const float x0 = in ? regs[v].x : 0.f, x1 = in ? regs[v].y : 0.f, x2 = in ? regs[v].z : 0.f, x3 = in ? regs[v].w : 0.f;
            uint32_t h0, m0, l0, h1, m1, l1;
            split2(x0, x1, h0, m0, l0);
            split2(x2, x3, h1, m1, l1);
            unsigned char *d = lds + x6_prow<ROWS>(r) * X6_ROWB + 8 * c;
            *reinterpret_cast<uint2 *>(d) = make_uint2(h0, h1);
            *reinterpret_cast<uint2 *>(d + PL) = make_uint2(m0, m1);
            *reinterpret_cast<uint2 *>(d + 2 * PL) = make_uint2(l0, l1);
        }
    } else {
        // regs[v] = rows 4c..4c+3 at k = NV*kg + v (stage_load KSEQ mapping)
        const int c = tid & (ROWS / 4 - 1), kg = tid / (ROWS / 4);
        float e[4][NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const bool in = FULL || ((mask >> v) & 1u);
            e[0][v] = in ? regs[v].x : 0.f; e[1][v] = in ? regs[v].y : 0.f;
            e[2][v] = in ? regs[v].z : 0.f; e[3][v] = in ? regs[v].w : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            unsigned char *d = lds + (q * (ROWS / 4 + 4) + c) * X6_ROWB + 2 * NV * kg;
            if (NV == 4) {
                uint32_t h0, m0, l0, h1, m1, l1;
                split2(e[q][0], e[q][1], h0, m0, l0);
                split2(e[q][2 % NV], e[q][3 % NV], h1, m1, l1);
                *reinterpret_cast<uint2 *>(d) = make_uint2(h0, h1);
                *reinterpret_cast<uint2 *>(d + PL) = make_uint2(m0, m1);
                *reinterpret_cast<uint2 *>(d + 2 * PL) = make_uint2(l0, l1);
            } else {
                static_assert(NV == 4 || NV == 2, "tile/threads combination");
                uint32_t h0, m0, l0;
                split2(e[q][0], e[q][1 % NV], h0, m0, l0);
                *reinterpret_cast<uint32_t *>(d) = h0;
                *reinterpret_cast<uint32_t *>(d + PL) = m0;
                *reinterpret_cast<uint32_t *>(d + 2 * PL) = l0;
            }
        }
    }
}

// B operand already split (weight planes kept by the optimiser step): the tile is three [ROWS][32] bf16 images,
// staged as plain 16-byte copies -- no conversion work in the loop.  Chunk c of a tile: plane c / (ROWS*4),
// row (c / 4) % ROWS, 16-byte quarter c % 4 of the row's 64 bytes.
template <int ROWS, int NT>
__device__ __forceinline__ void stage_load_pl(const uint16_t *__restrict__ src, int64_t pl_stride, int ld, int row0, int red0, int nrows,
                                              int nred, uint4 (&regs)[ROWS * 12 / NT], unsigned &mask) {
    constexpr int NV = ROWS * 12 / NT;
    mask = 0u;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int c = (threadIdx.x & (NT - 1)) + v * NT;
        const int pl = c / (ROWS * 4), r = (c >> 2) % ROWS, q = c & 3;
        const int grow = row0 + r, gk = red0 + 8 * q;
        const int rc_ = min(grow, nrows - 1), kc_ = min(gk, nred - 8);      // nred % 8 == 0 (checked by the launcher)
        regs[v] = *reinterpret_cast<const uint4 *>(src + pl * pl_stride + (size_t)rc_ * ld + kc_);
        mask |= (grow < nrows && gk < nred) ? (1u << v) : 0u;
    }
}
template <int ROWS, int NT>
__device__ __forceinline__ void stage_store_pl(unsigned char *__restrict__ lds, const uint4 (&regs)[ROWS * 12 / NT], unsigned mask) {
    constexpr int NV = ROWS * 12 / NT;
    constexpr int PL = x6_plane_bytes<ROWS>();
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int c = (threadIdx.x & (NT - 1)) + v * NT;
        const int pl = c / (ROWS * 4), r = (c >> 2) % ROWS, q = c & 3;
        const bool in = (mask >> v) & 1u;
        const uint4 x = in ? regs[v] : make_uint4(0u, 0u, 0u, 0u);
        *reinterpret_cast<uint4 *>(lds + pl * PL + x6_prow<ROWS>(r) * X6_ROWB + 16 * q) = x;
    }
}

// B operand from the SAME weight planes when the reduction runs over the planes' ROWS (input gradient: dz . W with W [k][n],
// n contiguous): the k-tile is staged as it lies in memory -- [32 k-rows][BN n-columns] bf16 per plane, plain 16-byte copies --
// and the fragments (8 consecutive k for one n per lane) come out of gfx950's transposing LDS read, two
// ds_read_b64_tr_b16 (4 k x 16 n each per 16-lane group) per operand.  Image: 8-row x 32-column subtiles of 512 B with the
// chunk XOR of cdna_hip_programming.md T10 image (a): off(row, ch) = GRP (row>>3) + 512 (ch>>2) + 64 (row&7) +
// 16 ((ch&3) ^ ((row>>2)&3)), ch = 16-byte chunk of the row, GRP = 512 BN/32 -- conflict-free for both the b128 stores'
// rows and the transposed reads.
// Subtile stride PLT_SUB = 512 + 64 bytes (LG_PLT_SUB): with subtiles exactly 512 B apart the 4 (stores of 16 B: 16 lanes, stores of 8 B: 32
// lanes) subtiles one k-row spans fall on the SAME 16 banks -- a 4-way conflict on every staging store (SQ_LDS_BANK_CONFLICT = 17 % of the LDS
// cycles of the input-gradient kernel, 29-33 % of the weight-gradient kernels': profiles/r03_kernel_clocks.txt).  The 64-byte pad rotates
// them onto the four quarters of the bank row; a transposed read stays inside one subtile and is unaffected.
#ifndef LG_PLT_SUB
#define LG_PLT_SUB 576
#endif
typedef short s16x4 __attribute__((ext_vector_type(4)));
template <int BN>
__device__ __forceinline__ int plt_off(int row, int ch) {
    return (BN / 32 * LG_PLT_SUB) * (row >> 3) + LG_PLT_SUB * (ch >> 2) + 64 * (row & 7) + 16 * ((ch & 3) ^ ((row >> 2) & 3));
}
template <int BN>
constexpr int plt_plane_bytes() { return BK / 8 * (BN / 32) * LG_PLT_SUB; }

template <int BN, int NT>
__device__ __forceinline__ void stage_load_plt(const uint16_t *__restrict__ src, int64_t pl_stride, int ld, int n0, int red0, int ncols,
                                               int nred, uint4 (&regs)[BK * BN / 8 * 3 / NT], unsigned &mask) {
    constexpr int CPR = BN / 8, NV = BK * CPR * 3 / NT;         // chunks per row; chunks per thread
    mask = 0u;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int c = (threadIdx.x & (NT - 1)) + v * NT;
        const int pl = c / (BK * CPR), k = (c / CPR) % BK, ch = c % CPR;
        const int gk = red0 + k, gn = n0 + 8 * ch;
        const int kc_ = min(gk, nred - 1), nc_ = min(gn, ncols - 8);      // ncols % 8 == 0 (checked by the launcher)
        regs[v] = *reinterpret_cast<const uint4 *>(src + pl * pl_stride + (size_t)kc_ * ld + nc_);
        mask |= (gk < nred && gn < ncols) ? (1u << v) : 0u;
    }
}
template <int BN, int NT>
__device__ __forceinline__ void stage_store_plt(unsigned char *__restrict__ lds, const uint4 (&regs)[BK * BN / 8 * 3 / NT], unsigned mask) {
    constexpr int CPR = BN / 8, NV = BK * CPR * 3 / NT;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int c = (threadIdx.x & (NT - 1)) + v * NT;
        const int pl = c / (BK * CPR), k = (c / CPR) % BK, ch = c % CPR;
        const bool in = (mask >> v) & 1u;
        const uint4 x = in ? regs[v] : make_uint4(0u, 0u, 0u, 0u);
        *reinterpret_cast<uint4 *>(lds + pl * plt_plane_bytes<BN>() + plt_off<BN>(k, ch)) = x;
    }
}
__device__ __forceinline__ s16x4 lds_read_tr16(const unsigned char *p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3))) *)p);
}

template <bool A_RC, bool B_RC, int TM, int TN, int WGM, int WGN, bool VEC, bool FULL = false, int B_PL = 0, bool LDB = false>
__device__ __forceinline__ void gemm_mainloop_x6(const float *__restrict__ A, const float *__restrict__ B, int lda, int ldb, int m0, int n0,
                                                 int M, int N, int k_begin, int k_end, unsigned char *__restrict__ lds, int wm, int wn, int li,
                                                 int lk, f32x16 (&acc)[TM][TN], const uint16_t *__restrict__ Bpl = nullptr,
                                                 int64_t pl_stride = 0) {
    constexpr int BM = 32 * TM * WGM, BN = 32 * TN * WGN, NT = 64 * WGM * WGN;
    constexpr int APL = x6_plane_bytes<BM>(), BPL = B_PL == 2 ? plt_plane_bytes<BN>() : x6_plane_bytes<BN>();
    constexpr int NVA = BM * BK / 4 / NT, NVB = BN * BK / 4 / NT;
    // two register sets: the global loads of k-tile t+2 are issued before the MFMAs of tile t, and tile t+1 (already
    // landed) is split and stored after them -- one full iteration to cover the L2/HBM latency.
    // LDB: two LDS stages -- the split + store of tile t+1 goes to the other stage with no barrier before it and can overlap
    // the MFMAs of tile t (one barrier per k-tile instead of two).  Built for the 64x64 configuration; measured NEUTRAL there
    // (12.5 vs 13.1 us per rollout GEMM; tools/kernel_avg.sh), as was a prefetch distance of 4: those launches are bound by
    // their fixed cost (~7 us for a K = 48 or K = 128 layer) and were replaced by the one-launch forward of ppo_mlp_fused.hip.
    // Kept selectable (LG_GEMM_LDB) for shapes the fused forward does not cover; off by default.
    constexpr int PD = 2;
    constexpr int NVP = B_PL ? BN * 12 / NT : 1;                 // 16-byte chunks of a plane k-tile per thread (either plane layout)
    struct Regs { float4 a[NVA], b[NVB]; uint4 p[NVP]; unsigned ma = 0, mb = 0; };
    Regs R[PD];
    unsigned char *lds_b = lds + 3 * APL;
    auto load_tile = [&](Regs &r, int k) {
#ifdef LG_EXP_A_SAME_TILE           // timing experiment only (make exp): every k-tile re-reads the first A tile (cache hits instead of HBM)
        stage_load<A_RC, BM, VEC, true, NT, FULL>(A, lda, m0, k_begin, M, k_end, r.a, r.ma);
#else
        stage_load<A_RC, BM, VEC, true, NT, FULL>(A, lda, m0, k, M, k_end, r.a, r.ma);
#endif
#ifdef LG_EXP_B_SAME_TILE           // timing experiment only: the same for the weight-plane tile
        if constexpr (B_PL == 2) stage_load_plt<BN, NT>(Bpl, pl_stride, ldb, n0, k_begin, N, k_end, r.p, r.mb);
        else if constexpr (B_PL == 1) stage_load_pl<BN, NT>(Bpl, pl_stride, ldb, n0, k_begin, N, k_end, r.p, r.mb);
#else
        if constexpr (B_PL == 2) stage_load_plt<BN, NT>(Bpl, pl_stride, ldb, n0, k, N, k_end, r.p, r.mb);
        else if constexpr (B_PL == 1) stage_load_pl<BN, NT>(Bpl, pl_stride, ldb, n0, k, N, k_end, r.p, r.mb);
#endif
        else stage_load<B_RC, BN, VEC, true, NT, FULL>(B, ldb, n0, k, N, k_end, r.b, r.mb);
    };
    constexpr int STAGE = 3 * APL + 3 * BPL;                     // bytes of one LDS stage
    auto store_tile = [&](Regs &r, int st) {
        stage_store_x6<A_RC, BM, NT, FULL>(lds + st * STAGE, r.a, r.ma);
        if constexpr (B_PL == 2) stage_store_plt<BN, NT>(lds_b + st * STAGE, r.p, r.mb);
        else if constexpr (B_PL == 1) stage_store_pl<BN, NT>(lds_b + st * STAGE, r.p, r.mb);
        else stage_store_x6<B_RC, BN, NT, FULL>(lds_b + st * STAGE, r.b, r.mb);
    };
    load_tile(R[0], k_begin);
#pragma unroll
    for (int d = 1; d < PD; ++d)
        if (k_begin + d * BK < k_end) load_tile(R[d], k_begin + d * BK);
    store_tile(R[0], 0);
    lds_barrier();
    // fragment of tile a, plane p, k-step s: base + a*8*X6_ROWB (32 logical rows = 8 physical) + p*PL + s*32
    const unsigned char *fa = lds + x6_prow<BM>(wm + li) * X6_ROWB + 16 * lk;
    const unsigned char *fb = lds_b + x6_prow<BN>(wn + li) * X6_ROWB + 16 * lk;
    // transposed-read bases of this lane (B_PL == 2): group g = lane >> 4 takes n-columns 16 (g & 1) .. +15 and k-rows
    // 8 (g >> 1) + 4 h .. +3 of a k-step; lane 4 q + p of the group addresses row q, half-chunk p (T10)
    const int tg = (threadIdx.x >> 4) & 3, tq = (threadIdx.x >> 2) & 3, tp = threadIdx.x & 3;
    const unsigned char *ft[2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
        ft[h] = lds_b + (BN / 32 * LG_PLT_SUB) * (tg >> 1) + LG_PLT_SUB * (wn / 32) + 64 * (4 * h + tq) +
                16 * ((2 * (tg & 1) + (tp >> 1)) ^ (2 * (tg >> 1) + h)) + 8 * (tp & 1);
    auto read_b = [&](int b, int p, int s, int so) -> bf16x8 {
        if constexpr (B_PL == 2) {
            const int o = so + LG_PLT_SUB * b + p * BPL + (BN / 32 * LG_PLT_SUB) * 2 * s;
            const s16x4 lo = lds_read_tr16(ft[0] + o), hi = lds_read_tr16(ft[1] + o);
            typedef short s16x8 __attribute__((ext_vector_type(8)));
            const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            return __builtin_bit_cast(bf16x8, v);
        } else {
            return *reinterpret_cast<const bf16x8 *>(fb + so + b * 8 * X6_ROWB + p * BPL + s * 32);
        }
    };

    // one k-tile: x = the set holding tile k0 + BK (stored to LDS after the MFMAs), y = the set tile k0 + PD BK is loaded into.
    // steady = std::true_type: both exist (the caller checked), so the loads and the store are UNCONDITIONAL.  hipcc counts
    // s_waitcnt vmcnt per path and takes the minimum where paths join: with the prefetch under `if (k0 + PD BK < k_end)` the
    // store of tile k0 + BK -- whose loads are a whole iteration old -- waited as if the prefetch had not been issued, i.e. until the
    // loads issued a few hundred cycles earlier had landed (vmcnt(3)..(0) instead of (8)..(5) in the ISA): every k-tile exposed a
    // memory round trip behind its MFMA block.  The steady-state loop below has no such join; the last 2 PD - 1 tiles run the guarded form.
    auto body = [&](auto steady, int k0, Regs &x, Regs &y, int st) {
        constexpr bool ST = decltype(steady)::value;
        const int so = LDB ? st * STAGE : 0;                     // stage the MFMAs of this k-tile read
        if (ST || k0 + PD * BK < k_end) load_tile(y, k0 + PD * BK);
        if constexpr (LDB) {
            if (ST || k0 + BK < k_end) store_tile(x, st ^ 1);
        }
        bf16x8 av[2][TM][3], bv[2][TN][3];
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int p = 0; p < 3; ++p) av[0][a][p] = *reinterpret_cast<const bf16x8 *>(fa + so + a * 8 * X6_ROWB + p * APL);
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int p = 0; p < 3; ++p) bv[0][b][p] = read_b(b, p, 0, so);
#pragma unroll
        for (int s = 0; s < BK / 16; ++s) {
            if (s + 1 < BK / 16) {
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int p = 0; p < 3; ++p)
                        av[(s + 1) & 1][a][p] = *reinterpret_cast<const bf16x8 *>(fa + so + a * 8 * X6_ROWB + p * APL + (s + 1) * 32);
#pragma unroll
                for (int b = 0; b < TN; ++b)
#pragma unroll
                    for (int p = 0; p < 3; ++p)
                        bv[(s + 1) & 1][b][p] = read_b(b, p, s + 1, so);
            }
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    const bf16x8 *x = av[s & 1][a], *y = bv[s & 1][b];
                    f32x16 c = acc[a][b];                 // smallest terms first
#ifndef LG_EXP_THREE_PRODUCTS        // timing experiment only (make exp): what a three-product scheme (two-term fp16 split) would execute
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[1], y[1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[0], y[2], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[2], y[0], c, 0, 0, 0);
#endif
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[0], y[1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[1], y[0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[0], y[0], c, 0, 0, 0);
                    acc[a][b] = c;
                }
        }
        if constexpr (!LDB) {
            lds_barrier();                     // every wave is done reading before the tile is refilled
            if (ST || k0 + BK < k_end) store_tile(x, 0);
        }
        lds_barrier();
    };
    int k0 = k_begin;
    if (LG_GEMM_STEADY)
        for (; k0 + (2 * PD - 1) * BK < k_end; k0 += PD * BK) {      // every tile of the group has its successor and its prefetch target
#pragma unroll
            for (int d = 0; d < PD; ++d) body(std::true_type{}, k0 + d * BK, R[(d + 1) % PD], R[d], d & 1);
        }
    for (; k0 < k_end; k0 += PD * BK) {
#pragma unroll
        for (int d = 0; d < PD; ++d)
            if (d == 0 || k0 + d * BK < k_end) body(std::false_type{}, k0 + d * BK, R[(d + 1) % PD], R[d], d & 1);
    }
}

// Epilogue of k_gemm.  acc[a][b][r]: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).  ACT: 1 = ELU and 0 = none are
// compiled in (the reference's configurations); -1 = the activation code g.elu is switched on per element.
template <int EPI, int TM, int TN, int BM, int BN, int ACT>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs &g, int z, int M, int N, int m0, int n0, int wm, int wn, int li, int lk, int ldc,
                                              f32x16 (&acc)[TM][TN]) {
    float *__restrict__ C = g.C[z];
    const bool interior = (m0 + BM <= M) && (n0 + BN <= N);     // workgroup-uniform
    const int elu = g.elu;                                      // activation code (workgroup-uniform)
    if (interior) {
        // unguarded path: loads of the epilogue operand are issued as one batch (no per-element branch,
        // which would serialise them behind s_waitcnt vmcnt(0)), then compute + store
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int n = n0 + wn + 32 * b + li;
            float csum = 0.f;
            const float bias = (EPI == 0 && g.bias[z]) ? g.bias[z][n] : 0.f;
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                const int mb = m0 + wm + 32 * a + 4 * lk;
                float aux[16];
                if (EPI == 1) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        aux[r] = g.aux[z][(size_t)(mb + (r & 3) + 8 * (r >> 2)) * g.ldaux[z] + n];
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float *cp = &C[(size_t)(mb + (r & 3) + 8 * (r >> 2)) * ldc + n];
                    float v = acc[a][b][r];
                    if (EPI == 0) {
                        v += bias;
                        v = (ACT == 1 ? (v > 0.f ? v : __expf(v) - 1.0f) : ACT == 0 ? v : act_fwd(elu, v));
                        *cp = v;
                    } else if (EPI == 1) {
                        v *= (ACT == 1 ? (aux[r] > 0.f ? 1.0f : aux[r] + 1.0f) : ACT == 0 ? 1.0f : act_bwd(elu, aux[r]));
                        *cp = v;
                        csum += v;
                    } else {
#ifdef LG_EXP_NO_DW_ATOMICS                                           // timing experiment only (make exp): what the split-K atomics cost
                        asm volatile("" :: "v"(v), "v"(cp));
#elif defined(LG_EXP_DW_PLAIN_STORE)                                  // timing experiment only: plain stores of the same bytes
                        *cp = v;
#else
                        acc_add(g, cp, v);
#endif
                    }
                }
            }
            if (EPI == 1 && g.colsum[z]) {
                csum += __shfl_xor(csum, 32);
                if (lk == 0) acc_add(g, &g.colsum[z][n], csum);
            }
        }
        return;
    }
#pragma unroll
    for (int b = 0; b < TN; ++b) {
        const int n = n0 + wn + 32 * b + li;
        float csum = 0.f;
        const float bias = (EPI == 0 && g.bias[z] && n < N) ? g.bias[z][n] : 0.f;
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (m < M && n < N) {
                    float v = acc[a][b][r];
                    if (EPI == 0) {
                        v += bias;
                        v = (ACT == 1 ? (v > 0.f ? v : __expf(v) - 1.0f) : ACT == 0 ? v : act_fwd(elu, v));
                        C[(size_t)m * ldc + n] = v;
                    } else if (EPI == 1) {
                        const float act = g.aux[z][(size_t)m * g.ldaux[z] + n];
                        v *= (ACT == 1 ? (act > 0.f ? 1.0f : act + 1.0f) : ACT == 0 ? 1.0f : act_bwd(elu, act));
                        C[(size_t)m * ldc + n] = v;
                        csum += v;
                    } else {
                        acc_add(g, &C[(size_t)m * ldc + n], v);
                    }
                }
            }
        if (EPI == 1 && g.colsum[z]) {
            csum += __shfl_xor(csum, 32);
            if (lk == 0 && n < N) acc_add(g, &g.colsum[z][n], csum);
        }
    }
}

#ifdef LG_EXP_GEMM_CLOCK             // diagnostic build only (make exp): in-kernel clock of the GEMM workgroups, s_memtime / s_memrealtime
__device__ unsigned long long g_gemm_clk[2 * 2048];
extern "C" void ppok_debug_read_gemm_clock(unsigned long long *host) { (void)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_gemm_clk), sizeof(g_gemm_clk)); }
#endif
template <bool A_RC, bool B_RC, int EPI, int TM, int TN, bool DBUF, bool X6 = false, int WGM = 2, int WGN = 2, int B_PL = 0, bool LDB = false>
__global__ void __launch_bounds__(64 * WGM * WGN, X6 ? (WGM * WGN == 8 ? 4 : 2) : 1) k_gemm(GemmArgs g) {
#ifdef LG_EXP_GEMM_CLOCK
    const unsigned long long clk_t0 = __builtin_amdgcn_s_memtime(), clk_r0 = __builtin_amdgcn_s_memrealtime();
    struct ClkEnd { unsigned long long t0, r0; __device__ ~ClkEnd() {
        const unsigned b = blockIdx.x + gridDim.x * blockIdx.z;
        if (threadIdx.x == 0 && b < 2048) { g_gemm_clk[2 * b] = __builtin_amdgcn_s_memtime() - t0; g_gemm_clk[2 * b + 1] = __builtin_amdgcn_s_memrealtime() - r0; } } } clk_end{clk_t0, clk_r0};
#endif
    static_assert(B_PL == 0 || (X6 && (B_PL == 1) == B_RC), "weight planes feed the split-bf16 mainloop: [n][k] planes as the reduction-"
                  "contiguous operand (1), the same planes read along their rows through the transposing LDS read (2)");
    constexpr int BM = 32 * TM * WGM, BN = 32 * TN * WGN;       // WGM x WGN waves, each TM x TN tiles of 32x32
    static_assert(X6 || (WGM == 2 && WGN == 2), "the fp32-input mainloop is written for 2x2 waves");
    const int z = blockIdx.z;
    const int M = g.M[z], N = g.N[z], K = g.K[z];
    const int tiles_n = (N + BN - 1) / BN, tiles_m = (M + BM - 1) / BM;
    if ((int)blockIdx.x >= tiles_n * tiles_m) return;
    int tm, tn;
    if (EPI == 2) { tm = blockIdx.x / tiles_n; tn = blockIdx.x % tiles_n; }
    else xcd_tile((int)blockIdx.x, tiles_n, tiles_m, tm, tn);      // the column tiles of a row block share the A rows: one XCD
    const int m0 = tm * BM, n0 = tn * BN;
    // reduction range of this workgroup (split only used by EPI 2)
    int k_begin = 0, k_end = K;
    if (EPI == 2) {
        int per = ((K + gridDim.y - 1) / gridDim.y + BK - 1) / BK * BK;
        k_begin = blockIdx.y * per;
        k_end = min(K, k_begin + per);
        if (k_begin >= k_end) return;
    }
    const float *__restrict__ A = g.A[z];
    const float *__restrict__ B = g.B[z];
    const int lda = g.lda[z], ldb = g.ldb[z], ldc = g.ldc[z];
    // 16-byte path: aligned rows and a contiguous-dimension limit that no float4 straddles
    const bool a_vec = (lda & 3) == 0 && ((uintptr_t)A & 15) == 0 && (((A_RC ? k_end : M) & 3) == 0) && (A_RC ? k_end : M) >= 4;
    const bool b_vec = (ldb & 3) == 0 && ((uintptr_t)B & 15) == 0 && (((B_RC ? k_end : N) & 3) == 0) && (B_RC ? k_end : N) >= 4;

    constexpr int AF = tile_floats<A_RC, BM>(), BF = tile_floats<B_RC, BN>();
    static_assert(!LDB || B_PL, "two LDS stages: weight-plane mainloops only");
    constexpr int LDS_BYTES = X6 ? (LDB ? 2 : 1) * 3 * (x6_plane_bytes<BM>() + (B_PL == 2 ? plt_plane_bytes<BN>() : x6_plane_bytes<BN>()))
                                 : (DBUF ? 2 : 1) * (AF + BF) * 4;
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[LDS_BYTES];
    float *lds = reinterpret_cast<float *>(lds_raw);

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wm = (wave / WGN) * 32 * TM, wn = (wave % WGN) * 32 * TN;
    const int li = lane & 31, lk = lane >> 5;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    if (X6) {
        // whole tile in range in all three dimensions (workgroup-uniform): staging without clamps and masks
        const bool full = a_vec && b_vec && m0 + BM <= M && n0 + BN <= N && ((k_end - k_begin) % BK) == 0;
        if (B_PL) {
            const bool fullp = a_vec && m0 + BM <= M && n0 + BN <= N && ((k_end - k_begin) % BK) == 0;
            if (fullp) gemm_mainloop_x6<A_RC, B_RC, TM, TN, WGM, WGN, true, true, B_PL, LDB>(A, B, lda, ldb, m0, n0, M, N, k_begin, k_end, lds_raw, wm, wn, li, lk, acc, g.Bpl[z], g.pl_stride);
            else if (a_vec) gemm_mainloop_x6<A_RC, B_RC, TM, TN, WGM, WGN, true, false, B_PL, LDB>(A, B, lda, ldb, m0, n0, M, N, k_begin, k_end, lds_raw, wm, wn, li, lk, acc, g.Bpl[z], g.pl_stride);
            else gemm_mainloop_x6<A_RC, B_RC, TM, TN, WGM, WGN, false, false, B_PL, LDB>(A, B, lda, ldb, m0, n0, M, N, k_begin, k_end, lds_raw, wm, wn, li, lk, acc, g.Bpl[z], g.pl_stride);
        } else
        if (full) gemm_mainloop_x6<A_RC, B_RC, TM, TN, WGM, WGN, true, true>(A, B, lda, ldb, m0, n0, M, N, k_begin, k_end, lds_raw, wm, wn, li, lk, acc);
        else if (a_vec && b_vec) gemm_mainloop_x6<A_RC, B_RC, TM, TN, WGM, WGN, true>(A, B, lda, ldb, m0, n0, M, N, k_begin, k_end, lds_raw, wm, wn, li, lk, acc);
        else gemm_mainloop_x6<A_RC, B_RC, TM, TN, WGM, WGN, false>(A, B, lda, ldb, m0, n0, M, N, k_begin, k_end, lds_raw, wm, wn, li, lk, acc);
    } else {
        if (a_vec && b_vec) gemm_mainloop<A_RC, B_RC, TM, TN, true, DBUF>(A, B, lda, ldb, m0, n0, M, N, k_begin, k_end, lds, wm, wn, li, lk, acc);
        else gemm_mainloop<A_RC, B_RC, TM, TN, false, DBUF>(A, B, lda, ldb, m0, n0, M, N, k_begin, k_end, lds, wm, wn, li, lk, acc);
    }

    // ---- epilogue (activation specialised on the workgroup-uniform code)
    if (EPI == 2 || g.elu == 1) gemm_epilogue<EPI, TM, TN, BM, BN, 1>(g, z, M, N, m0, n0, wm, wn, li, lk, ldc, acc);
    else if (g.elu == 0) gemm_epilogue<EPI, TM, TN, BM, BN, 0>(g, z, M, N, m0, n0, wm, wn, li, lk, ldc, acc);
    else gemm_epilogue<EPI, TM, TN, BM, BN, -1>(g, z, M, N, m0, n0, wm, wn, li, lk, ldc, acc);
}

// ------------------------------------------------------------------------------------------------
// Weight gradient dW[M][N] += A^T . B over a slice of the minibatch rows, A = dz [k][m], B = act [k][n], both with the
// REDUCTION index as the row index in memory.  Both k-tiles are staged the way they lie in HBM -- [32 k-rows][BM | BN columns],
// float4 along the rows, split into the three bf16 planes on the way (8-byte LDS stores, consecutive lanes on consecutive
// half-chunks) -- and every operand fragment (8 consecutive k of one column per lane) comes out of the transposing LDS read,
// as in the input-gradient kernel.  The k_gemm<false,false,2,...> path it replaces transposed while staging: 4-byte stores
// scattered over four physical rows per float4, 4-way bank conflicts, 2.7 us per k-tile with a CU to itself.
template <int ROWS, int NT>
__device__ __forceinline__ void stage_store_x6t(unsigned char *__restrict__ lds, const float4 (&regs)[ROWS * BK / 4 / NT], unsigned mask) {
    constexpr int NV = ROWS * BK / 4 / NT, PL = plt_plane_bytes<ROWS>();
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int idx = threadIdx.x + v * NT;
        const int c = idx & (ROWS / 4 - 1), r = idx / (ROWS / 4);          // float4 c of k-row r (stage_load<false,...> mapping)
        const bool in = (mask >> v) & 1u;
        const float x0 = in ? regs[v].x : 0.f, x1 = in ? regs[v].y : 0.f, x2 = in ? regs[v].z : 0.f, x3 = in ? regs[v].w : 0.f;
        uint32_t h0, m0, l0, h1, m1, l1;
        split2(x0, x1, h0, m0, l0);
        split2(x2, x3, h1, m1, l1);
        unsigned char *d = lds + plt_off<ROWS>(r, c >> 1) + 8 * (c & 1);
        *reinterpret_cast<uint2 *>(d) = make_uint2(h0, h1);
        *reinterpret_cast<uint2 *>(d + PL) = make_uint2(m0, m1);
        *reinterpret_cast<uint2 *>(d + 2 * PL) = make_uint2(l0, l1);
    }
}

// PD: k-tiles of global loads in flight per workgroup (register sets).  3 and 4 measured no faster for the first layer's thin
// gradient (42.8 / 56.9 vs 42.4 us: the fourth set costs a workgroup per CU) and slower for the 128 x 128 tiles (spills at 128 VGPRs);
// neither were 128 x 64 tiles or 96 reduction slices for that launch (profiles/r03_ab.txt).
template <int TM, int TN, int WGM, int WGN, int PD = 2>
__global__ void __launch_bounds__(64 * WGM * WGN, WGM * WGN == 8 ? 4 : 2) k_gemm_dw_t(GemmArgs g) {
    constexpr int BM = 32 * TM * WGM, BN = 32 * TN * WGN, NT = 64 * WGM * WGN;
    constexpr int APL = plt_plane_bytes<BM>(), BPL = plt_plane_bytes<BN>();
    constexpr int NVA = BM * BK / 4 / NT, NVB = BN * BK / 4 / NT;
    const int z = blockIdx.z;
    const int M = g.M[z], N = g.N[z], K = g.K[z];
    const int tiles_n = (N + BN - 1) / BN, tiles_m = (M + BM - 1) / BM;
    if ((int)blockIdx.x >= tiles_n * tiles_m) return;
    int slice, tile;                                               // the output tiles of one reduction slice share its rows: one XCD
    xcd_tile((int)(blockIdx.x + gridDim.x * blockIdx.y), (int)gridDim.x, (int)gridDim.y, slice, tile);
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    const int per = ((K + gridDim.y - 1) / gridDim.y + BK - 1) / BK * BK;
    const int k_begin = slice * per, k_end = min(K, k_begin + per);
    if (k_begin >= k_end) return;
    const float *__restrict__ A = g.A[z];
    const float *__restrict__ B = g.B[z];
    const int lda = g.lda[z], ldb = g.ldb[z], ldc = g.ldc[z];
    __shared__ __attribute__((aligned(16))) unsigned char lds[3 * (APL + BPL)];
    unsigned char *lds_b = lds + 3 * APL;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wm = (wave / WGN) * 32 * TM, wn = (wave % WGN) * 32 * TN;
    const int li = lane & 31, lk = lane >> 5;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    // transposed-read bases (T10): group tg = lane >> 4 takes columns 16 (tg & 1) .. +15 and k-rows 8 (tg >> 1) + 4 h .. +3
    const int tg = (threadIdx.x >> 4) & 3, tq = (threadIdx.x >> 2) & 3, tp = threadIdx.x & 3;
    const unsigned char *fa[2], *fb[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int lane_off = 64 * (4 * h + tq) + 16 * ((2 * (tg & 1) + (tp >> 1)) ^ (2 * (tg >> 1) + h)) + 8 * (tp & 1);
        fa[h] = lds + (BM / 32 * LG_PLT_SUB) * (tg >> 1) + LG_PLT_SUB * (wm / 32) + lane_off;
        fb[h] = lds_b + (BN / 32 * LG_PLT_SUB) * (tg >> 1) + LG_PLT_SUB * (wn / 32) + lane_off;
    }
    auto frag = [&](const unsigned char *const (&f)[2], int o) -> bf16x8 {
        const s16x4 lo = lds_read_tr16(f[0] + o), hi = lds_read_tr16(f[1] + o);
        typedef short s16x8 __attribute__((ext_vector_type(8)));
        const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    };
    const bool a_vec = (lda & 3) == 0 && ((uintptr_t)A & 15) == 0 && (M & 3) == 0 && M >= 4;
    const bool b_vec = (ldb & 3) == 0 && ((uintptr_t)B & 15) == 0 && (N & 3) == 0 && N >= 4;
    // The whole k-loop once per load form (16-byte loads for both operands, or the element-wise fallback): with the choice as a
    // branch INSIDE the loop hipcc's s_waitcnt insertion lost track of the prefetched registers across the join -- in the
    // steady-state loop below the split read a register set with no vmcnt wait at all (wrong sums), in the guarded form it
    // waited for everything.
    auto run = [&](auto vec) {
        constexpr bool VEC = decltype(vec)::value;
        float4 ra[PD][NVA], rb[PD][NVB];
        unsigned ma[PD], mb[PD];
    #pragma unroll
        for (int d = 0; d < PD; ++d) { ma[d] = 0u; mb[d] = 0u; }
        auto load = [&](int k0, float4 (&xa)[NVA], float4 (&xb)[NVB], unsigned &xma, unsigned &xmb) {
            stage_load<false, BM, VEC, false, NT>(A, lda, m0, k0, M, k_end, xa, xma);
            stage_load<false, BN, VEC, false, NT>(B, ldb, n0, k0, N, k_end, xb, xmb);
        };
    #pragma unroll
        for (int d = 0; d < PD; ++d)
            if (d == 0 || k_begin + d * BK < k_end) load(k_begin + d * BK, ra[d], rb[d], ma[d], mb[d]);
        stage_store_x6t<BM, NT>(lds, ra[0], ma[0]);
        stage_store_x6t<BN, NT>(lds_b, rb[0], mb[0]);
        lds_barrier();
        // one k-tile (in LDS; its register set c is free): loads of tile + PD into set c, MFMAs, then the next tile (set n) to LDS
        // steady: unconditional prefetch and store (see gemm_mainloop_x6: a guarded prefetch makes the store wait for it)
        auto body = [&](auto steady, int k0, float4 (&ca)[NVA], float4 (&cb)[NVB], unsigned &cma, unsigned &cmb, float4 (&na)[NVA], float4 (&nb)[NVB],
                        unsigned &nma, unsigned &nmb) {
            constexpr bool ST = decltype(steady)::value;
            if (ST || k0 + PD * BK < k_end) load(k0 + PD * BK, ca, cb, cma, cmb);
    #pragma unroll
            for (int s = 0; s < BK / 16; ++s) {
                bf16x8 av[TM][3], bv[TN][3];
    #pragma unroll
                for (int a = 0; a < TM; ++a)
    #pragma unroll
                    for (int p = 0; p < 3; ++p) av[a][p] = frag(fa, LG_PLT_SUB * a + p * APL + (BM / 32 * LG_PLT_SUB) * 2 * s);
    #pragma unroll
                for (int b = 0; b < TN; ++b)
    #pragma unroll
                    for (int p = 0; p < 3; ++p) bv[b][p] = frag(fb, LG_PLT_SUB * b + p * BPL + (BN / 32 * LG_PLT_SUB) * 2 * s);
    #pragma unroll
                for (int a = 0; a < TM; ++a)
    #pragma unroll
                    for (int b = 0; b < TN; ++b) {
                        const bf16x8 *x = av[a], *y = bv[b];
                        f32x16 c = acc[a][b];
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[1], y[1], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[0], y[2], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[2], y[0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[0], y[1], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[1], y[0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[0], y[0], c, 0, 0, 0);
                        acc[a][b] = c;
                    }
            }
            lds_barrier();
            if (ST || k0 + BK < k_end) {
                stage_store_x6t<BM, NT>(lds, na, nma);
                stage_store_x6t<BN, NT>(lds_b, nb, nmb);
            }
            lds_barrier();
        };
        int k0 = k_begin;
        if (LG_DW_STEADY)
            for (; k0 + (2 * PD - 1) * BK < k_end; k0 += PD * BK) {
    #pragma unroll
                for (int d = 0; d < PD; ++d)
                    body(std::true_type{}, k0 + d * BK, ra[d], rb[d], ma[d], mb[d], ra[(d + 1) % PD], rb[(d + 1) % PD], ma[(d + 1) % PD], mb[(d + 1) % PD]);
            }
        for (; k0 < k_end; k0 += PD * BK) {
    #pragma unroll
            for (int d = 0; d < PD; ++d)
                if (d == 0 || k0 + d * BK < k_end)
                    body(std::false_type{}, k0 + d * BK, ra[d], rb[d], ma[d], mb[d], ra[(d + 1) % PD], rb[(d + 1) % PD], ma[(d + 1) % PD], mb[(d + 1) % PD]);
        }
    };
    if (a_vec && b_vec) run(std::true_type{});
    else run(std::false_type{});
    // nstore: columns past it are products with the zero pad columns of the padded observations -- computed, not stored
    gemm_epilogue<2, TM, TN, BM, BN, 1>(g, z, M, g.nstore[z] ? g.nstore[z] : N, m0, n0, wm, wn, li, lk, ldc, acc);
}

extern "C" void ppok_debug_set_xcd_remap(int v) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_xcd_remap), &v, sizeof(int)); }
static int g_gemm_dw_t = 1;    // weight gradients through k_gemm_dw_t (0: the transposing-store path of k_gemm, for A/B)
extern "C" void ppok_debug_set_dw_t(int v) { g_gemm_dw_t = v; }

static int g_gemm_dbuf = 0;   // single LDS buffer (34 KB, 4 workgroups/CU) measured 3-15 % faster than double buffering
extern "C" void ppok_debug_set_dbuf(int v) { g_gemm_dbuf = v; }
static int g_gemm_x6 = 1;     // split-bf16 mainloop (fp32 accuracy on the bf16 matrix cores); 0 = fp32-input MFMA
static int g_gemm_w8 = 1;     // 128x128 tile on 8 waves (2x4, each 64x32) instead of 4 waves (2x2, each 64x64)
extern "C" void ppok_debug_set_x6(int v) { g_gemm_x6 = v & 1; g_gemm_w8 = (v >> 1) & 1; }

static int g_gemm_planes = 1;  // forward / input-gradient GEMMs take the weight operand from the optimiser's bf16 planes
extern "C" void ppok_debug_set_planes(int v) { g_gemm_planes = v; }

// forward (EPI 0) and input gradient (EPI 1) with B = pre-split weight planes, reduction-contiguous
#ifdef LG_EXP_KERNELS
static int g_gemm_t96 = 0;     // 96x128 tile where it fills the 512 workgroup slots in fuller rounds: measured slower end to end
extern "C" void ppok_debug_set_t96(int v) { g_gemm_t96 = v; }
#endif

#include "ppo_gemm_glds.h"
#ifdef LG_EXP_KERNELS
#include "exp/ppo_gemm_exp.h"
#endif

template <int EPI, bool B_RC = true, int PL = 1>
static void launch_gemm_pl(const GemmArgs &g, int nz, hipStream_t s) {
    int maxM = 0, maxN = 0;
    for (int z = 0; z < nz; ++z) { maxM = g.M[z] > maxM ? g.M[z] : maxM; maxN = g.N[z] > maxN ? g.N[z] : maxN; }
    {
        // LDS-DMA forward (ppo_gemm_glds.h): 40 KB workgroups, up to four per CU -- update -2.6 % (profiles/r04_ab.txt).  LG_GEMM_GLDS=0: the
        // register-staged k_gemm.  The input-gradient variant (bit 1; bit-identical, tests green) LOSES inside the update, where it shares the
        // CUs with the weight-gradient kernels of the side stream (minibatch 0.472 -> 0.486 ms): exp builds only.
        static const int glds = getenv("LG_GEMM_GLDS") ? atoi(getenv("LG_GEMM_GLDS")) : 1;
#ifdef LG_EXP_KERNELS
        constexpr bool have = true;
#else
        constexpr bool have = EPI == 0;
#endif
        if constexpr (have) {
            if ((glds & (EPI == 0 ? 1 : 2)) && maxM > 64 && glds_ok<PL>(g, nz)) {
                dim3 grid((unsigned)(((maxM + GLDS_BM - 1) / GLDS_BM) * (maxN / GLDS_BN)), 1, nz);
                hipLaunchKernelGGL((k_gemm_glds<EPI, PL>), grid, dim3(256), 0, s, g);
                return;
            }
        }
    }
    const long big_tiles = (long)((maxM + 127) / 128) * ((maxN + 127) / 128);
    if (big_tiles >= 192 && maxN > 64 && maxM > 64) {
        dim3 grid((unsigned)big_tiles, 1, nz);
#ifdef LG_EXP_KERNELS
        // 256 CUs x 2 resident workgroups: time ~ rounds x tile area.  24576 rows in 128-row tiles give 768 or 384
        // workgroups for the 256- and 128-wide layers (1.5 and 0.75 rounds); 96-row tiles give 1024 and 512.
        const long t96 = (long)((maxM + 95) / 96) * ((maxN + 127) / 128);
        const double c128 = (double)((big_tiles * nz + 511) / 512), c96 = 0.75 * (double)((t96 * nz + 511) / 512);
        if (g_gemm_t96 && c96 < c128) {
            hipLaunchKernelGGL((k_gemm<true, B_RC, EPI, 3, 1, false, true, 1, 4, PL>), dim3((unsigned)t96, 1, nz), dim3(256), 0, s, g);
            return;
        }
        static const int pp = getenv("LG_GEMM_PP") ? atoi(getenv("LG_GEMM_PP")) : 0;     // bit 0 forward, bit 1 input gradient
        if ((pp & 1) && EPI == 0 || (pp & 2) && EPI == 1) {
            dim3 gpp((unsigned)(((maxM + 255) / 256) * ((maxN + 127) / 128)), 1, nz);
            hipLaunchKernelGGL((k_gemm_pp<B_RC, EPI, PL>), gpp, dim3(512), 0, s, g);
            return;
        }
        static const int w4 = getenv("LG_GEMM_W4") ? atoi(getenv("LG_GEMM_W4")) : 0;
        if ((w4 & 1) && EPI == 0 || (w4 & 2) && EPI == 1) {
            hipLaunchKernelGGL((k_gemm<true, B_RC, EPI, 2, 2, false, true, 2, 2, PL>), grid, dim3(256), 0, s, g);
            return;
        }
#endif
        hipLaunchKernelGGL((k_gemm<true, B_RC, EPI, 2, 1, false, true, 2, 4, PL>), grid, dim3(512), 0, s, g);
    } else {
        dim3 grid((unsigned)(((maxM + 63) / 64) * ((maxN + 63) / 64)), 1, nz);
#ifdef LG_EXP_KERNELS
        static const int ldb = getenv("LG_GEMM_LDB") ? atoi(getenv("LG_GEMM_LDB")) : 0;
        if (ldb) { hipLaunchKernelGGL((k_gemm<true, B_RC, EPI, 1, 1, false, true, 2, 2, PL, true>), grid, dim3(256), 0, s, g); return; }
#endif
        hipLaunchKernelGGL((k_gemm<true, B_RC, EPI, 1, 1, false, true, 2, 2, PL>), grid, dim3(256), 0, s, g);
    }
}
static bool planes_ok(const GemmArgs &g, int nz) {
    if (!g_gemm_planes || !g_gemm_x6) return false;
    for (int z = 0; z < nz; ++z)
        if (!g.Bpl[z] || (g.ldb[z] & 7) || (g.K[z] & 7) || g.K[z] < 8 || ((uintptr_t)g.Bpl[z] & 15) || (g.pl_stride & 7)) return false;
    return true;
}

template <bool A_RC, bool B_RC, int EPI>
static void launch_gemm(const GemmArgs &g, int nz, int splits, hipStream_t s) {
    int maxM = 0, maxN = 0;
    for (int z = 0; z < nz; ++z) { maxM = g.M[z] > maxM ? g.M[z] : maxM; maxN = g.N[z] > maxN ? g.N[z] : maxN; }
    // small problems (rollout forward on 4096 rows, heads) use 64x64 tiles to fill more CUs
    const long big_tiles = (long)((maxM + 127) / 128) * ((maxN + 127) / 128);
    if (big_tiles * splits >= 192 && maxN > 64 && maxM > 64) {
        dim3 grid((unsigned)big_tiles, splits, nz);
        if (g_gemm_x6 && g_gemm_w8) hipLaunchKernelGGL((k_gemm<A_RC, B_RC, EPI, 2, 1, false, true, 2, 4>), grid, dim3(512), 0, s, g);
        else if (g_gemm_x6) hipLaunchKernelGGL((k_gemm<A_RC, B_RC, EPI, 2, 2, false, true>), grid, dim3(256), 0, s, g);
        else if (g_gemm_dbuf) hipLaunchKernelGGL((k_gemm<A_RC, B_RC, EPI, 2, 2, true>), grid, dim3(256), 0, s, g);
        else hipLaunchKernelGGL((k_gemm<A_RC, B_RC, EPI, 2, 2, false>), grid, dim3(256), 0, s, g);
    } else {
        dim3 grid((unsigned)(((maxM + 63) / 64) * ((maxN + 63) / 64)), splits, nz);
        if (g_gemm_x6) hipLaunchKernelGGL((k_gemm<A_RC, B_RC, EPI, 1, 1, false, true>), grid, dim3(256), 0, s, g);
        else hipLaunchKernelGGL((k_gemm<A_RC, B_RC, EPI, 1, 1, true>), grid, dim3(256), 0, s, g);
    }
}
// planes read along their rows (input gradient): column count and leading dimension in whole 16-byte chunks
static bool planes_t_ok(const GemmArgs &g, int nz) {
    if (!g_gemm_planes || !g_gemm_x6) return false;
    for (int z = 0; z < nz; ++z)
        if (!g.Bpl[z] || (g.ldb[z] & 7) || (g.N[z] & 7) || g.N[z] < 8 || ((uintptr_t)g.Bpl[z] & 15) || (g.pl_stride & 7)) return false;
    return true;
}
// g->Bpl (optional): bf16 planes of W [n_out][n_in] kept by the optimiser step; the caller passes both B (fp32 W) and Bpl,
// whichever path is eligible is taken.  Forward: the planes are the reduction-contiguous operand.  Input gradient: the
// SAME planes, reduced over their rows through the transposing LDS read (no second image of W^T to keep current).
extern "C" void ppok_gemm_fwd(const GemmArgs *g, int nz, hipStream_t s) {
    GemmArgs gp = *g;                          // the plane path's view of the reduction dimension (padded first layer, ppo_device.h)
    for (int z = 0; z < nz; ++z) {
        if (g->Kpl[z]) gp.K[z] = g->Kpl[z];
        if (g->ldbpl[z]) gp.ldb[z] = g->ldbpl[z];
    }
    if (planes_ok(gp, nz)) launch_gemm_pl<0>(gp, nz, s);
    else launch_gemm<true, true, 0>(*g, nz, 1, s);
}
extern "C" void ppok_gemm_dx(const GemmArgs *g, int nz, hipStream_t s) {
    if (planes_t_ok(*g, nz)) launch_gemm_pl<1, false, 2>(*g, nz, s);
    else launch_gemm<true, false, 1>(*g, nz, 1, s);
}
extern "C" void ppok_gemm_dw(const GemmArgs *g, int nz, int splits, hipStream_t s) {
    if (!(g_gemm_dw_t && g_gemm_x6)) {         // k_gemm's store guard is its N: compute the true width only
        GemmArgs gt = *g;
        for (int z = 0; z < nz; ++z) if (g->nstore[z]) gt.N[z] = g->nstore[z];
        launch_gemm<false, false, 2>(gt, nz, splits, s);
        return;
    }
    int maxM = 0, maxN = 0;
    for (int z = 0; z < nz; ++z) { maxM = g->M[z] > maxM ? g->M[z] : maxM; maxN = g->N[z] > maxN ? g->N[z] : maxN; }
    const long big_tiles = (long)((maxM + 127) / 128) * ((maxN + 127) / 128);
    if (big_tiles * splits >= 192 && maxN > 64 && maxM > 64) {
#ifdef LG_EXP_KERNELS
        static const int w4 = getenv("LG_DW_W4") ? atoi(getenv("LG_DW_W4")) : 0;     // A/B: 4 waves of 64x64 per 128x128 tile
        if (w4) { hipLaunchKernelGGL((k_gemm_dw_t<2, 2, 2, 2>), dim3((unsigned)big_tiles, splits, nz), dim3(256), 0, s, *g); return; }
#endif
        hipLaunchKernelGGL((k_gemm_dw_t<2, 1, 2, 4>), dim3((unsigned)big_tiles, splits, nz), dim3(512), 0, s, *g);
    } else {
        const unsigned tiles = ((maxM + 63) / 64) * ((maxN + 63) / 64);
        hipLaunchKernelGGL((k_gemm_dw_t<1, 1, 2, 2>), dim3(tiles, splits, nz), dim3(256), 0, s, *g);
    }
}

// ------------------------------------------------------------------------------------------------
// PPO.act epilogue: a ~ N(mu, sigma), log-prob, transition store (rsl_rl PPO.act / storage.add)
__global__ void k_act_sample(PpoDev P, const float *__restrict__ obs, const float *__restrict__ critic_obs,
                             const float *__restrict__ mu, const float *__restrict__ val, int t, int64_t act_count, int inject) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int N = P.N, A = P.A, O = P.O;
    if (t >= 0) {
        // storage.add of the observations: rows of step t are one contiguous block, so the copy is flat and coalesced
        // (grid-stride over the whole launch), not one strided row per lane
        const size_t tot = (size_t)N * O, step = (size_t)gridDim.x * blockDim.x;
        for (size_t k = i; k < tot; k += step) P.st_obs[(size_t)t * tot + k] = obs[k];
        if (P.st_critic_obs != P.st_obs) {
            const size_t totc = (size_t)N * P.OC;
            for (size_t k = i; k < totc; k += step) P.st_critic_obs[(size_t)t * totc + k] = critic_obs[k];
        }
    }
    const float *std = P.params + P.off_std;
    // sigma the rollout was sampled with (PPO.update's old_sigma_batch); before the row guard: with fewer envs than
    // action dimensions the lanes i in [N, A) would otherwise leave it at zero and the KL term at inf
    if (i < A && t == 0) P.st_sigma[i] = std[i];
    if (i >= N) return;
    float lp = 0.f;
    for (int a = 0; a < A; ++a) {
        float m = mu[(size_t)i * A + a], s = std[a];
        float z = inject ? P.noise[(size_t)i * A + a] : philox_normal(P.seed, (uint32_t)(P.env_offset + i), (uint64_t)act_count, a);
        float act = m + s * z;
        lp += -((act - m) * (act - m)) / (2.0f * s * s) - logf(s) - 0.9189385332046727f;
        P.act_actions[(size_t)i * A + a] = act;
        P.act_mu[(size_t)i * A + a] = m;
        if (t >= 0) {
            P.st_actions[((size_t)t * N + i) * A + a] = act;
            P.st_mu[((size_t)t * N + i) * A + a] = m;
        }
    }
    const float v = val[i];
    P.act_values[i] = v;
    P.act_log_prob[i] = lp;
    if (t >= 0) {
        P.st_values[(size_t)t * N + i] = v;
        P.st_log_prob[(size_t)t * N + i] = lp;
    }
}

// PPO.process_env_step: rewards += gamma * V * time_outs ; store
__global__ void k_process_step(PpoDev P, const float *__restrict__ rew, const uint8_t *__restrict__ dones,
                               const uint8_t *__restrict__ time_outs, int t) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.N) return;
    process_step_body(P, rew[i], dones[i] != 0, time_outs && time_outs[i], t, i);
}

// RolloutStorage.compute_returns: GAE reverse scan, one lane per env; block sums of adv, adv^2
__global__ void __launch_bounds__(256) k_gae(PpoDev P, const float *__restrict__ last_values) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int N = P.N, T = P.T;
    float s1 = 0.f, s2 = 0.f;
    if (i < N) {
        float adv = 0.f, next_v = last_values[i];
        for (int t = T - 1; t >= 0; --t) {
            const size_t k = (size_t)t * N + i;
            const float nnt = 1.0f - (P.st_dones[k] ? 1.0f : 0.0f);
            const float v = P.st_values[k];
            const float delta = P.st_rewards[k] + nnt * P.gamma * next_v - v;
            adv = delta + nnt * P.gamma * P.lam * adv;
            const float ret = adv + v;
            P.st_returns[k] = ret;
            const float a = ret - v;
            P.st_adv[k] = a;
            s1 += a; s2 += a * a;
            next_v = v;
        }
    }
    __shared__ float r1[256], r2[256];
    r1[threadIdx.x] = s1; r2[threadIdx.x] = s2;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) { r1[threadIdx.x] += r1[threadIdx.x + w]; r2[threadIdx.x] += r2[threadIdx.x + w]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        acc_add(P, &P.adv_partial[0], r1[0]);
        acc_add(P, &P.adv_partial[1], r2[0]);
        if (blockIdx.x == 0) P.adv_partial[2] = (float)((size_t)N * T);
    }
}
// advantages = (adv - mean) / (std_unbiased + 1e-8) over all samples (of all ranks after all-reduce)
__global__ void k_adv_normalize(PpoDev P) {
    const size_t n = (size_t)P.N * P.T;
    const float cnt = P.adv_partial[2], mean = P.adv_partial[0] / cnt;
    const float var = fmaxf((P.adv_partial[1] - cnt * mean * mean) / fmaxf(cnt - 1.0f, 1.0f), 0.f);
    const float inv = 1.0f / (sqrtf(var) + 1e-8f);
    for (size_t k = blockIdx.x * (size_t)blockDim.x + threadIdx.x; k < n; k += (size_t)gridDim.x * blockDim.x)
        P.st_adv[k] = (P.st_adv[k] - mean) * inv;
    if (blockIdx.x == 0 && threadIdx.x == 0) { P.stats[6] = mean; P.stats[7] = sqrtf(var); }
}

// mini_batch_generator's randperm(T*N), drawn on the device once per update and reused by every epoch:
// a keyed bijection of [0, 2^b) -- b = bits of n rounded up to even, 6-round balanced Feistel whose round function is one
// Philox block keyed by (seed, update index, round) -- restricted to [0, n) by cycle walking (re-encrypt until the value
// falls below n: a bijection of the superset walks every element of [0, n) to a distinct element of [0, n)).  No sort, no
// host round trip; each lane is independent.  Not torch.randperm's stream (parity with rsl_rl's sample order is unpinned).
__global__ void __launch_bounds__(256) k_randperm(PpoDev P, int n, int half_bits, uint64_t update_idx) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t mask = (1u << half_bits) - 1u;
    uint32_t x = (uint32_t)i;
    do {
        uint32_t l = x >> half_bits, r = x & mask;
#pragma unroll 1
        for (int round = 0; round < 6; ++round) {
            uint32_t c[4] = {r, (uint32_t)round, (uint32_t)update_idx, (uint32_t)(update_idx >> 32) ^ 0x9e3779b9u};
            philox4x32((uint32_t)P.seed, (uint32_t)(P.seed >> 32), c);
            const uint32_t t = l ^ (c[0] & mask);
            l = r; r = t;
        }
        x = (l << half_bits) | r;
    } while (x >= (uint32_t)n);
    P.perm[i] = (int32_t)x;
}

// mini_batch_generator: rows perm[mb*R .. (mb+1)*R) of the (T*N)-flattened storage
__global__ void k_gather(PpoDev P, int mb) {
    const int R = P.mb_rows, A = P.A, O = P.O;
    const int r = blockIdx.x;
    if (r >= R) return;
    const int src = P.perm[(size_t)mb * R + r];
    for (int k = threadIdx.x; k < O; k += blockDim.x) P.mb_obs[(size_t)r * P.Op + k] = P.st_obs[(size_t)src * O + k];
    if (P.st_critic_obs != P.st_obs)
        for (int k = threadIdx.x; k < P.OC; k += blockDim.x) P.mb_critic_obs[(size_t)r * P.OCp + k] = P.st_critic_obs[(size_t)src * P.OC + k];
    if ((int)threadIdx.x < A) {
        P.mb_actions[(size_t)r * A + threadIdx.x] = P.st_actions[(size_t)src * A + threadIdx.x];
        P.mb_mu[(size_t)r * A + threadIdx.x] = P.st_mu[(size_t)src * A + threadIdx.x];
    }
    if (threadIdx.x == 0) {
        P.mb_scalars[(size_t)r * 4 + 0] = P.st_values[src];
        P.mb_scalars[(size_t)r * 4 + 1] = P.st_returns[src];
        P.mb_scalars[(size_t)r * 4 + 2] = P.st_adv[src];
        P.mb_scalars[(size_t)r * 4 + 3] = P.st_log_prob[src];
    }
}

// same gather, 32 lanes per row moving 16 bytes each (obs, actions, mu as float4s; the four scalars as one float4):
// used when A is a multiple of 4.  An observation width that is not (235 rough terrain, 169 Cassie, 65 trajectory task: the
// storage rows are then not 16-byte aligned) is read a float per lane; the destination rows are Op / OCp long either way.
__device__ __forceinline__ void gather4_block(const PpoDev &P, int mb, int vblock) {
    const int R = P.mb_rows, A4 = P.A / 4;
    const int gid = vblock * 256 + threadIdx.x;
    const int r = gid >> 5, j = gid & 31;
    if (r >= R) return;
    const int src = P.perm[(size_t)mb * R + r];
    const bool own_critic = P.st_critic_obs != P.st_obs;
    const int O4 = (P.O & 3) ? 0 : P.O / 4, OC4 = (own_critic && !(P.OC & 3)) ? P.OC / 4 : 0;
    // eight loads in flight per lane, then the stores (a load / store pair per trip is one L2 round trip per trip: the stores may
    // alias the loads as far as the compiler knows)
    auto row_copy = [&](const float *__restrict__ from, float *__restrict__ to, int n) {
        for (int k0 = j; k0 < n; k0 += 256) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = from[min(k0 + 32 * u, n - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (k0 + 32 * u < n) to[k0 + 32 * u] = v[u];
        }
    };
    if (P.O & 3) row_copy(P.st_obs + (size_t)src * P.O, P.mb_obs + (size_t)r * P.Op, P.O);
    if (own_critic && (P.OC & 3)) row_copy(P.st_critic_obs + (size_t)src * P.OC, P.mb_critic_obs + (size_t)r * P.OCp, P.OC);
    for (int k = j; k < O4 + OC4 + 2 * A4 + 1; k += 32) {
        if (k < O4) reinterpret_cast<float4 *>(P.mb_obs)[(size_t)r * (P.Op / 4) + k] = reinterpret_cast<const float4 *>(P.st_obs)[(size_t)src * O4 + k];
        else if (k < O4 + OC4) reinterpret_cast<float4 *>(P.mb_critic_obs)[(size_t)r * (P.OCp / 4) + (k - O4)] = reinterpret_cast<const float4 *>(P.st_critic_obs)[(size_t)src * OC4 + (k - O4)];
        else if (k < O4 + OC4 + A4) reinterpret_cast<float4 *>(P.mb_actions)[(size_t)r * A4 + (k - O4 - OC4)] = reinterpret_cast<const float4 *>(P.st_actions)[(size_t)src * A4 + (k - O4 - OC4)];
        else if (k < O4 + OC4 + 2 * A4) reinterpret_cast<float4 *>(P.mb_mu)[(size_t)r * A4 + (k - O4 - OC4 - A4)] = reinterpret_cast<const float4 *>(P.st_mu)[(size_t)src * A4 + (k - O4 - OC4 - A4)];
        else reinterpret_cast<float4 *>(P.mb_scalars)[r] = make_float4(P.st_values[src], P.st_returns[src], P.st_adv[src], P.st_log_prob[src]);
    }
}
__global__ void __launch_bounds__(256) k_gather4(PpoDev P, int mb) { gather4_block(P, mb, blockIdx.x); }

// PPO.update loss for one minibatch: surrogate, clipped value loss, entropy bonus, KL(old || new);
// emits d loss / d mu (R x A), d loss / d value (R), and block-reduced d loss / d std, bias grads
// of both heads and the loss statistics.
__global__ void __launch_bounds__(256) k_loss(PpoDev P, const float *__restrict__ mu_new, const float *__restrict__ v_new,
                                              float *__restrict__ dmu, float *__restrict__ dval) {
    const int R = P.mb_rows, A = P.A;
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    const float *std = P.params + P.off_std;
    const float invR = 1.0f / (float)R;
    // per-thread partials at fixed slots: [0..MA) dstd, [MA..2MA) dbias_actor_head, 2MA dbias_critic_head, kl, vloss, sloss
    constexpr int MA = LG_PPO_MAX_A;
    float part[2 * MA + 4];
#pragma unroll
    for (int k = 0; k < 2 * MA + 4; ++k) part[k] = 0.f;
    // the two logarithms of every action dimension depend on sigma only: once per block instead of once per row
    __shared__ float s_logs[MA], s_logr[MA];
    if ((int)threadIdx.x < A) {
        const float s = std[threadIdx.x], so = P.st_sigma[threadIdx.x];
        s_logs[threadIdx.x] = logf(s);
        s_logr[threadIdx.x] = logf(s / so + 1.e-5f);
    }
    __syncthreads();
    if (r < R) {
        const float4 sc = reinterpret_cast<const float4 *>(P.mb_scalars)[r];
        const float v_old = sc.x, ret = sc.y, adv = sc.z, lp_old = sc.w;
        float lp = 0.f, kl = 0.f, dd[MA];
#pragma unroll
        for (int a = 0; a < MA; ++a) {
            dd[a] = 0.f;
            if (a < A) {
                const float s = std[a], so = P.st_sigma[a];
                const float m = mu_new[(size_t)r * A + a], mo = P.mb_mu[(size_t)r * A + a];
                const float d = P.mb_actions[(size_t)r * A + a] - m;
                dd[a] = d;
                lp += -(d * d) / (2.0f * s * s) - s_logs[a] - 0.9189385332046727f;
                kl += s_logr[a] + (so * so + (mo - m) * (mo - m)) / (2.0f * s * s) - 0.5f;
            }
        }
        const float ratio = expf(lp - lp_old);
        const float rc = fminf(fmaxf(ratio, 1.0f - P.clip), 1.0f + P.clip);
        const float s1 = -adv * ratio, s2 = -adv * rc;
        const float dl_dlp = (s1 >= s2 ? -adv : 0.0f) * ratio * invR;        // torch.max ties: see DESIGN.md
#pragma unroll
        for (int a = 0; a < MA; ++a)
            if (a < A) {
                const float s = std[a], d = dd[a];
                const float g = dl_dlp * d / (s * s);
                dmu[(size_t)r * A + a] = g;
                part[MA + a] = g;
                part[a] = dl_dlp * (d * d / (s * s * s) - 1.0f / s) - P.entropy_coef * invR / s;
            }
        const float v = v_new[r];
        float lv, dv;
        if (P.clipped_value) {
            const float dvv = v - v_old;
            const float vc = v_old + fminf(fmaxf(dvv, -P.clip), P.clip);
            const float l1 = (v - ret) * (v - ret), l2 = (vc - ret) * (vc - ret);
            const float inside = (dvv >= -P.clip && dvv <= P.clip) ? 1.0f : 0.0f;
            lv = fmaxf(l1, l2);
            if (l1 > l2) dv = 2.0f * (v - ret);
            else if (l1 < l2) dv = 2.0f * (vc - ret) * inside;
            else dv = (v - ret) + (vc - ret) * inside;
        } else {
            lv = (ret - v) * (ret - v);
            dv = 2.0f * (v - ret);
        }
        dv *= P.value_coef * invR;
        dval[r] = dv;
        part[2 * MA] = dv;
        part[2 * MA + 1] = kl;
        part[2 * MA + 2] = lv;
        part[2 * MA + 3] = fmaxf(s1, s2);
    }
    // wave64 butterfly per partial, one LDS slot per wave (summed in wave order below), one global atomic per block and partial
    __shared__ float red[4][2 * MA + 4];
#pragma unroll
    for (int k = 0; k < 2 * MA + 4; ++k) {
        if (k >= 2 * MA || (k % MA) < A) {
            float v = part[k];
#pragma unroll
            for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m);
            if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = v;
        }
    }
    __syncthreads();
    const int k = threadIdx.x;
    if (k < 2 * MA + 4 && (k >= 2 * MA || (k % MA) < A)) {
        const float v = ((red[0][k] + red[1][k]) + red[2][k]) + red[3][k];
        if (k < MA) acc_add(P, &P.grads[P.off_std + k], v);
        else if (k < 2 * MA) acc_add(P, &P.grads[P.off_bias_actor_head + (k - MA)], v);
        else if (k == 2 * MA) acc_add(P, &P.grads[P.off_bias_critic_head], v);
        else if (k == 2 * MA + 1) acc_add(P, &P.grads[P.num_params], v);             // KL sum rides in the grad buffer tail
        else if (k == 2 * MA + 2) acc_add(P, &P.loss_acc[0], v);
        else acc_add(P, &P.loss_acc[1], v);
    }
}

// ------------------------------------------------------------------------------------------------
// Fused head: [actor head 128->A | critic head 128->1] forward + PPO loss + head backward, one
// workgroup per 64 minibatch rows.  Replaces four launches (head GEMM, k_loss, head weight-gradient
// GEMM, head input-gradient GEMM) whose matrices are too thin for the MFMA tiles: the last hidden
// activations of both nets are staged once in LDS and reused for mu/V, for d(loss)/d(act3) and for
// the outer-product weight gradients.  H3 = last hidden width (<= 128).
#define HEAD_ROWS 64
#define HEAD_GRID 192
#define HEAD_NET_GRID 384                                  // workgroups per network of k_head_net (= scratch rows per network)
#define HEAD_PART_STRIDE(H3) ((LG_PPO_MAX_A + 1) * (H3) + 2 * LG_PPO_MAX_A + 4)
template <int H3>
__global__ void __launch_bounds__(256) k_head_fused(PpoDev P, const float *__restrict__ xa_g, const float *__restrict__ xc_g,
                                                    float *__restrict__ dza_g, float *__restrict__ dzc_g, int64_t w_a, int64_t b_a,
                                                    int64_t w_c, int64_t b_c, int64_t b_prev_a, int64_t b_prev_c) {
    constexpr int MA = LG_PPO_MAX_A, LDX = H3 + 1, NH = 256 / H3;   // NH row groups run concurrently in the backward part
    const int R = P.mb_rows, A = P.A, tid = threadIdx.x;
    __shared__ float xa[HEAD_ROWS * LDX], xc[HEAD_ROWS * LDX];
    __shared__ float wa[MA * H3], wc[H3];
    __shared__ float mus[HEAD_ROWS][MA + 1], dmus[HEAD_ROWS][MA + 1];    // [..][MA]: value / d value
    // per-action constants of the Gaussian terms (row independent): sigma_old^2, 1/(2 s^2), 1/s^2, 1/s, ln s + ln sqrt(2 pi),
    // ln(s / s_old + 1e-5) -- so the per-row loss needs no log and no division
    __shared__ float red[2 * MA + 4], s_so2[MA], s_i2s2[MA], s_is2[MA], s_is[MA], s_lgs[MA], s_klc[MA];
    if (tid < MA) {
        const float sg = tid < A ? P.params[P.off_std + tid] : 1.f, so = tid < A ? P.st_sigma[tid] : 1.f;
        s_so2[tid] = so * so; s_i2s2[tid] = 1.0f / (2.0f * sg * sg); s_is2[tid] = 1.0f / (sg * sg); s_is[tid] = 1.0f / sg;
        s_lgs[tid] = logf(sg) + 0.9189385332046727f; s_klc[tid] = logf(sg / so + 1.e-5f) - 0.5f;
    }
    for (int i = tid; i < A * H3; i += 256) wa[i] = P.params[w_a + i];
    for (int i = tid; i < H3; i += 256) wc[i] = P.params[w_c + i];
    if (tid < 2 * MA + 4) red[tid] = 0.f;
    // backward accumulators of this thread's column, kept in registers across all row tiles of the block
    const int c = tid % H3, half = tid / H3;
    float dwa[MA], dwc = 0.f, dba = 0.f, dbc = 0.f, part[2 * MA + 4];
#pragma unroll
    for (int a = 0; a < MA; ++a) dwa[a] = 0.f;
#pragma unroll
    for (int k = 0; k < 2 * MA + 4; ++k) part[k] = 0.f;
    const int ntiles = (R + HEAD_ROWS - 1) / HEAD_ROWS;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int r0 = tile * HEAD_ROWS;
        __syncthreads();                                   // previous tile fully consumed (also covers the weight staging)
        {   // stage both activation tiles: unconditional 16-byte loads from clamped rows (rows past R are
            // never used: every consumer below checks r < R), issued as one batch
            constexpr int NV = HEAD_ROWS * H3 / 4 / 256;
            float4 va[NV], vc[NV];
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int i = tid + v * 256, r = i / (H3 / 4), c4 = i % (H3 / 4);
                const size_t row = (size_t)min(r0 + r, R - 1);
                va[v] = *reinterpret_cast<const float4 *>(xa_g + row * H3 + 4 * c4);
                vc[v] = *reinterpret_cast<const float4 *>(xc_g + row * H3 + 4 * c4);
            }
            __builtin_amdgcn_sched_barrier(0);       // keep the 2*NV loads in flight together (hipcc otherwise pairs each with its LDS store)
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int i = tid + v * 256, r = i / (H3 / 4), c4 = i % (H3 / 4);
                float *da = xa + r * LDX + 4 * c4, *dc = xc + r * LDX + 4 * c4;
                da[0] = va[v].x; da[1] = va[v].y; da[2] = va[v].z; da[3] = va[v].w;
                dc[0] = vc[v].x; dc[1] = vc[v].y; dc[2] = vc[v].z; dc[3] = vc[v].w;
            }
        }
        __syncthreads();
        {   // head forward: lane = row (conflict-free x reads, broadcast weight reads); wave w -> outputs w, w+4, w+8, w+12
            const int r = tid & 63, w = tid >> 6;
            const float *x = xa + r * LDX;
            float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
            const float *w0 = wa + (w < A ? w : 0) * H3, *w1 = wa + (w + 4 < A ? w + 4 : 0) * H3;
            const float *w2 = wa + (w + 8 < A ? w + 8 : 0) * H3, *w3 = wa + (w + 12 < A ? w + 12 : 0) * H3;
#pragma unroll 8
            for (int k = 0; k < H3; ++k) {
                const float xv = x[k];
                o0 += xv * w0[k]; o1 += xv * w1[k]; o2 += xv * w2[k]; o3 += xv * w3[k];
            }
            if (w < A) mus[r][w] = o0 + P.params[b_a + w];
            if (w + 4 < A) mus[r][w + 4] = o1 + P.params[b_a + w + 4];
            if (w + 8 < A) mus[r][w + 8] = o2 + P.params[b_a + w + 8];
            if (w + 12 < A) mus[r][w + 12] = o3 + P.params[b_a + w + 12];
            if (w == 3) {                                  // wave 3 carries the fewest actor outputs: it also does the value
                float sacc = P.params[b_c];
                const float *y = xc + r * LDX;
#pragma unroll 8
                for (int k = 0; k < H3; ++k) sacc += y[k] * wc[k];
                mus[r][MA] = sacc;
            }
        }
        __syncthreads();
        if (tid < HEAD_ROWS) {   // loss of one row (same arithmetic as k_loss), wave 0 only; partials stay in registers
            const int r = r0 + tid;
            const float invR = 1.0f / (float)R;
            float dvl = 0.f;
#pragma unroll
            for (int a = 0; a < MA; ++a) dmus[tid][a] = 0.f;
            // operands of this row: unconditional loads from a clamped row / action index, issued up front
            const size_t rr = (size_t)min(r, R - 1);
            const float4 sc = reinterpret_cast<const float4 *>(P.mb_scalars)[rr];
            float act_r[MA], mo_r[MA];
#pragma unroll
            for (int a = 0; a < MA; ++a) {
                const int ac = min(a, A - 1);
                act_r[a] = P.mb_actions[rr * A + ac];
                mo_r[a] = P.mb_mu[rr * A + ac];
            }
            if (r < R) {
                const float v_old = sc.x, ret = sc.y, adv = sc.z, lp_old = sc.w;
                float lp = 0.f, kl = 0.f, dd[MA];
#pragma unroll
                for (int a = 0; a < MA; ++a) {
                    dd[a] = 0.f;
                    if (a < A) {
                        const float m = mus[tid][a], mo = mo_r[a];
                        const float d = act_r[a] - m;
                        dd[a] = d;
                        lp += -(d * d) * s_i2s2[a] - s_lgs[a];
                        kl += s_klc[a] + (s_so2[a] + (mo - m) * (mo - m)) * s_i2s2[a];
                    }
                }
                const float ratio = expf(lp - lp_old);
                const float rc = fminf(fmaxf(ratio, 1.0f - P.clip), 1.0f + P.clip);
                const float s1 = -adv * ratio, s2 = -adv * rc;
                const float dl_dlp = (s1 >= s2 ? -adv : 0.0f) * ratio * invR;
#pragma unroll
                for (int a = 0; a < MA; ++a)
                    if (a < A) {
                        const float d = dd[a];
                        const float g = dl_dlp * d * s_is2[a];
                        dmus[tid][a] = g;
                        part[MA + a] += g;
                        part[a] += dl_dlp * (d * d * s_is2[a] * s_is[a] - s_is[a]) - P.entropy_coef * invR * s_is[a];
                    }
                const float v = mus[tid][MA];
                float lv, dv;
                if (P.clipped_value) {
                    const float dvv = v - v_old;
                    const float vc = v_old + fminf(fmaxf(dvv, -P.clip), P.clip);
                    const float l1 = (v - ret) * (v - ret), l2 = (vc - ret) * (vc - ret);
                    const float inside = (dvv >= -P.clip && dvv <= P.clip) ? 1.0f : 0.0f;
                    lv = fmaxf(l1, l2);
                    if (l1 > l2) dv = 2.0f * (v - ret);
                    else if (l1 < l2) dv = 2.0f * (vc - ret) * inside;
                    else dv = (v - ret) + (vc - ret) * inside;
                } else {
                    lv = (ret - v) * (ret - v);
                    dv = 2.0f * (v - ret);
                }
                dvl = dv * P.value_coef * invR;
                part[2 * MA] += dvl;
                part[2 * MA + 1] += kl;
                part[2 * MA + 2] += lv;
                part[2 * MA + 3] += fmaxf(s1, s2);
            }
            dmus[tid][MA] = dvl;
        }
        __syncthreads();
        // head backward for column c: dz3 = (dmu . W) * ELU'(act3) ; dW += dmu^T act3 ; colsum(dz3) -> previous bias grad
        if (half < NH) {
            float wcol[MA];
#pragma unroll
            for (int a = 0; a < MA; ++a) wcol[a] = a < A ? wa[a * H3 + c] : 0.f;
            const float wcc = wc[c];
            for (int r = half; r < HEAD_ROWS && r0 + r < R; r += NH) {
                float g = 0.f;
                const float x = xa[r * LDX + c], y = xc[r * LDX + c];
#pragma unroll
                for (int a = 0; a < MA; ++a) {
                    const float dm = dmus[r][a];
                    g += dm * wcol[a];
                    dwa[a] += dm * x;
                }
                const float dz = g * (x > 0.f ? 1.0f : x + 1.0f);
                const float dv = dmus[r][MA];
                const float dzc = dv * wcc * (y > 0.f ? 1.0f : y + 1.0f);
                dza_g[(size_t)(r0 + r) * H3 + c] = dz;
                dzc_g[(size_t)(r0 + r) * H3 + c] = dzc;
                dwc += dv * y;
                dba += dz;
                dbc += dzc;
            }
        }
    }
    // ---- flush: loss partials of the row lanes (wave 0) transposed through LDS (lane r writes column r, thread k adds row k:
    // a 36 x 6 shuffle butterfly on one wave is several times slower), then column accumulators reduced over the NH row groups
    __shared__ float ptmp[(2 * MA + 4) * HEAD_ROWS];
    if (tid < HEAD_ROWS) {
#pragma unroll
        for (int k = 0; k < 2 * MA + 4; ++k) ptmp[k * HEAD_ROWS + tid] = part[k];
    }
    __syncthreads();
    if (tid < 2 * MA + 4 && (tid >= 2 * MA || (tid % MA) < A)) {
        const int k = tid;
        float v = 0.f;
        for (int r = 0; r < HEAD_ROWS; ++r) v += ptmp[k * HEAD_ROWS + r];
        if (k < MA) acc_add(P, &P.grads[P.off_std + k], v);
        else if (k < 2 * MA) acc_add(P, &P.grads[b_a + (k - MA)], v);
        else if (k == 2 * MA) acc_add(P, &P.grads[b_c], v);
        else if (k == 2 * MA + 1) acc_add(P, &P.grads[P.num_params], v);
        else if (k == 2 * MA + 2) acc_add(P, &P.loss_acc[0], v);
        else acc_add(P, &P.loss_acc[1], v);
    }
    float *acc = xa;                                      // reuse the tile buffer: [MA + 3][NH][H3]
    __syncthreads();
    if (half < NH) {
#pragma unroll
        for (int a = 0; a < MA; ++a) acc[(a * NH + half) * H3 + c] = dwa[a];
        acc[((MA + 0) * NH + half) * H3 + c] = dwc;
        acc[((MA + 1) * NH + half) * H3 + c] = dba;
        acc[((MA + 2) * NH + half) * H3 + c] = dbc;
    }
    __syncthreads();
    for (int i = tid; i < (MA + 3) * H3; i += 256) {
        const int q = i / H3, cc = i % H3;
        float v = 0.f;
        for (int h = 0; h < NH; ++h) v += acc[(q * NH + h) * H3 + cc];
        if (q < MA) { if (q < A) acc_add(P, &P.grads[w_a + (int64_t)q * H3 + cc], v); }
        else if (q == MA) acc_add(P, &P.grads[w_c + cc], v);
        else if (q == MA + 1) acc_add(P, &P.grads[b_prev_a + cc], v);
        else acc_add(P, &P.grads[b_prev_c + cc], v);
    }
}

// The same fused head with one network per workgroup (blockIdx.y: 0 actor, 1 critic).  The two heads share nothing but the
// row index -- surrogate / entropy / KL need mu only, the value loss needs V only -- so each workgroup stages ONE activation
// tile: 50 KB of LDS instead of 84, three workgroups per CU (all 768 of a 24576-row minibatch resident).  Used for H3 = 128, where
// k_head_fused is one wave per SIMD.
//
// Round 4: the arithmetic is the old kernel's, the operand paths are not.  With a row per lane and the head weights and the rows'
// output gradients read from LDS for every multiply-add, the kernel moved 1.25 LDS dwords per FMA and the twelve waves of a CU queued
// on the LDS port for 13 us of its 29.  Now every operand that is uniform over a wave comes through the SCALAR path:
//   forward   lane = row, wave = a quarter of k; the weights are wave-uniform and arrive by s_load from global memory (K$), 32 x reads
//             per lane instead of 640; the four partial sums per (row, output) meet in LDS, added in a fixed order
//   loss      every wave evaluates the 64 rows (lane = row), so each wave holds d loss / d out of all rows in its own registers; wave 0
//             alone accumulates the row sums
//   backward  wave = every fourth row, lane = columns (lane, lane + 64): a row's output gradients are v_readlane'd into SGPRs and feed
//             the multiply-adds as scalar operands; one LDS read per column and row
template <int H3, int NA>
__global__ void __launch_bounds__(256, 3) k_head_net(PpoDev P, const float *__restrict__ xa_g, const float *__restrict__ xc_g,
                                                     float *__restrict__ dza_g, float *__restrict__ dzc_g, const float *__restrict__ wa_g,
                                                     const float *__restrict__ wc_g, int64_t b_a, int64_t b_c) {
    static_assert(H3 == 128, "lane = columns (lane, lane + 64); wave = a quarter of k");
    static_assert(NA <= LG_PPO_MAX_A, "NA = the action count the register arrays are sized for (>= P.A); the scratch row keeps LG_PPO_MAX_A slots");
    constexpr int MA = LG_PPO_MAX_A, LDX = H3 + 1, NWV = 4, KQ = H3 / NWV;
    const int R = P.mb_rows, A = P.A, tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool actor = blockIdx.y == 0;
    const float *__restrict__ x_g = actor ? xa_g : xc_g;
    const float *__restrict__ w_g = actor ? wa_g : wc_g;
    float *__restrict__ dz_g = actor ? dza_g : dzc_g;
    const int nout = actor ? A : 1;
    // x tile, then the partial outputs [NWV][HEAD_ROWS][NA + 1]; after the tile loop the same floats hold the column sums and the row sums.
    // s_am: the tile's rows of the minibatch's actions and old means, staged with the tile (the loss lanes' own loads of them sank to
    // their use under the 168-register cap and cost the actor's loss phase 6 us of exposed latency: stamps, round 4)
    // (52.9 KB in all: three workgroups per CU is 158.8 of the 160 KB -- one more KB and a third of the workgroups start a round late)
    constexpr int OP = NA + 1, NBUF = HEAD_ROWS * LDX + NWV * HEAD_ROWS * OP;
    __shared__ float buf[NBUF];
    __shared__ float s_am[2 * HEAD_ROWS * NA];
    __shared__ float s_so2[MA], s_i2s2[MA], s_is2[MA], s_is[MA], s_lgs[MA], s_klc[MA], s_bias[MA];
    float *x = buf, *outs_p = buf + HEAD_ROWS * LDX;
    if (tid < MA) {
        const float sg = tid < A ? P.params[P.off_std + tid] : 1.f, so = tid < A ? P.st_sigma[tid] : 1.f;
        s_so2[tid] = so * so; s_i2s2[tid] = 1.0f / (2.0f * sg * sg); s_is2[tid] = 1.0f / (sg * sg); s_is[tid] = 1.0f / sg;
        s_lgs[tid] = logf(sg) + 0.9189385332046727f; s_klc[tid] = logf(sg / so + 1.e-5f) - 0.5f;
        s_bias[tid] = tid < nout ? P.params[(actor ? b_a : b_c) + tid] : 0.f;
    }
    float wcol0[NA], wcol1[NA];                             // the head weights of this lane's two columns (backward)
#pragma unroll
    for (int a = 0; a < NA; ++a) {
        wcol0[a] = a < nout ? w_g[a * H3 + lane] : 0.f;
        wcol1[a] = a < nout ? w_g[a * H3 + lane + 64] : 0.f;
    }
    float dw0[NA], dw1[NA], db0 = 0.f, db1 = 0.f, part[2 * MA + 4];
#pragma unroll
    for (int a = 0; a < NA; ++a) { dw0[a] = 0.f; dw1[a] = 0.f; }
#pragma unroll
    for (int k = 0; k < 2 * MA + 4; ++k) part[k] = 0.f;
    const int ntiles = (R + HEAD_ROWS - 1) / HEAD_ROWS;
    const float invR = 1.0f / (float)R;
#ifdef LG_HEAD_STAMPS
    unsigned long long ts[8]; ts[0] = __builtin_amdgcn_s_memrealtime();
#define HSTAMP(k) ts[k] = __builtin_amdgcn_s_memrealtime()
#else
#define HSTAMP(k) do { } while (0)
#endif
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int r0 = tile * HEAD_ROWS;
        // what the loss needs of this lane's row, requested before anything waits
        const int r = r0 + lane;
        const size_t rr = (size_t)min(r, R - 1);
        const float4 sc = reinterpret_cast<const float4 *>(P.mb_scalars)[rr];
        __syncthreads();
        if (actor) {                                        // rows r0 .. r0 + 63 of [R][A]: one contiguous run
            const size_t base = (size_t)r0 * A, lim = (size_t)R * A;
            for (int i = tid; i < HEAD_ROWS * A; i += 256) {
                const size_t g = base + i < lim ? base + i : lim - 1;
                s_am[i] = P.mb_actions[g];
                s_am[HEAD_ROWS * NA + i] = P.mb_mu[g];
            }
        }
        {
            constexpr int NV = HEAD_ROWS * H3 / 4 / 256;
            float4 vx[NV];
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int i = tid + v * 256, rw = i / (H3 / 4), c4 = i % (H3 / 4);
                vx[v] = *reinterpret_cast<const float4 *>(x_g + (size_t)min(r0 + rw, R - 1) * H3 + 4 * c4);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int i = tid + v * 256, rw = i / (H3 / 4), c4 = i % (H3 / 4);
                float *d = x + rw * LDX + 4 * c4;
                d[0] = vx[v].x; d[1] = vx[v].y; d[2] = vx[v].z; d[3] = vx[v].w;
            }
        }
        __syncthreads();
        HSTAMP(1);
        {   // forward: lane = row, wave = k quarter; weights through the scalar cache.  One guard per output (not per multiply-add group):
            // the scheduler works inside basic blocks, and a row's 32 activations stay in registers across the outputs
            const float *xr = x + lane * LDX + KQ * wv;
            const float *__restrict__ wq = w_g + KQ * wv;
            float xv[KQ];
#pragma unroll
            for (int j = 0; j < KQ; ++j) xv[j] = xr[j];
#pragma unroll
            for (int a = 0; a < NA; ++a)
                if (a < nout) {
                    float o = 0.f;
#pragma unroll
                    for (int j = 0; j < KQ; ++j) o = __builtin_fmaf(xv[j], wq[a * H3 + j], o);
                    outs_p[(wv * HEAD_ROWS + lane) * OP + a] = o;
                }
        }
        __syncthreads();
        HSTAMP(2);
        float dout[NA];                                     // d loss / d out of row `lane`, in every wave
#pragma unroll
        for (int a = 0; a < NA; ++a) dout[a] = 0.f;
        {
            float outv[NA];
#pragma unroll
            for (int a = 0; a < NA; ++a) {
                outv[a] = 0.f;
                if (a < nout) {
                    const float *op = outs_p + lane * OP + a;
                    constexpr int QS = HEAD_ROWS * OP;
                    outv[a] = (op[0] + op[QS]) + (op[2 * QS] + op[3 * QS]);
                }
            }
            const bool sums = wv == 0;                      // the row sums are wave 0's
            if (actor) {
                if (r < R) {
                    const float adv = sc.z, lp_old = sc.w;
                    float lp = 0.f, kl = 0.f, dd[NA];
#pragma unroll
                    for (int a = 0; a < NA; ++a) {
                        dd[a] = 0.f;
                        if (a < A) {
                            const float m = outv[a] + s_bias[a], mo = s_am[HEAD_ROWS * NA + lane * A + a];
                            const float d = s_am[lane * A + a] - m;
                            dd[a] = d;
                            lp += -(d * d) * s_i2s2[a] - s_lgs[a];
                            kl += s_klc[a] + (s_so2[a] + (mo - m) * (mo - m)) * s_i2s2[a];
                        }
                    }
                    const float ratio = expf(lp - lp_old);
                    const float rc = fminf(fmaxf(ratio, 1.0f - P.clip), 1.0f + P.clip);
                    const float s1 = -adv * ratio, s2 = -adv * rc;
                    const float dl_dlp = (s1 >= s2 ? -adv : 0.0f) * ratio * invR;
#pragma unroll
                    for (int a = 0; a < NA; ++a)
                        if (a < A) {
                            const float d = dd[a];
                            const float g = dl_dlp * d * s_is2[a];
                            dout[a] = g;
                            if (sums) {
                                part[MA + a] += g;
                                part[a] += dl_dlp * (d * d * s_is2[a] * s_is[a] - s_is[a]) - P.entropy_coef * invR * s_is[a];
                            }
                        }
                    if (sums) { part[2 * MA + 1] += kl; part[2 * MA + 3] += fmaxf(s1, s2); }
                }
            } else if (r < R) {
                const float v_old = sc.x, ret = sc.y;
                const float v = s_bias[0] + outv[0];
                float lv, dv;
                if (P.clipped_value) {
                    const float dvv = v - v_old;
                    const float vc = v_old + fminf(fmaxf(dvv, -P.clip), P.clip);
                    const float l1 = (v - ret) * (v - ret), l2 = (vc - ret) * (vc - ret);
                    const float inside = (dvv >= -P.clip && dvv <= P.clip) ? 1.0f : 0.0f;
                    lv = fmaxf(l1, l2);
                    if (l1 > l2) dv = 2.0f * (v - ret);
                    else if (l1 < l2) dv = 2.0f * (vc - ret) * inside;
                    else dv = (v - ret) + (vc - ret) * inside;
                } else {
                    lv = (ret - v) * (ret - v);
                    dv = 2.0f * (v - ret);
                }
                const float dvl = dv * P.value_coef * invR;
                dout[0] = dvl;
                if (sums) { part[2 * MA] += dvl; part[2 * MA + 2] += lv; }
            }
        }
        // backward for rows wv, wv + 4, ...: dz = (dout . W) act'(x); dW += dout^T x; the row's dout as scalars.  Written twice: without
        // the per-output guard when the register arrays are exactly as long as the head is wide (the actor of every registered task)
        HSTAMP(3);
        auto backward = [&](auto full) {
            for (int rw = wv; rw < HEAD_ROWS && r0 + rw < R; rw += NWV) {
                const float xv0 = x[rw * LDX + lane], xv1 = x[rw * LDX + lane + 64];
                float g0 = 0.f, g1 = 0.f;
#pragma unroll
                for (int a = 0; a < NA; ++a)
                    if (decltype(full)::value || a < nout) {
                        const float sa = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dout[a]), rw));
                        g0 = __builtin_fmaf(sa, wcol0[a], g0); g1 = __builtin_fmaf(sa, wcol1[a], g1);
                        dw0[a] = __builtin_fmaf(sa, xv0, dw0[a]); dw1[a] = __builtin_fmaf(sa, xv1, dw1[a]);
                    }
                const float dz0 = g0 * (xv0 > 0.f ? 1.0f : xv0 + 1.0f), dz1 = g1 * (xv1 > 0.f ? 1.0f : xv1 + 1.0f);
                float *dzr = dz_g + (size_t)(r0 + rw) * H3;
                dzr[lane] = dz0; dzr[lane + 64] = dz1;
                db0 += dz0; db1 += dz1;
            }
        };
        if (nout == NA) backward(std::true_type{}); else backward(std::false_type{});
        HSTAMP(4);
    }
    // Sums over the rows of this workgroup leave through a scratch row, not through atomics: 768 workgroups adding into the
    // same few dozen addresses serialise in L2 (24 us of the first version's 46 were that queue).  k_head_finish folds the rows.
    // Column sums: the four waves' partials [MA + 1][NWV][H3] meet in LDS; the loss lanes' row sums (wave 0) are transposed through
    // LDS (lane r writes its partials as column r, thread k adds the 64 entries of row k).
    float *acc = buf, *ptmp = buf + (MA + 1) * NWV * H3;
    static_assert((MA + 1) * NWV * H3 + (2 * MA + 4) * HEAD_ROWS <= NBUF, "sums fit in the tile buffer");
    __syncthreads();                                        // every wave is done with the tile
#pragma unroll
    for (int a = 0; a < MA; ++a) {
        acc[(a * NWV + wv) * H3 + lane] = a < NA ? dw0[a < NA ? a : 0] : 0.f;
        acc[(a * NWV + wv) * H3 + lane + 64] = a < NA ? dw1[a < NA ? a : 0] : 0.f;
    }
    acc[(MA * NWV + wv) * H3 + lane] = db0;
    acc[(MA * NWV + wv) * H3 + lane + 64] = db1;
    if (tid < HEAD_ROWS) {
#pragma unroll
        for (int k = 0; k < 2 * MA + 4; ++k) ptmp[k * HEAD_ROWS + tid] = part[k];
    }
    float *__restrict__ prow = P.head_part + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * HEAD_PART_STRIDE(H3);
    __syncthreads();
    if (tid < 2 * MA + 4) {
        float v = 0.f;
        for (int rw = 0; rw < HEAD_ROWS; ++rw) v += ptmp[tid * HEAD_ROWS + rw];
        prow[(MA + 1) * H3 + tid] = v;
    }
    for (int i = tid; i < (MA + 1) * H3; i += 256) {
        const int q = i / H3, cc = i % H3;
        const float *ap = acc + q * NWV * H3 + cc;
        prow[i] = (ap[0] + ap[H3]) + (ap[2 * H3] + ap[3 * H3]);
    }
#ifdef LG_HEAD_STAMPS
    HSTAMP(5);
    if (tid == 0 && (blockIdx.x % 97) == 0)
        printf("head wg %d net %d: stage %llu fwd %llu loss %llu bwd %llu sums %llu (x10 ns) start %llu\n", blockIdx.x, blockIdx.y, ts[1] - ts[0], ts[2] - ts[1],
               ts[3] - ts[2], ts[4] - ts[3], ts[5] - ts[4], ts[0] % 100000ull);
#endif
}

// Folds the scratch rows of k_head_net into the gradient buffer: thread = one sum, blockIdx.y = a chunk of the rows,
// blockIdx.z = the network; HEAD_FIN_CHUNKS-way atomics instead of 768-way.
#define HEAD_FIN_CHUNKS 16
template <int H3>
__global__ void __launch_bounds__(256) k_head_finish(PpoDev P, int nrows, int64_t w_a, int64_t b_a, int64_t w_c, int64_t b_c, int64_t b_prev_a,
                                                     int64_t b_prev_c) {
    constexpr int MA = LG_PPO_MAX_A, NW = (MA + 1) * H3, NTOT = NW + 2 * MA + 4;
    const int i = blockIdx.x * 256 + threadIdx.x, A = P.A;
    if (i >= NTOT) return;
    const bool actor = blockIdx.z == 0;
    const int nout = actor ? A : 1;
    // drop the sums nobody consumes before reading anything
    if (i < NW) { if (i / H3 < MA && i / H3 >= nout) return; }
    else {
        const int k = i - NW;
        const bool used = actor ? ((k < MA && k < A) || (k >= MA && k < 2 * MA && k - MA < A) || k == 2 * MA + 1 || k == 2 * MA + 3)
                                : (k == 2 * MA || k == 2 * MA + 2);
        if (!used) return;
    }
    const int r0 = (int)((long)nrows * blockIdx.y / HEAD_FIN_CHUNKS), r1 = (int)((long)nrows * (blockIdx.y + 1) / HEAD_FIN_CHUNKS);
    const float *__restrict__ src = P.head_part + ((size_t)blockIdx.z * nrows + r0) * HEAD_PART_STRIDE(H3) + i;
    float v = 0.f;
#pragma unroll 8
    for (int r = r0; r < r1; ++r, src += HEAD_PART_STRIDE(H3)) v += *src;
    if (i < NW) {
        const int q = i / H3, cc = i % H3;
        if (q < MA) acc_add(P, &P.grads[(actor ? w_a : w_c) + (int64_t)q * H3 + cc], v);
        else acc_add(P, &P.grads[(actor ? b_prev_a : b_prev_c) + cc], v);
    } else {
        const int k = i - NW;
        if (actor) {
            if (k < MA) acc_add(P, &P.grads[P.off_std + k], v);
            else if (k < 2 * MA) acc_add(P, &P.grads[b_a + (k - MA)], v);
            else if (k == 2 * MA + 1) acc_add(P, &P.grads[P.num_params], v);
            else acc_add(P, &P.loss_acc[1], v);
        } else {
            if (k == 2 * MA) acc_add(P, &P.grads[b_c], v);
            else acc_add(P, &P.loss_acc[0], v);
        }
    }
}

// KL-adaptive learning rate (rsl_rl PPO.update) + reset of the norm accumulator
// Optimiser step in two launches.  k_opt_prepare: squared gradient norm (block partials -> atomics into
// loss_acc[2 + par]) and, on one lane, the KL-adaptive learning rate, the loss statistics and the Adam step
// count.  k_opt_adam: clip_grad_norm_(max_norm) + torch.optim.Adam step (betas 0.9/0.999, eps 1e-8), then
// zeroes what it consumed -- the gradient buffer (+ KL tail) for the next minibatch and the OTHER norm slot
// (par alternates per step, so no block can still be reading the slot that is cleared).
// Workgroups past `prep_blocks` gather the NEXT minibatch into the other buffer set (G = P with that set's pointers): the
// gather depends on the rollout storage and the permutation only, and this launch leaves most of the chip idle.
__global__ void __launch_bounds__(256) k_opt_prepare(PpoDev P, int par, int prep_blocks, PpoDev G, int gather_mb) {
    if ((int)blockIdx.x >= prep_blocks) { gather4_block(G, gather_mb, (int)blockIdx.x - prep_blocks); return; }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const float kl = P.grads[P.num_params] / ((float)P.mb_rows * (float)P.world);
        float lr = P.stats[0];
        if (P.adaptive) {
            if (kl > P.desired_kl * 2.0f) lr = fmaxf(1e-5f, lr / 1.5f);
            else if (kl < P.desired_kl / 2.0f && kl > 0.0f) lr = fminf(1e-2f, lr * 1.5f);
        }
        P.stats[0] = lr;
        P.stats[1] = kl;
        P.stats[2] += P.loss_acc[0] / (float)P.mb_rows;
        P.stats[3] += P.loss_acc[1] / (float)P.mb_rows;
        P.stats[4] += 1.0f;                                  // Adam step count t (k_opt_adam reads the new value)
        P.stats[5] += 1.0f;
        P.loss_acc[0] = 0.f; P.loss_acc[1] = 0.f;
    }
    float s = 0.f;
    const float inv_world = 1.0f / (float)P.world;
    for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < P.num_params; k += (int64_t)prep_blocks * blockDim.x) {
        float g = P.grads[k] * inv_world;
        s += g * g;
    }
    __shared__ float red[256];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) acc_add(P, &P.loss_acc[2 + par], red[0]);
}
// Deterministic mode: add the fixed-point shadow (PpoDev::det64) into the float buffers it stands for and clear it.  One adder per
// element and launch, so the float result does not depend on the order the contributions arrived in.
__global__ void __launch_bounds__(256) k_det_fold(PpoDev P) {
    const int64_t n = P.num_params + 10;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const long long q = P.det64[i];
        if (q == 0) continue;
        float *t = i < P.num_params + 2 ? P.grads + i : i < P.num_params + 6 ? P.loss_acc + (i - P.num_params - 2) : P.adv_partial + (i - P.num_params - 6);
        *t += (float)((double)q * (1.0 / LG_DET_SCALE));
        P.det64[i] = 0;
    }
}
// split-bf16 image of parameter k: pl_dest[k] = its element index inside a plane (weights) or -1 (biases, std)
__device__ __forceinline__ void write_planes(const PpoDev &P, int64_t k, float x) {
    const int d = P.pl_dest[k];
    if (d < 0) return;
    uint32_t h, m, l;
    split2(x, 0.f, h, m, l);
    P.wpl[d] = (uint16_t)h; P.wpl[P.pl_stride + d] = (uint16_t)m; P.wpl[2 * P.pl_stride + d] = (uint16_t)l;
}
// rebuild every plane from the fp32 parameters (begin of an update: parameters may have been written through
// the zero-copy views -- checkpoint load, initial broadcast)
__global__ void __launch_bounds__(256) k_sync_planes(PpoDev P) {
    for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < P.num_params; k += (int64_t)gridDim.x * blockDim.x)
        write_planes(P, k, P.params[k]);
}
__global__ void __launch_bounds__(256) k_opt_adam(PpoDev P, int par) {
    const float total = sqrtf(P.loss_acc[2 + par]);
    const float coef = fminf(P.max_grad_norm / (total + 1e-6f), 1.0f);
    const float lr = P.stats[0], t = P.stats[4];
    const float bc1 = 1.0f - powf(0.9f, t), bc2 = 1.0f - powf(0.999f, t);
    const float inv_world = 1.0f / (float)P.world;
    for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < P.num_params; k += (int64_t)gridDim.x * blockDim.x) {
        const float g = P.grads[k] * inv_world * coef;
        const float m = 0.9f * P.adam_m[k] + 0.1f * g;
        const float v = 0.999f * P.adam_v[k] + 0.001f * g * g;
        P.adam_m[k] = m;
        P.adam_v[k] = v;
        const float pn = P.params[k] - (lr / bc1) * m / (sqrtf(v) / sqrtf(bc2) + 1e-8f);
        P.params[k] = pn;
        P.grads[k] = 0.f;
        write_planes(P, k, pn);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        P.grads[P.num_params] = 0.f; P.grads[P.num_params + 1] = 0.f;
        P.loss_acc[2 + (par ^ 1)] = 0.f;
    }
}

extern "C" {
void ppok_act_sample(const PpoDev *P, const float *obs, const float *cobs, const float *mu, const float *val, int t,
                     int64_t cnt, int inject, hipStream_t s) {
    // one lane per env for the sampling; at least 256 blocks so the observation copy is spread over the chip
    const int blocks = (P->N + 63) / 64;
    hipLaunchKernelGGL(k_act_sample, dim3(blocks > 256 ? blocks : 256), dim3(64), 0, s, *P, obs, cobs, mu, val, t, cnt, inject);
}
void ppok_process_step(const PpoDev *P, const float *rew, const uint8_t *dones, const uint8_t *tos, int t, hipStream_t s) {
    hipLaunchKernelGGL(k_process_step, dim3((P->N + 255) / 256), dim3(256), 0, s, *P, rew, dones, tos, t);
}
void ppok_det_fold(const PpoDev *P, hipStream_t s) { hipLaunchKernelGGL(k_det_fold, dim3(512), dim3(256), 0, s, *P); }
void ppok_gae(const PpoDev *P, const float *last_values, hipStream_t s) {
    (void)hipMemsetAsync(P->adv_partial, 0, 3 * sizeof(float), s);
    hipLaunchKernelGGL(k_gae, dim3((P->N + 255) / 256), dim3(256), 0, s, *P, last_values);
    if (P->det64) ppok_det_fold(P, s);
}
void ppok_adv_normalize(const PpoDev *P, hipStream_t s) {
    hipLaunchKernelGGL(k_adv_normalize, dim3(256), dim3(256), 0, s, *P);
}
void ppok_randperm(const PpoDev *P, int n, uint64_t update_idx, hipStream_t s) {
    int bits = 2;
    while ((1ll << bits) < n) ++bits;
    bits += bits & 1;                                          // balanced halves
    hipLaunchKernelGGL(k_randperm, dim3((n + 255) / 256), dim3(256), 0, s, *P, n, bits / 2, update_idx);
}
void ppok_gather(const PpoDev *P, int mb, hipStream_t s) {
    if ((P->A & 3) == 0)
        hipLaunchKernelGGL(k_gather4, dim3((P->mb_rows * 32 + 255) / 256), dim3(256), 0, s, *P, mb);
    else
        hipLaunchKernelGGL(k_gather, dim3(P->mb_rows), dim3(64), 0, s, *P, mb);
}
size_t ppok_head_part_floats() { return (size_t)2 * HEAD_NET_GRID * HEAD_PART_STRIDE(128); }
// returns 0 when the fused head kernel supports this width, -1 otherwise (caller falls back to GEMMs + k_loss)
int ppok_head_fused(const PpoDev *P, int H3, const float *xa, const float *xc, float *dza, float *dzc, int64_t w_a, int64_t b_a,
                    int64_t w_c, int64_t b_c, int64_t b_prev_a, int64_t b_prev_c, hipStream_t s) {
    const int ntiles = (P->mb_rows + HEAD_ROWS - 1) / HEAD_ROWS;
    static const int head_grid = getenv("LG_HEAD_GRID") ? atoi(getenv("LG_HEAD_GRID")) : HEAD_GRID;
    dim3 grid(ntiles < head_grid ? ntiles : head_grid), block(256);
    static const int per_net = getenv("LG_HEAD_PER_NET") ? atoi(getenv("LG_HEAD_PER_NET")) : 1;
    if (H3 == 128 && per_net) {
        const int nrows = ntiles < HEAD_NET_GRID ? ntiles : HEAD_NET_GRID;
        if (P->A <= 12)
            hipLaunchKernelGGL((k_head_net<128, 12>), dim3(nrows, 2), block, 0, s, *P, xa, xc, dza, dzc, P->params + w_a, P->params + w_c, b_a, b_c);
        else
            hipLaunchKernelGGL((k_head_net<128, LG_PPO_MAX_A>), dim3(nrows, 2), block, 0, s, *P, xa, xc, dza, dzc, P->params + w_a, P->params + w_c, b_a, b_c);
        constexpr int NTOT = HEAD_PART_STRIDE(128);
        hipLaunchKernelGGL((k_head_finish<128>), dim3((NTOT + 255) / 256, HEAD_FIN_CHUNKS, 2), block, 0, s, *P, nrows, w_a, b_a, w_c, b_c,
                           b_prev_a, b_prev_c);
    } else if (H3 == 128) hipLaunchKernelGGL((k_head_fused<128>), grid, block, 0, s, *P, xa, xc, dza, dzc, w_a, b_a, w_c, b_c, b_prev_a, b_prev_c);
    else if (H3 == 64) hipLaunchKernelGGL((k_head_fused<64>), grid, block, 0, s, *P, xa, xc, dza, dzc, w_a, b_a, w_c, b_c, b_prev_a, b_prev_c);
    else if (H3 == 32) hipLaunchKernelGGL((k_head_fused<32>), grid, block, 0, s, *P, xa, xc, dza, dzc, w_a, b_a, w_c, b_c, b_prev_a, b_prev_c);
    else return -1;
    return 0;
}
void ppok_loss(const PpoDev *P, const float *mu, const float *v, float *dmu, float *dval, hipStream_t s) {
    hipLaunchKernelGGL(k_loss, dim3((P->mb_rows + 255) / 256), dim3(256), 0, s, *P, mu, v, dmu, dval);
}
void ppok_sync_planes(const PpoDev *P, hipStream_t s) { hipLaunchKernelGGL(k_sync_planes, dim3(256), dim3(256), 0, s, *P); }
// G / gather_mb: buffer set and index of a minibatch to gather beside the norm reduction (gather_mb < 0: none); returns
// whether the gather was taken (the 16-byte row layout of k_gather4)
int ppok_step(const PpoDev *P, int par, const PpoDev *G, int gather_mb, hipStream_t s) {
    const bool g4 = gather_mb >= 0 && (P->A & 3) == 0;
    const int gblocks = g4 ? (P->mb_rows * 32 + 255) / 256 : 0;
    hipLaunchKernelGGL(k_opt_prepare, dim3(128 + gblocks), dim3(256), 0, s, *P, par, 128, g4 ? *G : *P, gather_mb);
    if (P->det64) ppok_det_fold(P, s);                   // the squared gradient norm
    // one parameter per thread: the per-parameter chain (4 loads, Adam, 4 stores + the three plane stores through pl_dest) is a
    // memory round trip that a grid-stride loop repeats serially (6 x for [512,256,128] on 256 workgroups: 12.2 us; 8.9 us on 1024, 10.2 on 2048)
    static const int adam_max = getenv("LG_ADAM_WGS") ? atoi(getenv("LG_ADAM_WGS")) : 1024;
    const long want = (P->num_params + 255) / 256;
    hipLaunchKernelGGL(k_opt_adam, dim3((unsigned)(want < adam_max ? (want > 0 ? want : 1) : adam_max)), dim3(256), 0, s, *P, par);
    return g4 ? 1 : 0;
}
}

// Debug / microbenchmark entry (tools/gemm_bench.py): C[M,N] = A . B with the layouts of `mode`
// (0: A[m][k] B[n][k] forward; 1: A[m][k] B[k][n] input-gradient; 2: A[k][m] B[k][n] weight-gradient, C zeroed by caller).
extern "C" void ppok_debug_gemm(const float *A, const float *B, float *C, int M, int N, int K, int mode, int splits, void *stream) {
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A[0] = A; g.B[0] = B; g.C[0] = C; g.M[0] = M; g.N[0] = N; g.K[0] = K; g.ldc[0] = N;
    if (mode == 0) { g.lda[0] = K; g.ldb[0] = K; launch_gemm<true, true, 0>(g, 1, 1, (hipStream_t)stream); }
    else if (mode == 1) { g.lda[0] = K; g.ldb[0] = N; g.aux[0] = C; g.ldaux[0] = N; g.elu = 1; launch_gemm<true, false, 1>(g, 1, 1, (hipStream_t)stream); }
    else { g.lda[0] = M; g.ldb[0] = N; ppok_gemm_dw(&g, 1, splits, (hipStream_t)stream); }
}

// Debug entry for the weight-plane operand paths: W [rows][cols] fp32 is split into its three bf16 planes (caller-provided
// scratch of 3 x plane_stride uint16, plane_stride = rows * cols rounded up to 8), then
//   mode 0: C[M][rows] = A[M][cols] . W^T   (forward: planes as the reduction-contiguous operand)
//   mode 1: C[M][cols] = A[M][rows] . W     (input gradient: the same planes through the transposing LDS read)
__global__ void __launch_bounds__(256) k_debug_split(const float *__restrict__ W, uint16_t *__restrict__ pl, int64_t n, int64_t stride) {
    for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) {
        uint32_t h, m, l;
        split2(W[k], 0.f, h, m, l);
        pl[k] = (uint16_t)h; pl[stride + k] = (uint16_t)m; pl[2 * stride + k] = (uint16_t)l;
    }
}
extern "C" int ppok_debug_gemm_planes(const float *A, const float *W, float *C, uint16_t *planes, int64_t plane_stride, int M, int rows,
                                      int cols, int mode, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_debug_split, dim3(256), dim3(256), 0, s, W, planes, (int64_t)rows * cols, plane_stride);
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A[0] = A; g.B[0] = W; g.C[0] = C; g.M[0] = M; g.Bpl[0] = planes; g.pl_stride = plane_stride; g.ldb[0] = cols;
    if (mode == 0) {
        g.N[0] = rows; g.K[0] = cols; g.lda[0] = cols; g.ldc[0] = rows;
        if (!planes_ok(g, 1)) return -1;
        launch_gemm_pl<0>(g, 1, s);
    } else {
        g.N[0] = cols; g.K[0] = rows; g.lda[0] = rows; g.ldc[0] = cols; g.aux[0] = C; g.ldaux[0] = cols; g.elu = 0;
        if (!planes_t_ok(g, 1)) return -1;
        launch_gemm_pl<1, false, 2>(g, 1, s);
    }
    return 0;
}
