// Measured-and-rejected GEMM structures of rounds 2-3, kept buildable for A/B only: compiled into liblegged_hip_exp.so by
// `make -C legged_gym_dev_amd/csrc exp EXPFLAGS=-DLG_EXP_KERNELS` (never into the product library).  Results: DESIGN.md section 4's table,
// profiles/r03_ab.txt.  Included by ppo_kernels.hip after gemm_epilogue when LG_EXP_KERNELS is defined.
//   LG_GEMM_PP   ping-pong kernel (two 4-wave groups out of step inside one workgroup): 1.3x slower
//   LG_GEMM_W4 / LG_DW_W4   128 x 128 tile on 4 waves of 64 x 64: neutral (one instantiation spills 20 B)
//   LG_GEMM_LDB  two LDS stages for the 64 x 64 configuration: neutral
//   ppok_debug_set_t96   96 x 128 tiles: slower end to end
#pragma once

// ------------------------------------------------------------------------------------------------
// Ping-pong mainloop (k_gemm_pp).  In gemm_mainloop_x6 a workgroup alternates between an operand-delivery phase (wait for the
// global loads, split to bf16 planes, write LDS) and an MFMA phase, and measured per launch the two ADD UP (forward 24576x512->256:
// ~18 us of delivery + ~20 us of MFMA per net): two resident workgroups do not reliably fall out of step.  Here one workgroup of 8
// waves holds TWO groups of 4 waves (one per SIMD each), each with its own 128 x 128 tile (64 x 64 per wave) and its own LDS stage,
// and the workgroup-wide barrier forces them out of step: while group 0 multiplies k-tile t, group 1 splits and stores its k-tile t;
// after the barrier they swap.  The matrix pipe of a SIMD always has exactly one wave feeding it; the other wave of that SIMD is in its
// delivery phase (VALU / LDS writes / global-load issue), which runs in the MFMAs' shadow.
//   phase A(t): group 0 MFMA(t)      | group 1 STORE(t)
//   phase B(t): group 0 STORE(t + 1) | group 1 MFMA(t)
// Loads run two k-tiles ahead in two register sets (set t & 1 holds tile t; refilled with tile t + 2 right after it is stored).
template <bool B_RC, bool VEC, bool FULL, int B_PL>
__device__ __forceinline__ void gemm_mainloop_x6_pp(const float *__restrict__ A, const float *__restrict__ B, int lda, int ldb, int m0, int n0,
                                                    int M, int N, int k_begin, int k_end, unsigned char *__restrict__ lds, int grp, int wm, int wn,
                                                    int li, int lk, f32x16 (&acc)[2][2], const uint16_t *__restrict__ Bpl, int64_t pl_stride) {
    constexpr int TM = 2, TN = 2, BM = 128, BN = 128, NT = 256;
    constexpr int APL = x6_plane_bytes<BM>(), BPL = B_PL == 2 ? plt_plane_bytes<BN>() : x6_plane_bytes<BN>();
    constexpr int NVA = BM * BK / 4 / NT, NVP = BN * 12 / NT;
    struct Regs { float4 a[NVA]; uint4 p[NVP]; unsigned ma = 0, mb = 0; };
    Regs R0, R1;
    unsigned char *lds_b = lds + 3 * APL;
    auto load_tile = [&](Regs &r, int k) __attribute__((always_inline)) {
        stage_load<true, BM, VEC, true, NT, FULL>(A, lda, m0, k, M, k_end, r.a, r.ma);
        if constexpr (B_PL == 2) stage_load_plt<BN, NT>(Bpl, pl_stride, ldb, n0, k, N, k_end, r.p, r.mb);
        else stage_load_pl<BN, NT>(Bpl, pl_stride, ldb, n0, k, N, k_end, r.p, r.mb);
    };
    auto store_tile = [&](Regs &r) __attribute__((always_inline)) {
        stage_store_x6<true, BM, NT, FULL>(lds, r.a, r.ma);
        if constexpr (B_PL == 2) stage_store_plt<BN, NT>(lds_b, r.p, r.mb);
        else stage_store_pl<BN, NT>(lds_b, r.p, r.mb);
    };
    const unsigned char *fa = lds + x6_prow<BM>(wm + li) * X6_ROWB + 16 * lk;
    const unsigned char *fb = lds_b + x6_prow<BN>(wn + li) * X6_ROWB + 16 * lk;
    const int tg = (threadIdx.x >> 4) & 3, tq = (threadIdx.x >> 2) & 3, tp = threadIdx.x & 3;
    const unsigned char *ft[2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
        ft[h] = lds_b + (BN / 32 * LG_PLT_SUB) * (tg >> 1) + LG_PLT_SUB * (wn / 32) + 64 * (4 * h + tq) +
                16 * ((2 * (tg & 1) + (tp >> 1)) ^ (2 * (tg >> 1) + h)) + 8 * (tp & 1);
    auto read_b = [&](int b, int p, int s) __attribute__((always_inline)) -> bf16x8 {
        if constexpr (B_PL == 2) {
            const int o = LG_PLT_SUB * b + p * BPL + (BN / 32 * LG_PLT_SUB) * 2 * s;
            const s16x4 lo = lds_read_tr16(ft[0] + o), hi = lds_read_tr16(ft[1] + o);
            typedef short s16x8 __attribute__((ext_vector_type(8)));
            const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            return __builtin_bit_cast(bf16x8, v);
        } else {
            return *reinterpret_cast<const bf16x8 *>(fb + b * 8 * X6_ROWB + p * BPL + s * 32);
        }
    };
    auto mfma_tile = [&]() __attribute__((always_inline)) {
        bf16x8 av[2][TM][3], bv[2][TN][3];
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int p = 0; p < 3; ++p) av[0][a][p] = *reinterpret_cast<const bf16x8 *>(fa + a * 8 * X6_ROWB + p * APL);
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int p = 0; p < 3; ++p) bv[0][b][p] = read_b(b, p, 0);
#pragma unroll
        for (int s = 0; s < BK / 16; ++s) {
            if (s + 1 < BK / 16) {
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int p = 0; p < 3; ++p)
                        av[(s + 1) & 1][a][p] = *reinterpret_cast<const bf16x8 *>(fa + a * 8 * X6_ROWB + p * APL + (s + 1) * 32);
#pragma unroll
                for (int b = 0; b < TN; ++b)
#pragma unroll
                    for (int p = 0; p < 3; ++p) bv[(s + 1) & 1][b][p] = read_b(b, p, s + 1);
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
    };
    load_tile(R0, k_begin);
    if (k_begin + BK < k_end) load_tile(R1, k_begin + BK);
    if (grp == 0) {                                          // phase 0: group 0 stores its first tile, group 1 only loads
        store_tile(R0);
        if (k_begin + 2 * BK < k_end) load_tile(R0, k_begin + 2 * BK);
    }
    lds_barrier();
    // one k-tile per iteration, two phases; the register set of tile t is R0 for even t
    auto step = [&](int k0, Regs &cur, Regs &nxt) __attribute__((always_inline)) {
        // phase A
        if (grp == 0) mfma_tile();
        else {
            store_tile(cur);
            if (k0 + 2 * BK < k_end) load_tile(cur, k0 + 2 * BK);
        }
        lds_barrier();
        // phase B
        if (grp == 0) {
            if (k0 + BK < k_end) {
                store_tile(nxt);
                if (k0 + 3 * BK < k_end) load_tile(nxt, k0 + 3 * BK);
            }
        } else mfma_tile();
        lds_barrier();
    };
    for (int k0 = k_begin; k0 < k_end; k0 += 2 * BK) {
        step(k0, R0, R1);
        if (k0 + BK < k_end) step(k0 + BK, R1, R0);
    }
}


// Forward (EPI 0) / input gradient (EPI 1) on the weight planes with the ping-pong mainloop: 8 waves = two groups of four, workgroup
// tile 256 x 128 (group g: rows m0 + 128 g ...), one workgroup per CU.
template <bool B_RC, int EPI, int B_PL>
__global__ void __launch_bounds__(512, 1) k_gemm_pp(GemmArgs g) {
    static_assert((B_PL == 1) == B_RC && (B_PL == 1 || B_PL == 2), "weight planes: [n][k] reduction-contiguous (1) or read along their rows (2)");
    constexpr int BM = 128, BN = 128;
    const int z = blockIdx.z;
    const int M = g.M[z], N = g.N[z], K = g.K[z];
    const int tiles_n = (N + BN - 1) / BN, tiles_m = (M + 2 * BM - 1) / (2 * BM);
    if ((int)blockIdx.x >= tiles_n * tiles_m) return;
    int tm, tn;
    xcd_tile((int)blockIdx.x, tiles_n, tiles_m, tm, tn);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, grp = wave >> 2, lw = wave & 3;
    const int m0 = tm * 2 * BM + grp * BM, n0 = tn * BN;
    const float *__restrict__ A = g.A[z];
    const int lda = g.lda[z], ldb = g.ldb[z], ldc = g.ldc[z];
    const bool a_vec = (lda & 3) == 0 && ((uintptr_t)A & 15) == 0 && ((K & 3) == 0) && K >= 4;
    constexpr int STAGE = 3 * (x6_plane_bytes<BM>() + (B_PL == 2 ? plt_plane_bytes<BN>() : x6_plane_bytes<BN>()));
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[2 * STAGE];
    const int wm = (lw / 2) * 64, wn = (lw % 2) * 64, li = lane & 31, lk = lane >> 5;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    // (a group whose rows lie past M still runs the loop -- clamped loads, masked to zero -- for the barriers; it stores nothing)
    const bool fullp = a_vec && m0 + BM <= M && n0 + BN <= N && (K % BK) == 0;
    unsigned char *lds = lds_raw + grp * STAGE;
    if (fullp) gemm_mainloop_x6_pp<B_RC, true, true, B_PL>(A, g.B[z], lda, ldb, m0, n0, M, N, 0, K, lds, grp, wm, wn, li, lk, acc, g.Bpl[z], g.pl_stride);
    else if (a_vec) gemm_mainloop_x6_pp<B_RC, true, false, B_PL>(A, g.B[z], lda, ldb, m0, n0, M, N, 0, K, lds, grp, wm, wn, li, lk, acc, g.Bpl[z], g.pl_stride);
    else gemm_mainloop_x6_pp<B_RC, false, false, B_PL>(A, g.B[z], lda, ldb, m0, n0, M, N, 0, K, lds, grp, wm, wn, li, lk, acc, g.Bpl[z], g.pl_stride);
    if (m0 >= M) return;
    if (g.elu == 1) gemm_epilogue<EPI, 2, 2, BM, BN, 1>(g, z, M, N, m0, n0, wm, wn, li, lk, ldc, acc);
    else if (g.elu == 0) gemm_epilogue<EPI, 2, 2, BM, BN, 0>(g, z, M, N, m0, n0, wm, wn, li, lk, ldc, acc);
    else gemm_epilogue<EPI, 2, 2, BM, BN, -1>(g, z, M, N, m0, n0, wm, wn, li, lk, ldc, acc);
}

