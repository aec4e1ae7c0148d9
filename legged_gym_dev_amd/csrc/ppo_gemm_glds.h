// Forward and input-gradient GEMMs of the update on the LDS-DMA path (round 4; the measured prototype is tools/micro/gemm_glds.hip, its table
// profiles/r04_gemm_glds_proto.txt).  Same arithmetic as k_gemm<.., B_PL = 1>: C = act(A . W^T + b) with every fp32 product summed from six
// bf16 MFMAs, the weight operand from the optimiser's pre-split planes -- bit-identical results (same split, same order of the six
// products, same k order) -- but the operands reach LDS by global_load_lds_dwordx4: no VGPR staging, no ds_write, no second barrier.
//   * A stays fp32 in LDS ([row][32 k] = 128-byte rows, 4 B per element against the 6 B of three planes) and is split into its three bf16
//     terms when a wave reads its fragment.  4 waves stacked in M, each 32 rows x 128 columns: a row block is split by one wave only.
//   * Both images are lane-linear (the DMA writes base + 16 lane).  The ds_read_b128 fragment reads are kept conflict-free by permuting
//     the 16-byte chunks of a row on the SOURCE side and applying the same XOR on the read (cdna_hip_programming.md rule 21):
//     A: chunk c of row r at position c ^ ((r >> 1) & 7); W: chunk c of row n at position c ^ ((n >> 2) & 3).
//   * ONE LDS stage of 40 KB: up to four workgroups per CU.  The schedule inside a workgroup (DMA -> wait -> multiply) overlaps nothing;
//     the other workgroups of the CU fill in.  Measured, deeper pipelines inside the workgroup buy nothing on this chip (the kernel
//     is bound by what the chip lets it draw, not by issue slots: +-6 % over 25 structures), while more resident workgroups remove
//     the half-empty second round of 768-tile launches: 83 -> 75 us for the 512 -> 256 layer, 30.7 -> 24-26 us for 256 -> 128.
//   * The DMA is issued from inline asm.  hipcc's s_waitcnt insertion orders every LDS read behind every LDS-DMA it knows to be in
//     flight; with one stage that is what the loop wants anyway, but the asm form also keeps M0 / the saddr addressing in our hands.
// Input gradient (EPI 1, B_PL 2: dX = dz . W reduced over the planes' ROWS): the k-tile of the planes is brought as it lies in memory,
// [32 k][128 n] bf16 per plane, into the subtile image of the transposing LDS read (plt_off with 512-byte subtiles: the 64-byte pad of
// the register-staged kernel only served its ds_write banks) -- the DMA picks, per lane, the 16-byte chunk that belongs at its LDS
// position -- and the fragments come out of ds_read_b64_tr_b16 exactly as in k_gemm<.., B_PL = 2>.
// Requirements (checked by the launcher, glds_ok): K a multiple of 32 (the first layer's rows are padded to that, PpoDev::Op), N a
// multiple of 128, 16-byte aligned rows.  Any M: rows past M are clamped on the way in and not stored.
#pragma once

__device__ __forceinline__ void glds16(const void *sbase, unsigned voff, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_addr) : "memory", "m0");
}

__device__ __forceinline__ void glds_split8(const float4 &lo, const float4 &hi, bf16x8 &h, bf16x8 &m, bf16x8 &l) {
    uint32_t hh[4], mm[4], ll[4];
    split2(lo.x, lo.y, hh[0], mm[0], ll[0]);
    split2(lo.z, lo.w, hh[1], mm[1], ll[1]);
    split2(hi.x, hi.y, hh[2], mm[2], ll[2]);
    split2(hi.z, hi.w, hh[3], mm[3], ll[3]);
    h = __builtin_bit_cast(bf16x8, make_uint4(hh[0], hh[1], hh[2], hh[3]));
    m = __builtin_bit_cast(bf16x8, make_uint4(mm[0], mm[1], mm[2], mm[3]));
    l = __builtin_bit_cast(bf16x8, make_uint4(ll[0], ll[1], ll[2], ll[3]));
}

#define GLDS_BM 128
#define GLDS_BN 128
#define GLDS_A_BYTES (GLDS_BM * 128)
#define GLDS_PL (GLDS_BN * 64)
#define GLDS_STAGE (GLDS_A_BYTES + 3 * GLDS_PL)

#define GLDS_SUB 512                 // subtile of the transposed-read image: 8 k-rows x 32 n-columns of bf16, no pad
// register budget: four waves per SIMD (128 registers) for the forward; the input gradient's epilogue (activation derivative from aux, column
// sums) needs a few more: three per SIMD (168), i.e. three 40 KB workgroups per CU -- still one round for the update's 768-tile launches
template <int EPI, int B_PL>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(EPI == 0 ? 4 : 3, EPI == 0 ? 4 : 3))) k_gemm_glds(GemmArgs g) {
    static_assert((EPI == 0 && B_PL == 1) || (EPI == 1 && B_PL == 2), "forward on [n][k] planes, input gradient on the same planes read along their rows");
#ifdef LG_EXP_GEMM_CLOCK             // diagnostic build only (make exp): in-kernel clock of the workgroups, s_memtime / s_memrealtime (as in k_gemm)
    const unsigned long long clk_t0 = __builtin_amdgcn_s_memtime(), clk_r0 = __builtin_amdgcn_s_memrealtime();
    struct ClkEnd { unsigned long long t0, r0; __device__ ~ClkEnd() {
        const unsigned b = blockIdx.x + gridDim.x * blockIdx.z;
        if (threadIdx.x == 0 && b < 2048) { g_gemm_clk[2 * b] = __builtin_amdgcn_s_memtime() - t0; g_gemm_clk[2 * b + 1] = __builtin_amdgcn_s_memrealtime() - r0; } } } clk_end{clk_t0, clk_r0};
#endif
    constexpr int NW = 4, GA = GLDS_BM / 8 / NW, GB = 3 * GLDS_BN / 16 / NW;       // DMA instructions per wave and k-tile: 4 + 6
    __shared__ __attribute__((aligned(1024))) unsigned char lds[GLDS_STAGE];
    const int z = blockIdx.z;
    const int M = g.M[z], N = g.N[z], K = (B_PL == 1 && g.Kpl[z]) ? g.Kpl[z] : g.K[z];
    const int tiles_n = N / GLDS_BN, tiles_m = (M + GLDS_BM - 1) / GLDS_BM;
    if ((int)blockIdx.x >= tiles_n * tiles_m) return;
    int tm, tn;
    xcd_tile((int)blockIdx.x, tiles_n, tiles_m, tm, tn);            // the column tiles of a row block share the A rows: one XCD
    const int m0 = tm * GLDS_BM, n0 = tn * GLDS_BN;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lk = lane >> 5;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int lda = g.lda[z], ldb = (B_PL == 1 && g.ldbpl[z]) ? g.ldbpl[z] : g.ldb[z];
    const unsigned char *__restrict__ A = reinterpret_cast<const unsigned char *>(g.A[z]);
    const unsigned char *__restrict__ W = reinterpret_cast<const unsigned char *>(g.Bpl[z]);
    const int nt = K / BK;

    // DMA source offsets of this lane (bytes from the k-tile's first column / first k-row)
    unsigned a_src[GA], b_src[GB];
#pragma unroll
    for (int i = 0; i < GA; ++i) {
        const int ins = wave + i * NW, row = 8 * ins + (lane >> 3), cpos = lane & 7;
        a_src[i] = (unsigned)((size_t)min(m0 + row, M - 1) * lda * 4 + 16 * (cpos ^ ((row >> 1) & 7)));
    }
#pragma unroll
    for (int i = 0; i < GB; ++i) {
        const int ins = wave + i * NW, p = ins / (GLDS_BN / 16), q = ins % (GLDS_BN / 16);
        if constexpr (B_PL == 1) {
            const int row = 16 * q + (lane >> 2), cpos = lane & 3;
            b_src[i] = (unsigned)((p * g.pl_stride + (int64_t)(n0 + row) * ldb) * 2 + 16 * (cpos ^ ((row >> 2) & 3)));
        } else {
            // 1 KB piece q of the plane image: k-rows 8 (q >> 1) .. + 7, subtiles 2 (q & 1), 2 (q & 1) + 1; lane: subtile, row, position
            const int sub = 2 * (q & 1) + (lane >> 5), krow = 8 * (q >> 1) + ((lane >> 2) & 7), pos = lane & 3;
            const int ch = 4 * sub + (pos ^ ((krow >> 2) & 3));                  // the 16-byte chunk (8 n-columns) that belongs there
            b_src[i] = (unsigned)((p * g.pl_stride + (int64_t)krow * ldb + n0 + 8 * ch) * 2);
        }
    }
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)lds;
    const size_t b_step = B_PL == 1 ? (size_t)BK * 2 : (size_t)BK * ldb * 2;       // bytes from one k-tile of the planes to the next

    // fragment addresses
    const int r = 32 * wave + li;
    const int fa = r * 128 + 16 * ((2 * lk) ^ ((r >> 1) & 7));
    int fb[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) { const int n = 32 * b + li; fb[b] = GLDS_A_BYTES + n * 64 + 16 * (lk ^ ((n >> 2) & 3)); }
    // transposed-read bases (B_PL 2; cdna_hip_programming.md T10): group tg = lane >> 4 takes n-columns 16 (tg & 1) .. + 15 and k-rows
    // 8 (tg >> 1) + 4 h .. + 3 of a k-step; lane 4 q + p of the group addresses row q, half-chunk p
    const int tg = (tid >> 4) & 3, tq = (tid >> 2) & 3, tp = tid & 3;
    int ft[2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
        ft[h] = GLDS_A_BYTES + (GLDS_BN / 32 * GLDS_SUB) * (tg >> 1) + 64 * (4 * h + tq) + 16 * ((2 * (tg & 1) + (tp >> 1)) ^ (2 * (tg >> 1) + h)) + 8 * (tp & 1);

    f32x16 acc[1][4];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[0][b][q] = 0.f;

    for (int t = 0; t < nt; ++t) {
        if (t) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");         // every wave has read tile t - 1
        {
            const unsigned char *ab = A + (size_t)t * BK * 4, *wb = W + (size_t)t * b_step;
#pragma unroll
            for (int i = 0; i < GA; ++i) glds16(ab, a_src[i], lds0 + (wave_u + i * NW) * 1024);
#pragma unroll
            for (int i = 0; i < GB; ++i) {
                const int ins = wave_u + i * NW;
                glds16(wb, b_src[i], lds0 + GLDS_A_BYTES + (ins / (GLDS_BN / 16)) * GLDS_PL + (ins % (GLDS_BN / 16)) * 1024);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");                  // tile t has landed, for every wave
        float4 ar[2][2];
        bf16x8 av[3], bv[2][4][3];
        auto read = [&](int s, int q) __attribute__((always_inline)) {
            const int o = fa ^ (64 * s);
            ar[q][0] = *reinterpret_cast<const float4 *>(lds + o);
            ar[q][1] = *reinterpret_cast<const float4 *>(lds + (o ^ 16));
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    if constexpr (B_PL == 1) bv[q][b][p] = *reinterpret_cast<const bf16x8 *>(lds + (fb[b] ^ (32 * s)) + p * GLDS_PL);
                    else {
                        const int o2 = GLDS_SUB * b + p * GLDS_PL + (GLDS_BN / 32 * GLDS_SUB) * 2 * s;
                        const s16x4 lo = lds_read_tr16(lds + ft[0] + o2), hi = lds_read_tr16(lds + ft[1] + o2);
                        typedef short s16x8 __attribute__((ext_vector_type(8)));
                        const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                        bv[q][b][p] = __builtin_bit_cast(bf16x8, v);
                    }
                }
        };
        read(0, 0);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (s + 1 < 2) read(s + 1, (s + 1) & 1);
            glds_split8(ar[s & 1][0], ar[s & 1][1], av[0], av[1], av[2]);
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const bf16x8 *x = av, *y = bv[s & 1][b];
                f32x16 c = acc[0][b];                 // smallest terms first (as gemm_mainloop_x6)
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[1], y[1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[0], y[2], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[2], y[0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[0], y[1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[1], y[0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[0], y[0], c, 0, 0, 0);
                acc[0][b] = c;
            }
        }
    }
    // epilogue of k_gemm: wave offsets (32 wave, 0), TM = 1, TN = 4
    if (g.elu == 1) gemm_epilogue<EPI, 1, 4, GLDS_BM, GLDS_BN, 1>(g, z, M, N, m0, n0, 32 * wave, 0, li, lk, g.ldc[z], acc);
    else if (g.elu == 0) gemm_epilogue<EPI, 1, 4, GLDS_BM, GLDS_BN, 0>(g, z, M, N, m0, n0, 32 * wave, 0, li, lk, g.ldc[z], acc);
    else gemm_epilogue<EPI, 1, 4, GLDS_BM, GLDS_BN, -1>(g, z, M, N, m0, n0, 32 * wave, 0, li, lk, g.ldc[z], acc);
}

// B_PL 1: the plane path's view of the problem (K = Kpl, ldb = ldbpl already substituted by the caller)
template <int B_PL>
static bool glds_ok(const GemmArgs &g, int nz) {
    for (int z = 0; z < nz; ++z) {
        const int K = (B_PL == 1 && g.Kpl[z]) ? g.Kpl[z] : g.K[z], ldb = (B_PL == 1 && g.ldbpl[z]) ? g.ldbpl[z] : g.ldb[z];
        if (!g.Bpl[z] || K < BK || (K % BK) || (g.N[z] % GLDS_BN) || g.M[z] < 1 || (g.lda[z] & 3) || ((uintptr_t)g.A[z] & 15) ||
            (ldb & 7) || ((uintptr_t)g.Bpl[z] & 15) || (g.pl_stride & 7))
            return false;
        const size_t b_span = B_PL == 1 ? (size_t)g.N[z] * ldb : (size_t)BK * ldb + g.N[z];       // 32-bit lane offsets
        if ((size_t)g.M[z] * g.lda[z] * 4 >= (1ull << 32) || ((size_t)2 * g.pl_stride + b_span) * 2 >= (1ull << 32)) return false;
    }
    return true;
}
