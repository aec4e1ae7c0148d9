// Feasibility probe (round 3): forward GEMM C = ELU(A . W^T + b) with the ACTIVATION operand read straight from row-major fp32 HBM
// into MFMA fragments (v_mfma_f32_16x16x32_bf16: a lane's fragment is 8 consecutive k of one row = 32 B, four lanes cover a row's
// 128-byte line) and split into its three bf16 terms in registers; only the pre-split weight tile goes through LDS (double-buffered,
// one barrier per k-tile).  Workgroup = 4 waves stacked in M, wave tile 64 x 64, workgroup tile 256 x 64.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/gemm_adirect.hip -o tools/micro/_bin/gemm_adirect && tools/micro/_bin/gemm_adirect
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t cvt_pk_bf16(float a, float b) {
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ void split8(const float (&x)[8], bf16x8 &h, bf16x8 &m, bf16x8 &l) {
    uint32_t hh[4], mm[4], ll[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float x0 = x[2 * i], x1 = x[2 * i + 1];
        hh[i] = cvt_pk_bf16(x0, x1);
        f32x2 r = f32x2{x0, x1} - f32x2{__uint_as_float(hh[i] << 16), __uint_as_float(hh[i] & 0xffff0000u)};
        mm[i] = cvt_pk_bf16(r.x, r.y);
        r -= f32x2{__uint_as_float(mm[i] << 16), __uint_as_float(mm[i] & 0xffff0000u)};
        ll[i] = cvt_pk_bf16(r.x, r.y);
    }
    h = __builtin_bit_cast(bf16x8, make_uint4(hh[0], hh[1], hh[2], hh[3]));
    m = __builtin_bit_cast(bf16x8, make_uint4(mm[0], mm[1], mm[2], mm[3]));
    l = __builtin_bit_cast(bf16x8, make_uint4(ll[0], ll[1], ll[2], ll[3]));
}

#define BK 32
#define ROWB 80                      // LDS row: 32 bf16 + 16 B pad
// TMF = A fragments (16 rows each) per wave; workgroup = 4 waves stacked in M = 64 TMF rows x BN columns.  Loads run TWO k-tiles
// ahead (three register sets for the fp32 A fragments, two for the weight chunks): one k-tile of MFMAs is shorter than an L2 / HBM
// round trip under load.
// FLAGS (timing experiments, wrong results): 1 = no A reload from HBM, 4 = no weight staging, 8 = MFMAs dropped to one per block
template <int BN, int TMF, int FLAGS = 0>
__global__ void __launch_bounds__(256, TMF == 2 ? 3 : 2) k_fwd(const float *__restrict__ A, int lda, const uint16_t *__restrict__ Wpl, int64_t pl_stride,
                                                const float *__restrict__ bias, float *__restrict__ C, int ldc, int M, int N, int K) {
    constexpr int NB = BN / 16, PL = BN * ROWB, STAGE = 3 * PL, NCH = 3 * BN * 4 / 256, WR = 16 * TMF;
    static_assert(BN == 64, "weight staging map: 256 threads x 3 planes");
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * STAGE];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, lg = lane >> 4;
    const int tiles_n = N / BN;
    const int xc = blockIdx.x & 7, xj = blockIdx.x >> 3;
    const int m0 = (xc + 8 * (xj / tiles_n)) * (4 * WR) + wave * WR, n0 = (xj % tiles_n) * BN;
    f32x4 acc[TMF][NB];
#pragma unroll
    for (int a = 0; a < TMF; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float *ap[TMF];
#pragma unroll
    for (int a = 0; a < TMF; ++a) ap[a] = A + (size_t)(m0 + 16 * a + lr) * lda + 8 * lg;
    float4 raw0[TMF][2], raw1[TMF][2], raw2[TMF][2];
    static_assert(NCH == 3, "three 16-byte weight chunks per thread");
    struct W3 { uint4 a, b, c; } wA, wB;
    auto load_a = [&](int k0, float4 (&r)[TMF][2]) {
#pragma unroll
        for (int a = 0; a < TMF; ++a) {
            r[a][0] = *reinterpret_cast<const float4 *>(ap[a] + k0);
            r[a][1] = *reinterpret_cast<const float4 *>(ap[a] + k0 + 4);
        }
    };
    // chunk v of this thread: plane v (BN * 4 = 256 chunks per plane), row tid >> 2, quarter tid & 3
    const uint16_t *wsrc = Wpl + (size_t)(n0 + (tid >> 2)) * K + 8 * (tid & 3);
    unsigned char *wdst = lds + (tid >> 2) * ROWB + 16 * (tid & 3);
    auto load_w = [&](int k0, W3 &w) __attribute__((always_inline)) {
        w.a = *reinterpret_cast<const uint4 *>(wsrc + k0);
        w.b = *reinterpret_cast<const uint4 *>(wsrc + pl_stride + k0);
        w.c = *reinterpret_cast<const uint4 *>(wsrc + 2 * pl_stride + k0);
    };
    auto store_w = [&](int st, const W3 &w) __attribute__((always_inline)) {
        *reinterpret_cast<uint4 *>(wdst + st * STAGE) = w.a;
        *reinterpret_cast<uint4 *>(wdst + st * STAGE + PL) = w.b;
        *reinterpret_cast<uint4 *>(wdst + st * STAGE + 2 * PL) = w.c;
    };
    load_a(0, raw0);
    load_w(0, wA);
    if (BK < K) { load_a(BK, raw1); load_w(BK, wB); }
    store_w(0, wA);
    if (2 * BK < K) load_w(2 * BK, wA);
    __syncthreads();
    const unsigned char *fb = lds + lr * ROWB + 16 * lg;
    // k-tile at k0: cur = its A fragments (landed), far = the set tile k0 + 2 BK is loaded into, wn = chunks of tile k0 + BK (landed,
    // stored to the other LDS stage after the MFMAs), wf = the set tile k0 + 2 BK's chunks are loaded into (= the set just stored...
    // so the load is issued after the store)
    auto tile = [&](int k0, float4 (&cur)[TMF][2], float4 (&far)[TMF][2], W3 &wn, int st) __attribute__((always_inline)) {
        if (k0 + 2 * BK < K && !(FLAGS & 1)) load_a(k0 + 2 * BK, far);
        const unsigned char *f = fb + st * STAGE;
        bf16x8 bh[NB], bm[NB], bl[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const unsigned char *g = f + b * 16 * ROWB;
            bh[b] = *reinterpret_cast<const bf16x8 *>(g); bm[b] = *reinterpret_cast<const bf16x8 *>(g + PL); bl[b] = *reinterpret_cast<const bf16x8 *>(g + 2 * PL);
        }
        bf16x8 ah, am, al, nh, nm, nl;
        {
            const float x[8] = {cur[0][0].x, cur[0][0].y, cur[0][0].z, cur[0][0].w, cur[0][1].x, cur[0][1].y, cur[0][1].z, cur[0][1].w};
            split8(x, ah, am, al);
        }
#pragma unroll
        for (int a = 0; a < TMF; ++a) {
            if (a + 1 < TMF) {
                const float x[8] = {cur[a + 1][0].x, cur[a + 1][0].y, cur[a + 1][0].z, cur[a + 1][0].w,
                                    cur[a + 1][1].x, cur[a + 1][1].y, cur[a + 1][1].z, cur[a + 1][1].w};
                split8(x, nh, nm, nl);
            }
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                f32x4 c = acc[a][b];                     // smallest terms first
                if (!(FLAGS & 8)) {
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm[b], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[b], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[b], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm[b], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh[b], c, 0, 0, 0);
                }
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[b], c, 0, 0, 0);
                acc[a][b] = c;
            }
            if (a + 1 < TMF) { ah = nh; am = nm; al = nl; }
        }
        if (k0 + BK < K && !(FLAGS & 4)) {
            store_w(st ^ 1, wn);                         // the other stage: nobody reads it during this k-tile
            if (k0 + 3 * BK < K) load_w(k0 + 3 * BK, wn);     // ... and the freed set takes the chunks of tile k0 + 3 BK
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // LDS-only barrier: __syncthreads() would drain the prefetches (vmcnt(0))
    };
    // weight sets: tile t's chunks sit in wst[t & 1]; tile t stores wst[(t + 1) & 1] and refills it with tile t + 3
    for (int k0 = 0; k0 < K; k0 += 6 * BK) {
        tile(k0, raw0, raw2, wB, 0);
        if (k0 + BK < K) tile(k0 + BK, raw1, raw0, wA, 1);
        if (k0 + 2 * BK < K) tile(k0 + 2 * BK, raw2, raw1, wB, 0);
        if (k0 + 3 * BK < K) tile(k0 + 3 * BK, raw0, raw2, wA, 1);
        if (k0 + 4 * BK < K) tile(k0 + 4 * BK, raw1, raw0, wB, 0);
        if (k0 + 5 * BK < K) tile(k0 + 5 * BK, raw2, raw1, wA, 1);
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int n = n0 + 16 * b + lr;
        const float bs = bias[n];
#pragma unroll
        for (int a = 0; a < TMF; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[a][b][r] + bs;
                v = v > 0.f ? v : __expf(v) - 1.0f;
                C[(size_t)(m0 + 16 * a + 4 * lg + r) * ldc + n] = v;
            }
    }
}

static void split_host(float x, uint16_t &h, uint16_t &m, uint16_t &l) {
    auto rn = [](float v) { union { float f; uint32_t u; } c; c.f = v; uint32_t r = c.u + 0x7fffu + ((c.u >> 16) & 1u); return (uint16_t)(r >> 16); };
    auto up = [](uint16_t b) { union { float f; uint32_t u; } c; c.u = (uint32_t)b << 16; return c.f; };
    h = rn(x); float r = x - up(h); m = rn(r); r -= up(m); l = rn(r);
}

int main(int argc, char **argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 49152, N = argc > 2 ? atoi(argv[2]) : 256, K = argc > 3 ? atoi(argv[3]) : 512;
    std::vector<float> hA((size_t)M * K), hW((size_t)N * K), hb(N);
    srand(1);
    for (auto &v : hA) v = (float)rand() / RAND_MAX * 2 - 1;
    for (auto &v : hW) v = ((float)rand() / RAND_MAX * 2 - 1) * 0.1f;
    for (auto &v : hb) v = (float)rand() / RAND_MAX - 0.5f;
    const int64_t pls = (int64_t)N * K;
    std::vector<uint16_t> hpl(3 * pls);
    for (int64_t i = 0; i < pls; ++i) split_host(hW[i], hpl[i], hpl[pls + i], hpl[2 * pls + i]);
    float *A, *b, *C; uint16_t *pl;
    hipMalloc(&A, hA.size() * 4); hipMalloc(&b, N * 4); hipMalloc(&C, (size_t)M * N * 4); hipMalloc(&pl, hpl.size() * 2);
    hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice); hipMemcpy(b, hb.data(), N * 4, hipMemcpyHostToDevice);
    hipMemcpy(pl, hpl.data(), hpl.size() * 2, hipMemcpyHostToDevice);
    int flags = 0, tmf = 4;
    auto run = [&](int bn) {
        dim3 grid((M / (64 * tmf)) * (N / bn));
#define L(T, F) hipLaunchKernelGGL((k_fwd<64, T, F>), grid, dim3(256), 0, 0, A, K, pl, pls, b, C, N, M, N, K)
        if (tmf == 4) switch (flags) { case 1: L(4, 1); break; case 4: L(4, 4); break; case 5: L(4, 5); break; case 8: L(4, 8); break; default: L(4, 0); }
        else switch (flags) { case 1: L(2, 1); break; case 4: L(2, 4); break; case 5: L(2, 5); break; case 8: L(2, 8); break; default: L(2, 0); }
    };
    for (int fl : {0, 1, 4, 5, 8, 100, 101, 104, 105, 108}) {
        const int bn = 64;
        flags = fl % 100; tmf = fl >= 100 ? 2 : 4;
        if (N % bn) continue;
        run(bn);
        hipDeviceSynchronize();
        std::vector<float> hC((size_t)M * N);
        hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost);
        double maxerr = 0;
        for (int t = 0; t < 2000; ++t) {
            const int i = rand() % M, j = rand() % N;
            double s = hb[j];
            for (int k = 0; k < K; ++k) s += (double)hA[(size_t)i * K + k] * hW[(size_t)j * K + k];
            s = s > 0 ? s : std::exp(s) - 1.0;
            maxerr = std::fmax(maxerr, std::fabs(s - hC[(size_t)i * N + j]));
        }
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int w = 0; w < 3; ++w) run(bn);
        hipEventRecord(e0);
        const int reps = 20;
        for (int r = 0; r < reps; ++r) run(bn);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / reps;
        printf("TMF %d flags %2d A-direct fwd M=%d N=%d K=%d BN=%d: %.1f us  %.1f TF fp32-equivalent (%.3f of 416.7)  max err %.2e\n", tmf, flags, M, N, K, bn, us,
               2.0 * M * N * K / us / 1e6, 2.0 * M * N * K / us / 1e6 / 416.7, maxerr);
    }
    return 0;
}
