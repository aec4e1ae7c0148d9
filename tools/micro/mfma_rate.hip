// Issue rate of v_mfma_f32_32x32x16_bf16 on gfx950: cycles (s_memtime) per MFMA for one wave per SIMD, with NACC independent
// accumulators (NACC = 1: every MFMA depends on the previous one).   hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC>
__global__ void __launch_bounds__(256) k(float *out, unsigned long long *cyc, int iters) {
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    uint4 ua = make_uint4(threadIdx.x, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u), ub = make_uint4(0x3f803f80u, threadIdx.x * 3u, 0x3f803f80u, 1u);
    bf16x8 x = __builtin_bit_cast(bf16x8, ua), y = __builtin_bit_cast(bf16x8, ub);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[a], 0, 0, 0);
    }
    float s = 0.f;
    for (int a = 0; a < NACC; ++a)
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC>
void run(int blocks, int waves_note) {
    float *out; unsigned long long *cyc;
    hipMalloc(&out, blocks * 256 * 4); hipMalloc(&cyc, blocks * 8);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[1]; hipMemcpy(h, cyc, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * 8 * NACC;
    printf("accumulators %d, %4d blocks x 4 waves: %6.1f s_memtime ticks per MFMA per wave, %7.1f ns per MFMA per SIMD -> %7.1f TF bf16 whole chip at this occupancy\n",
           NACC, blocks, h[0] / n, ms * 1e6 / n, blocks * 4.0 * n * 32768.0 / (ms * 1e-3) / 1e12);
    hipFree(out); hipFree(cyc);
}
int main() {
    run<1>(256, 1); run<2>(256, 1); run<4>(256, 1); run<4>(512, 2); run<4>(1024, 4); run<1>(1024, 4);
    return 0;
}
