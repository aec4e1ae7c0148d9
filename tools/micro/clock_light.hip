// In-kernel shader clock under LIGHT load (one wave per SIMD, a dependent VALU chain, ~100 us: the shape of the control-loop kernel) and under a
// saturating VALU load, by the recipe of MI355X_MICROARCH.md (DVFS give-back, item 6): clock = d s_memtime / d s_memrealtime x 100 MHz.
//   hipcc --offload-arch=gfx950 -O3 clock_light.hip -o _bin/clock_light
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

__global__ void __launch_bounds__(256) k(float *out, unsigned long long *st, int iters) {
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    unsigned long long t0, r0, t1, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0) :: "memory");
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) a = a * b + 0.5f;           // dependent chain
    }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) :: "memory");
    out[blockIdx.x * 256 + threadIdx.x] = a;
    if (threadIdx.x == 0) { st[2 * blockIdx.x] = t1 - t0; st[2 * blockIdx.x + 1] = r1 - r0; }
}

static void run(const char *label, int blocks, int iters, int reps, int gap_us) {
    float *out; unsigned long long *st;
    (void)hipMalloc(&out, (size_t)blocks * 256 * 4); (void)hipMalloc(&st, (size_t)blocks * 16);
    std::vector<unsigned long long> h(2 * blocks);
    std::vector<double> clk;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0.f;
    for (int r = 0; r < reps; ++r) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, st, iters);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (gap_us) { hipEvent_t w; (void)w; for (volatile int spin = 0; spin < gap_us * 300; ++spin) {} }
    }
    (void)hipMemcpy(h.data(), st, (size_t)blocks * 16, hipMemcpyDeviceToHost);
    for (int b = 0; b < blocks; ++b) clk.push_back((double)h[2 * b] / (double)h[2 * b + 1] * 0.1);
    std::sort(clk.begin(), clk.end());
    printf("%-46s %5d workgroups, %7.1f us per launch: in-kernel clock median %.2f GHz (min %.2f, max %.2f)\n", label, blocks, ms * 1e3, clk[clk.size() / 2], clk.front(), clk.back());
    (void)hipFree(out); (void)hipFree(st);
}

int main() {
    run("light: 1 wave/SIMD, launches back to back", 256, 600, 200, 0);
    run("light: 1 wave/SIMD, host gaps between launches", 256, 600, 50, 200);
    run("light: quarter of the chip", 64, 600, 200, 0);
    run("heavy: 8 waves/SIMD dependent VALU chains", 256 * 8, 600, 100, 0);
    run("long light launch (2 ms)", 256, 12000, 20, 0);
    return 0;
}
