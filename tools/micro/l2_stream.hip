// What one CU can pull out of the L2 with 16-byte-per-lane loads: every workgroup streams the SAME `span` bytes (L2 resident after the
// first pass), 1 KB per wave-load, `waves` waves per CU; bytes per cycle per CU by s_memtime.  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void __launch_bounds__(1024) k(const uint4 *__restrict__ src, unsigned *out, unsigned long long *cyc, int n16, int passes) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    uint4 acc = make_uint4(0, 0, 0, 0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int p = 0; p < passes; ++p)
        for (int i = wave * 64 + lane; i + 7 * nw * 64 < n16; i += 8 * nw * 64) {
            uint4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[i + u * nw * 64];
#pragma unroll
            for (int u = 0; u < 8; ++u) { acc.x ^= v[u].x; acc.y += v[u].y; acc.z ^= v[u].z; acc.w += v[u].w; }
        }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x ^ acc.y ^ acc.z ^ acc.w;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    const size_t span = 1u << 20;                       // 1 MB: the weight image of one net of the rollout forward
    uint4 *src; unsigned *out; unsigned long long *cyc;
    hipMalloc(&src, span); hipMemset(src, 1, span); hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 8);
    for (int waves : {4, 8, 16}) {
        const int passes = 20, n16 = span / 16;
        hipLaunchKernelGGL(k, dim3(256), dim3(64 * waves), 0, 0, src, out, cyc, n16, 2);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256), dim3(64 * waves), 0, 0, src, out, cyc, n16, passes);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        double mean = 0; for (int i = 0; i < 256; ++i) mean += h[i]; mean /= 256;
        const double bytes = (double)passes * (span / (8.0 * waves * 1024)) * 0 + (double)passes * span;   // every workgroup reads the whole span per pass
        printf("%2d waves per CU, 256 workgroups: %6.1f bytes per cycle per CU (s_memtime), %7.2f TB/s over the chip, kernel %.1f us\n", waves,
               bytes / mean, 256.0 * bytes / (ms * 1e-3) / 1e12, ms * 1e3);
    }
    return 0;
}
