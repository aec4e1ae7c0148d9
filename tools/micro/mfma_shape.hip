// Sustained rate of v_mfma_f32_32x32x16_bf16 against v_mfma_f32_16x16x32_bf16 on gfx950 with RANDOM operands (the chip's clock under
// MFMA load depends on the data: tools/gemm_clock.py), registers only, all SIMDs busy for tens of milliseconds.
//   hipcc --offload-arch=gfx950 -O3 mfma_shape.hip -o _bin/mfma_shape
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ inline unsigned mix(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
// bf16 pairs with exponents near 1.0 and random mantissas / signs (zero = 1: constant operands)
__device__ inline bf16x8 operand(unsigned seed, int zero) {
    unsigned w[4];
    for (int i = 0; i < 4; ++i) {
        const unsigned r = mix(seed * 4u + i);
        w[i] = zero ? 0x3f803f80u : ((r & 0x807f807fu) | 0x3f003f00u);
    }
    return __builtin_bit_cast(bf16x8, make_uint4(w[0], w[1], w[2], w[3]));
}

template <int SHAPE>   // 0: 32x32x16, 4 accumulator blocks; 1: 16x16x32, 16 accumulator blocks (the same 64x64 wave tile)
__global__ void __launch_bounds__(256) k(float *out, int iters, int zero) {
    bf16x8 x[4], y[4];
    for (int i = 0; i < 4; ++i) { x[i] = operand(threadIdx.x * 8 + i, zero); y[i] = operand(threadIdx.x * 8 + 4 + i + blockIdx.x * 4096, zero); }
    float s = 0.f;
    if constexpr (SHAPE == 0) {
        f32x16 acc[2][2];
        for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)            // two k16 steps = one k32
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[a + 2 * kk], y[b + 2 * kk], acc[a][b], 0, 0, 0);
        }
        for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
    } else {
        f32x4 acc[4][4];
        for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) for (int r = 0; r < 4; ++r) acc[a][b][r] = 0.f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x[a], y[b], acc[a][b], 0, 0, 0);
        }
        for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) for (int r = 0; r < 4; ++r) s += acc[a][b][r];
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int SHAPE>
void run(int blocks, int zero, float *out) {
    const int iters = 60000;       // 64x64x32 per iteration and wave
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<SHAPE>, dim3(blocks), dim3(256), 0, 0, out, iters / 4, zero);
    float best = 1e30f, last = 0.f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<SHAPE>, dim3(blocks), dim3(256), 0, 0, out, iters, zero);
        hipEventRecord(e1); hipDeviceSynchronize();
        hipEventElapsedTime(&last, e0, e1);
        if (last < best) best = last;
    }
    const double flop = (double)blocks * 4 * iters * 64.0 * 64.0 * 32.0 * 2.0;
    printf("%s  %s operands, %4d blocks x 4 waves: %7.2f ms (last %7.2f) -> %7.1f TF bf16\n", SHAPE ? "16x16x32" : "32x32x16",
           zero ? "constant" : "random  ", blocks, best, last, flop / (best * 1e-3) / 1e12);
}
int main() {
    float *out; hipMalloc(&out, 2048 * 256 * 4);
    for (int zero = 1; zero >= 0; --zero)
        for (int blocks : {256, 512}) { run<0>(blocks, zero, out); run<1>(blocks, zero, out); }
    run<0>(512, 0, out); run<1>(512, 0, out);
    return 0;
}
