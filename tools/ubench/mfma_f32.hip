// Calibration microbenchmark (dev tool): issue rate of v_mfma_f32_16x16x4_f32 and of the bf16 16x16x32 form on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef short v8s __attribute__((ext_vector_type(8)));
template <int NACC>
__global__ __launch_bounds__(256) void k_f32(float* out, int iters, float a, float b) {
    v4f acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (v4f){0.f, 0.f, 0.f, 0.f};
    long long t0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    long long t1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s + (float)(t1 - t0) * 1e-30f;
}
template <int NACC>
__global__ __launch_bounds__(256) void k_bf16(float* out, int iters, short a, short b) {
    v4f acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (v4f){0.f, 0.f, 0.f, 0.f};
    v8s va = {a, a, a, a, a, a, a, a}, vb = {b, b, b, b, b, b, b, b};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, vb, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename F> float timeit(F f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0, 0); f(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    float* out; hipMalloc(&out, 4096 * 256 * 4);
    const int iters = 20000;
    for (int blocks : {256, 512, 1024}) {
        float m1 = timeit([&] { hipLaunchKernelGGL((k_f32<1>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 0.5f); });
        float m4 = timeit([&] { hipLaunchKernelGGL((k_f32<4>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 0.5f); });
        float b4 = timeit([&] { hipLaunchKernelGGL((k_bf16<4>), dim3(blocks), dim3(256), 0, 0, out, iters, (short)0x3f80, (short)0x3f00); });
        // per wave: iters*NACC MFMAs; blocks of 4 waves, 256 CUs x 4 SIMDs
        double waves_per_simd = blocks * 4.0 / 1024.0;
        printf("blocks=%4d  f32 dep-chain: %.1f ns/MFMA/wave   f32 4 indep: %.1f ns/MFMA/wave (%.1f TF)   bf16 16x16x32 4 indep: %.1f ns/MFMA/wave (%.0f TF)\n", blocks,
               m1 * 1e6 / iters / waves_per_simd, m4 * 1e6 / (iters * 4) / waves_per_simd, blocks * 4.0 * iters * 4 * 2.0 * 16 * 16 * 4 / (m4 * 1e-3) / 1e12,
               b4 * 1e6 / (iters * 4) / waves_per_simd, blocks * 4.0 * iters * 4 * 2.0 * 16 * 16 * 32 / (b4 * 1e-3) / 1e12);
    }
    return 0;
}
