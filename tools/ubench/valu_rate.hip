// Calibration microbenchmark (dev tool, round 6): issue cost of the VALU instructions a flash-attention softmax is made of, alone and side by side with each
// other and with the bf16 MFMA, at 1 / 2 / 4 waves per SIMD.  Answers: is v_exp_f32 a quarter-rate instruction on gfx950, does it share issue slots with plain /
// packed fp32 arithmetic, and how many VALU slots does one v_mfma_f32_32x32x16_bf16 hide.  Each wave runs `iters` rounds of an unrolled body on 16 independent
// register chains; cycles per wave-instruction = s_memtime delta of wave 0 / instructions issued by ONE wave (so at W waves per SIMD the figure is the issue
// interval seen by a wave; x 1/W = the SIMD's cost per instruction).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

enum { T_FMA, T_EXP, T_PKFMA, T_PKMUL, T_MAX3, T_CVT, T_LDEXP, T_FLOOR, T_MAX, T_ADD, T_EXP_FMA1, T_EXP_FMA2, T_EXP_FMA4, T_EXP_PKFMA1, T_EXP_PKFMA2,
       T_MFMA32, T_MFMA16, T_MFMA32_FMA4, T_MFMA32_FMA8, T_MFMA32_EXP2, T_MFMA32_EXP4, T_MFMA32_SOFTMAX, T_MFMA16_FMA2, T_MFMA16_FMA4, T_PERM32, T_DPP, T_SHFL, T_RCP, T_N };
static const char* names[] = {"v_fma_f32", "v_exp_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_max3_f32", "v_cvt_pk_bf16_f32", "v_ldexp_f32", "v_floor_f32", "v_max_f32", "v_add_f32",
                              "exp + 1 fma", "exp + 2 fma", "exp + 4 fma", "exp + 1 pk_fma", "exp + 2 pk_fma", "mfma32x32x16 (4 acc)", "mfma16x16x32 (4 acc)", "mfma32 + 4 fma", "mfma32 + 8 fma",
                              "mfma32 + 2 exp", "mfma32 + 4 exp", "mfma32 + softmax mix (2 exp 2 fma 1 max3 2 add 1 cvt)", "mfma16 + 2 fma", "mfma16 + 4 fma", "v_permlane32_swap", "dpp row mov", "ds_bpermute (shfl_xor 32)", "v_rcp_f32"};
// instructions counted per body (per chain step): what the printed cycles are divided by
static const int per_body[] = {16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 4, 4, 4, 4, 4, 4, 4, 4, 4, 16, 16, 16, 16};

template <int TEST>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, int iters, float seed) {
    float r[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = seed + (float)(threadIdx.x & 7) * 0.001f + (float)i * 0.01f;
    v16f acc32[4]; v4f acc16[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { acc32[i] = (v16f)(0.f); acc16[i] = (v4f)(0.f); }
    const v8bf fa = (v8bf)((__bf16)seed), fb = (v8bf)((__bf16)0.5f);
    const float c1 = 0.999f, c2 = 0.0001f;
    const v2f pc1 = {0.999f, 0.999f}, pc2 = {0.0001f, 0.0001f};
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if constexpr (TEST == T_FMA) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(c1), "v"(c2));
        } else if constexpr (TEST == T_EXP) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));
        } else if constexpr (TEST == T_RCP) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_rcp_f32 %0, %0" : "+v"(r[i]));
        } else if constexpr (TEST == T_PKFMA || TEST == T_PKMUL) {
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                v2f p = {r[i], r[i + 1]};
                if constexpr (TEST == T_PKFMA) { asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p) : "v"(pc1), "v"(pc2)); asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p) : "v"(pc1), "v"(pc2)); }
                else { asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p) : "v"(pc1)); asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p) : "v"(pc1)); }
                r[i] = p.x; r[i + 1] = p.y;
            }
        } else if constexpr (TEST == T_MAX3) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(c1), "v"(c2));
        } else if constexpr (TEST == T_MAX) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(r[i]) : "v"(c1));
        } else if constexpr (TEST == T_ADD) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[i]) : "v"(c2));
        } else if constexpr (TEST == T_CVT) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(r[i]) : "v"(c1));
        } else if constexpr (TEST == T_LDEXP) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(r[i]) : "v"(0));
        } else if constexpr (TEST == T_FLOOR) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_floor_f32 %0, %0" : "+v"(r[i]));
        } else if constexpr (TEST == T_PERM32) {
#pragma unroll
            for (int i = 0; i < 16; i += 2) { asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(r[i]), "+v"(r[i + 1])); asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(r[i]), "+v"(r[i + 1])); }
        } else if constexpr (TEST == T_DPP) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_mov_b32_dpp %0, %0 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(r[i]));
        } else if constexpr (TEST == T_SHFL) {
#pragma unroll
            for (int i = 0; i < 16; ++i) r[i] = __shfl_xor(r[i], 32) + 1.f;       // counts the add too
        } else if constexpr (TEST >= T_EXP_FMA1 && TEST <= T_EXP_FMA4) {
            constexpr int NF = TEST == T_EXP_FMA1 ? 1 : TEST == T_EXP_FMA2 ? 2 : 4;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));
#pragma unroll
                for (int f = 0; f < NF; ++f) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[(i + 5 + f) & 15]) : "v"(c1), "v"(c2));
            }
        } else if constexpr (TEST == T_EXP_PKFMA1 || TEST == T_EXP_PKFMA2) {
            constexpr int NF = TEST == T_EXP_PKFMA1 ? 1 : 2;
            v2f p[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) p[j] = (v2f){r[8 + 2 * j], r[9 + 2 * j]};
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                asm volatile("v_exp_f32 %0, %0" : "+v"(r[i & 7]));
#pragma unroll
                for (int f = 0; f < NF; ++f) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[(i + f) & 3]) : "v"(pc1), "v"(pc2));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) { r[8 + 2 * j] = p[j].x; r[9 + 2 * j] = p[j].y; }
        } else if constexpr (TEST == T_MFMA32 || (TEST >= T_MFMA32_FMA4 && TEST <= T_MFMA32_SOFTMAX)) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc32[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc32[i], 0, 0, 0);
                if constexpr (TEST == T_MFMA32_FMA4 || TEST == T_MFMA32_FMA8) {
                    constexpr int NF = TEST == T_MFMA32_FMA4 ? 4 : 8;
#pragma unroll
                    for (int f = 0; f < NF; ++f) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[(4 * i + f) & 15]) : "v"(c1), "v"(c2));
                } else if constexpr (TEST == T_MFMA32_EXP2 || TEST == T_MFMA32_EXP4) {
                    constexpr int NF = TEST == T_MFMA32_EXP2 ? 2 : 4;
#pragma unroll
                    for (int f = 0; f < NF; ++f) asm volatile("v_exp_f32 %0, %0" : "+v"(r[(4 * i + f) & 15]));
                } else if constexpr (TEST == T_MFMA32_SOFTMAX) {
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[(4 * i) & 15]) : "v"(c1), "v"(c2));
                    asm volatile("v_exp_f32 %0, %0" : "+v"(r[(4 * i + 1) & 15]));
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[(4 * i + 2) & 15]) : "v"(c1), "v"(c2));
                    asm volatile("v_exp_f32 %0, %0" : "+v"(r[(4 * i + 3) & 15]));
                    asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(r[(4 * i + 4) & 15]) : "v"(c1), "v"(c2));
                    asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[(4 * i + 5) & 15]) : "v"(c2));
                    asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[(4 * i + 6) & 15]) : "v"(c2));
                    asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(r[(4 * i + 7) & 15]) : "v"(c1));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        } else if constexpr (TEST == T_MFMA16 || TEST == T_MFMA16_FMA2 || TEST == T_MFMA16_FMA4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc16[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc16[i], 0, 0, 0);
                if constexpr (TEST != T_MFMA16) {
                    constexpr int NF = TEST == T_MFMA16_FMA2 ? 2 : 4;
#pragma unroll
                    for (int f = 0; f < NF; ++f) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[(4 * i + f) & 15]) : "v"(c1), "v"(c2));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += r[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc32[i][0] + acc32[i][5] + acc16[i][0] + acc16[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int TEST>
void run(float* out, long long* cyc, int blocks_per_cu) {
    const int iters = 4000, blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<TEST>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters, 0.25f); hipDeviceSynchronize();
    hipEventRecord(e0, 0); hipLaunchKernelGGL((k<TEST>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters, 0.25f); hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks * 4); hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (auto v : h) mean += (double)v; mean /= h.size();
    const double ninstr = (double)iters * per_body[TEST];
    // wall-derived: ns per counted instruction per SIMD = ms / (ninstr x waves per SIMD)
    printf("%-58s w/SIMD=%d  memtime ticks per counted instr (one wave) %7.2f   wall ns per counted instr per SIMD %6.3f  (%.3f ms)\n", names[TEST], blocks_per_cu, mean / ninstr,
           ms * 1e6 / (ninstr * blocks_per_cu), ms);
}
template <int TEST> void sweep(float* out, long long* cyc) { for (int w : {1, 2, 4}) run<TEST>(out, cyc, w); }
template <int... TS> void all(float* out, long long* cyc, std::integer_sequence<int, TS...>) { (sweep<TS>(out, cyc), ...); }
int main() {
    float* out; long long* cyc; hipMalloc(&out, 2048 * 256 * 4); hipMalloc(&cyc, 2048 * 4 * 8);
    all(out, cyc, std::make_integer_sequence<int, T_N>{});
    return 0;
}
