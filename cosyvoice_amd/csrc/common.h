// cosyvoice_amd — device-side helpers shared by every gfx950 kernel.
// Wave = 64 lanes everywhere (CDNA4).  All reductions are fixed-order (deterministic).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cv {

// register budget of a kernel as an occupancy window (the CPU emulator's hip_runtime.h blanks it)
#ifndef CV_WAVES_PER_EU
#define CV_WAVES_PER_EU(lo, hi) __attribute__((amdgpu_waves_per_eu(lo, hi)))
#endif

typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned short bf16_t;   // raw bfloat16 bits (weights are stored as bf16, math is fp32)

constexpr int WAVE = 64;

// Padding (dwords) of an LDS tile row whose MFMA fragments are fetched with ds_read_b128 as "lane (r = lane & 15, g = lane >> 4) reads 16 bytes at
// row r, dword 4 g".  The LDS serves a b128 read in four groups of 16 lanes - {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32
// (MI355X_MICROARCH.md, LDS table) - i.e. eight rows at g and the other eight at g + 1, over 64 banks: the 16 slots of a group are distinct exactly
// when the row pitch is 8 (mod 16) dwords.  Rounds 1-3 padded by 4 (pitch 4 mod 16: every fragment read 2-way conflicted, 8 LDS cycles instead of 4).
// 0, in a form the optimiser cannot fold: keeps a load whose address adds it where the source puts it (inside a loop, next to the loads it should travel with).
#ifndef CV_OPAQUE_ZERO
__device__ __forceinline__ int cv_opaque_zero() { int z; asm volatile("v_mov_b32 %0, 0" : "=v"(z)); return z; }
#else
__device__ __forceinline__ int cv_opaque_zero() { return 0; }
#endif

constexpr int LDS_PAD = 8;

__device__ __forceinline__ float bf16_to_f32(bf16_t h) { return __uint_as_float(((unsigned)h) << 16); }

// activation ids (shared with the host side: keep in sync with api.cpp / cosyvoice_amd.h)
enum Act : int {
    ACT_NONE = 0, ACT_SILU = 1, ACT_GELU_ERF = 2, ACT_ELU = 3, ACT_LEAKY = 4, ACT_TANH = 5,
    ACT_MISH = 6, ACT_ABS = 7, ACT_SNAKE = 8,
    ACT_LOGCLAMP = 9,      // log(max(x, p)): log-mel of matcha.utils.audio.mel_spectrogram
    ACT_GELU_TANH = 10,    // nn.GELU(approximate='tanh'): the DiT feed-forward (flow/DiT/modules.py:514)
    ACT_RELU = 11,         // CosyVoice-300M: LegacyLinearNoSubsampling and the TransformerEncoder feed-forward (transformer/subsampling.py:338-383)
};

// Activations.  At ~1 workgroup per CU nothing hides VALU work, and libm's erff / tanhf / log1pf expand to 40-150 instructions per
// element (inlined at every unrolled epilogue site they also made the GEMM kernels 20-90 K instructions long), so the hot
// activations are written on the hardware transcendentals: v_exp_f32 (exp2), v_rcp_f32.
//   SiLU  x / (1 + e^-x)                       GELU  0.5 x (1 + erf(x / sqrt 2)), erf by Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7)
//   Mish  x tanh(ln(1 + e^x)) = x w / (w + 2), w = e^x (e^x + 2)             tanh  1 - 2 / (e^2x + 1)
// Absolute error <= 3e-7 * max(1, |x|) against the libm forms (tests/test_ops.py::test_activations); ELU keeps expm1f (cancellation
// at 0, and it is only used by the f0 predictor).
__device__ __forceinline__ float fast_exp(float x) { return exp2f(x * 1.4426950408889634f); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_erf(float x) {
    const float ax = fabsf(x), t = fast_rcp(1.f + 0.3275911f * ax);
    const float poly = ((((1.061405429f * t - 1.453152027f) * t + 1.421413741f) * t - 0.284496736f) * t + 0.254829592f) * t;
    const float y = 1.f - poly * fast_exp(-ax * ax);
    return x < 0.f ? -y : y;
}
// GELU(erf) of four values on the PACKED fp32 pipe (v_pk_mul_f32 / v_pk_add_f32: two lanes' worth of arithmetic per issue slot, gfx90a+): the expressions of fast_erf
// and of act4_call's GELU branch, operation for operation - every product and sum is rounded once in both forms, so the results are the same bits.  The exponential
// is the bare v_exp_f32: exp2f() only adds the rescaling of results below 2^-126, and those vanish in `1 - poly * e` either way.  Round 5: the FF1 epilogue of the flow
// estimator evaluates 64 x 1024 of these per 64-row band - at ~32 scalar instructions each it, not the matrix pipe, was what a band took (tools/ubench/band_probe).
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f gelu_erf2(v2f t) {
    const v2f x = t * 0.70710678118654752440f;
    const v2f ax = __builtin_elementwise_abs(x);
    const v2f den = 1.f + 0.3275911f * ax;
    v2f r; r.x = fast_rcp(den.x); r.y = fast_rcp(den.y);
    const v2f poly = ((((1.061405429f * r - 1.453152027f) * r + 1.421413741f) * r - 0.284496736f) * r + 0.254829592f) * r;
    const v2f a = (-ax * ax) * 1.4426950408889634f;
    v2f e; e.x = __builtin_amdgcn_exp2f(a.x); e.y = __builtin_amdgcn_exp2f(a.y);
    const v2f y = 1.f - poly * e;
    v2f er; er.x = x.x < 0.f ? -y.x : y.x; er.y = x.y < 0.f ? -y.y : y.y;
    return 0.5f * t * (1.f + er);
}
__device__ __forceinline__ float4 gelu_erf4(float4 v) {
    const v2f a = gelu_erf2((v2f){v.x, v.y}), b = gelu_erf2((v2f){v.z, v.w});
    return make_float4(a.x, a.y, b.x, b.y);
}
// ONE out-of-line copy per kernel, four elements per call: inlined at every unrolled prologue / epilogue site the activation code
// bloats the GEMM k-loop past the instruction cache again (measured: +8..15 us per launch on the 64x64 / 128x64 tiles).
// The activation id is uniform over a launch but arrives in a VECTOR register (a function argument): compared there, every `if (act == ...)` becomes an
// exec-mask region and all seven candidates are walked per element (945 instructions for four values; measured round 4: the GELU epilogue of a 64 x 64 FF1 tile
// cost 7600 clocks per wave against 2900 for the same tile without activation).  Read into a scalar register first, the chain is seven scalar compares and ONE
// body.  Same expressions per element as before: same bits.
__device__ __noinline__ float4 act4_call(int act_v, float4 x) {
    const int act = __builtin_amdgcn_readfirstlane(act_v);
    float v[4] = {x.x, x.y, x.z, x.w};
    if (act == ACT_SILU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float t = v[e]; v[e] = t * fast_rcp(1.f + fast_exp(-t)); }
    } else if (act == ACT_GELU_ERF) {
        const float4 g = gelu_erf4(x);                       // the packed form of 0.5 t (1 + fast_erf(t / sqrt 2)): same bits
        v[0] = g.x; v[1] = g.y; v[2] = g.z; v[3] = g.w;
    } else if (act == ACT_MISH) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float t = v[e]; const float n = fast_exp(fminf(t, 20.f)), w = n * (n + 2.f); v[e] = t > 20.f ? t : t * w * fast_rcp(w + 2.f); }
    } else if (act == ACT_TANH) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float t = v[e]; const float e2 = fast_exp(2.f * fminf(fmaxf(t, -15.f), 15.f)); v[e] = 1.f - 2.f * fast_rcp(e2 + 1.f); }
    } else if (act == ACT_ELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float t = v[e]; v[e] = t > 0.f ? t : expm1f(t); }
    } else if (act == ACT_GELU_TANH) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float t = v[e];
            const float u = 0.7978845608028654f * (t + 0.044715f * t * t * t), e2 = fast_exp(2.f * fminf(fmaxf(u, -15.f), 15.f));
            v[e] = 0.5f * t * (2.f - 2.f * fast_rcp(e2 + 1.f));
        }
    }
    return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ float4 apply_act4(int act, float4 v, float p) {
    if (act == ACT_NONE) return v;
    if (act == ACT_LEAKY) return make_float4(v.x > 0.f ? v.x : v.x * p, v.y > 0.f ? v.y : v.y * p, v.z > 0.f ? v.z : v.z * p, v.w > 0.f ? v.w : v.w * p);
    if (act == ACT_ABS) return make_float4(fabsf(v.x), fabsf(v.y), fabsf(v.z), fabsf(v.w));
    if (act == ACT_RELU) return make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
    if (act == ACT_LOGCLAMP) return make_float4(logf(fmaxf(v.x, p)), logf(fmaxf(v.y, p)), logf(fmaxf(v.z, p)), logf(fmaxf(v.w, p)));
    return act4_call(act, v);
}
__device__ __forceinline__ float apply_act(int act, float v, float p) {      // scalar sites (tails, small kernels)
    if (act == ACT_NONE) return v;
    if (act == ACT_LEAKY) return v > 0.f ? v : v * p;
    if (act == ACT_ABS) return fabsf(v);
    if (act == ACT_RELU) return fmaxf(v, 0.f);
    if (act == ACT_LOGCLAMP) return logf(fmaxf(v, p));
    return act4_call(act, make_float4(v, 0.f, 0.f, 0.f)).x;
}

// Snake(x) = x + sin^2(alpha x) / (alpha + 1e-9)   (reference: cosyvoice/transformer/activation.py:73-84)
__device__ __noinline__ float snake_f(float x, float alpha) {
    float s = sinf(x * alpha);
    return x + (1.0f / (alpha + 1e-9f)) * s * s;
}

// XCD-aware workgroup remap (cdna_hip_programming.md T1): workgroup b runs on XCD b % 8, each XCD has a private 4 MB L2.
// Returns a bijective index b' such that the workgroups of one XCD get a CONTIGUOUS range of b' — the caller then lays tiles
// out so that neighbours in b' share their big operand, which therefore crosses the fabric once instead of once per XCD.
__device__ __forceinline__ int xcd_remap(int b, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, xcd = b & 7, j = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
}

__device__ __forceinline__ float group16_sum(float v);
__device__ __forceinline__ float group16_max(float v);
__device__ __forceinline__ float wave_sum(float v) {
    v = group16_sum(v);
    v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    v = group16_max(v);
    v = fmaxf(v, __shfl_xor(v, 16)); v = fmaxf(v, __shfl_xor(v, 32));
    return v;
}
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

// fp32 pair -> packed bf16 pair, round-to-nearest-even: one v_cvt_pk_bf16_f32 on gfx950 (the same conversion the host compiler
// emits for the emulator build and torch's .bfloat16() performs)
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}

// Buffer addressing: a 128-bit resource (base, byte size) + a 32-bit byte offset per lane.  The hardware range-checks every dword
// against the size and returns 0 outside, so zero padding, ragged tiles and masked rows cost no compare / select / 64-bit add in the
// inner loops (a negative offset is a huge unsigned one).  BUF_OOB is the offset used to force a zero: operands must be < 1.75 GiB.
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
constexpr int BUF_OOB = 0x7F000000;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float4 buf_load_f4(__amdgpu_buffer_rsrc_t rs, int byte_off) {
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 0);
    return make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
}
__device__ __forceinline__ uint2 buf_load_u2(__amdgpu_buffer_rsrc_t rs, int byte_off) {
    const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(rs, byte_off, 0, 0);
    return make_uint2(v[0], v[1]);
}

// Exchanges inside a 16-lane DPP row run on the VALU (no ds_bpermute round trip through the LDS crossbar): xor 1 / xor 2 as
// quad permutes, then "the other quad of my half" (row_half_mirror) and "the other half of my row" (row_mirror).  After the
// two quad steps every lane of a quad holds the same value, so the mirrors pair equal partners: all 16 lanes end bit-identical.
template <int CTRL>
__device__ __forceinline__ float dpp_row(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
// reduce over aligned groups of 16 lanes
__device__ __forceinline__ float group16_sum(float v) {
    v += dpp_row<0xB1>(v);          // quad_perm [1,0,3,2]
    v += dpp_row<0x4E>(v);          // quad_perm [2,3,0,1]
    v += dpp_row<0x141>(v);         // row_half_mirror
    v += dpp_row<0x140>(v);         // row_mirror
    return v;
}
__device__ __forceinline__ float group16_max(float v) {
    v = fmaxf(v, dpp_row<0xB1>(v)); v = fmaxf(v, dpp_row<0x4E>(v)); v = fmaxf(v, dpp_row<0x141>(v)); v = fmaxf(v, dpp_row<0x140>(v));
    return v;
}

// LDS-DMA: 16 bytes per lane from global memory straight into LDS (global_load_lds_dwordx4, cdna_hip_programming.md sections 5 and 5.7): no staging registers,
// no ds_write pass.  The LDS destination is WAVE-UNIFORM base + lane * 16 (M0; `lds_wave_base` must be the same in every lane), the global source is per lane - a
// swizzled LDS image is obtained by permuting the SOURCE addresses.  Issued through inline asm ON PURPOSE: the compiler's own builtin is counted by its waitcnt
// pass, which cannot tell the DMA's destination from the LDS buffer being read and puts a vmcnt(0) in front of every fragment read (the DMA then never overlaps
// the MFMAs; measured round 4).  Hidden in asm the DMA is not counted at all: the kernel waits for it itself - CV_VMCNT0() before the barrier that publishes the
// stage (a __syncthreads() does NOT wait for it).  M0 is saved and restored inside the statement (guide 5.7).  The CPU emulator's hip_runtime.h supplies its own
// CV_GLDS16 (a 16-byte copy per lane) and an empty CV_VMCNT0.
#ifndef CV_GLDS16
__device__ __forceinline__ void cv_glds16(const void* gsrc, const void* lds_wave_base) {
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)lds_wave_base);     // low 32 bits of a flat LDS address = the LDS byte offset
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}
#define CV_GLDS16(gptr, lds_wave_base) cv_glds16((gptr), (lds_wave_base))
#define CV_VMCNT0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define CV_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")      // counted: the newest n vector-memory operations (DMA pieces included) may still be in flight
#endif

// Scheduling fence: the machine scheduler moves no instruction across it (no instruction is emitted).  Used where the ORDER of independent global
// loads matters - the vector-memory counter retires in issue order, so what is requested first can be waited for first.  (An asm "memory" clobber would
// do the same to the loads but forces register arrays that live across it into scratch.)
__device__ __forceinline__ void order_memory() { __builtin_amdgcn_sched_barrier(0); }

// LDS hand-off between the lanes of ONE wave (a region no other wave touches): a wave's DS operations execute in issue order, so all that is needed
// is that every lane has issued its writes and that the compiler keeps the order - a wave barrier, no s_barrier, no drain of the vector-memory counter
// (loads requested before the hand-off stay in flight).
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// block-wide sum for blocks of up to 16 waves; `red` is an LDS scratch of >= 16 floats.
__device__ __forceinline__ float block_sum(float v, float* red) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}
__device__ __forceinline__ float block_max(float v, float* red) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_max(v);
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float t = red[0];
    for (int i = 1; i < nw; ++i) t = fmaxf(t, red[i]);
    return t;
}

}  // namespace cv
