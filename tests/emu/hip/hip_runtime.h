// TEST INFRASTRUCTURE ONLY — not part of the product.
//
// A CPU emulator of the HIP execution model, used so that the gfx950 kernels under
// cosyvoice_amd/csrc can be debugged in a container that has no GPU.  The kernel
// sources are compiled UNMODIFIED with the host clang++ and `-I tests/emu`, so that
// `#include <hip/hip_runtime.h>` resolves to this file.  Every thread of a workgroup
// is a fiber; __syncthreads(), wave shuffles and MFMA are rendezvous points.
//
// Only tests/ may build or load the resulting libcosyvoice_amd_emu.so.  The product
// loader (cosyvoice_amd/_lib.py) never looks for it.
#pragma once
#include <cstdint>
#include <cstddef>
#include <cstring>
#include <cstdlib>
#include <cstdio>
#include <cmath>
#include <functional>
#include <algorithm>
using std::min; using std::max;

#define CV_EMU 1

// ---- qualifiers -----------------------------------------------------------
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __noinline__ inline __attribute__((noinline))
#define __shared__ static
#define __launch_bounds__(...)
#define CV_WAVES_PER_EU(lo, hi)
#define __restrict__ __restrict

// ---- basic types ----------------------------------------------------------
struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3_ { unsigned x, y, z; };
extern uint3_ threadIdx, blockIdx;
extern dim3 blockDim, gridDim;

struct float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct uint2 { unsigned x, y; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
struct ushort4 { unsigned short x, y, z, w; };
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }

typedef int hipError_t;
#define hipSuccess 0
#define hipErrorUnknown 999
typedef struct emuStream_* hipStream_t;
typedef struct emuEvent_* hipEvent_t;
typedef struct emuGraph_* hipGraph_t;
typedef struct emuGraphExec_* hipGraphExec_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum hipStreamCaptureMode { hipStreamCaptureModeGlobal, hipStreamCaptureModeThreadLocal, hipStreamCaptureModeRelaxed };

// ---- runtime API subset ----------------------------------------------------
namespace emu {
void launch(dim3 grid, dim3 block, hipStream_t s, std::function<void()> body);
void* wave_exchange(const void* mine, size_t bytes);   // returns snapshot base; lane i at base + i*SLOT
void block_barrier();
int lane_id();
constexpr int SLOT = 256;
}

hipError_t hipMalloc(void** p, size_t n);
hipError_t hipFree(void* p);
hipError_t hipHostMalloc(void** p, size_t n, unsigned flags = 0);
hipError_t hipHostFree(void* p);
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind k);
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t st);
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t st);
hipError_t hipMemcpy2DAsync(void* d, size_t dpitch, const void* s, size_t spitch, size_t width, size_t height, hipMemcpyKind k, hipStream_t st);
hipError_t hipMemset2DAsync(void* d, size_t dpitch, int v, size_t width, size_t height, hipStream_t st);
hipError_t hipMemset(void* d, int v, size_t n);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipDeviceSynchronize();
hipError_t hipGetLastError();
hipError_t hipPeekAtLastError();
const char* hipGetErrorString(hipError_t e);
hipError_t hipStreamCreate(hipStream_t* s);
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { return hipStreamCreate(s); }
constexpr unsigned hipStreamNonBlocking = 1, hipEventDisableTiming = 2;
hipError_t hipEventCreate(hipEvent_t* e);
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipEventCreate(hipEvent_t* e);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags = 0);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);
hipError_t hipStreamBeginCapture(hipStream_t s, hipStreamCaptureMode m);
hipError_t hipStreamEndCapture(hipStream_t s, hipGraph_t* g);
hipError_t hipGraphInstantiate(hipGraphExec_t* e, hipGraph_t g, void*, void*, size_t);
hipError_t hipGraphLaunch(hipGraphExec_t e, hipStream_t s);
hipError_t hipGraphDestroy(hipGraph_t g);
hipError_t hipGraphExecDestroy(hipGraphExec_t e);
hipError_t hipSetDevice(int d);
hipError_t hipGetDevice(int* d);

// kernel launch: arguments are captured BY VALUE at launch (like a real launch / graph capture)
#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...) \
    emu::launch((grid), (block), (stream), [=]() { kern(__VA_ARGS__); })

// ---- device intrinsics ------------------------------------------------------
static inline void __syncthreads() { emu::block_barrier(); }

template <typename T>
static inline T __shfl(T v, int src, int width = 64) {
    char* base = (char*)emu::wave_exchange(&v, sizeof(T));
    int lane = emu::lane_id();
    int grp = lane / width * width;
    T r; memcpy(&r, base + (size_t)(grp + (src % width)) * emu::SLOT, sizeof(T)); return r;
}
template <typename T>
static inline T __shfl_xor(T v, int mask, int width = 64) {
    char* base = (char*)emu::wave_exchange(&v, sizeof(T));
    int lane = emu::lane_id();
    int src = lane ^ mask;
    if (src / width != lane / width) src = lane;
    T r; memcpy(&r, base + (size_t)src * emu::SLOT, sizeof(T)); return r;
}
template <typename T>
static inline T __shfl_down(T v, unsigned d, int width = 64) {
    char* base = (char*)emu::wave_exchange(&v, sizeof(T));
    int lane = emu::lane_id();
    int src = lane + (int)d;
    if (src / width != lane / width) src = lane;
    T r; memcpy(&r, base + (size_t)src * emu::SLOT, sizeof(T)); return r;
}
template <typename T>
static inline T __shfl_up(T v, unsigned d, int width = 64) {
    char* base = (char*)emu::wave_exchange(&v, sizeof(T));
    int lane = emu::lane_id();
    int src = lane - (int)d;
    if (src < 0 || src / width != lane / width) src = lane;
    T r; memcpy(&r, base + (size_t)src * emu::SLOT, sizeof(T)); return r;
}
unsigned long long emu_ballot(int pred);
static inline unsigned long long __ballot(int pred) { return emu_ballot(pred); }
static inline int __all(int pred) { return emu_ballot(!pred) == 0ull; }
static inline int __any(int pred) { return emu_ballot(pred) != 0ull; }

static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static inline int __float_as_int(float f) { int u; memcpy(&u, &f, 4); return u; }
static inline float __int_as_float(int u) { float f; memcpy(&f, &u, 4); return f; }
static inline float __fdividef(float a, float b) { return a / b; }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline float __frcp_rn(float x) { return 1.0f / x; }
static inline long long clock64() { return 0; }
static inline long long wall_clock64() { return 0; }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }

float atomicAdd(float* p, float v);
int atomicAdd(int* p, int v);
unsigned atomicAdd(unsigned* p, unsigned v);
int atomicMax(int* p, int v);

// MFMA emulation (layouts per cdna_hip_programming.md §3)
typedef float emu_v4f __attribute__((ext_vector_type(4)));
typedef float emu_v16f __attribute__((ext_vector_type(16)));
typedef short emu_v8s __attribute__((ext_vector_type(8)));
emu_v4f emu_mfma_f32_16x16x4f32(float a, float b, emu_v4f c);
emu_v16f emu_mfma_f32_32x32x2f32(float a, float b, emu_v16f c);
emu_v4f emu_mfma_f32_16x16x32_bf16(emu_v8s a, emu_v8s b, emu_v4f c);
emu_v16f emu_mfma_f32_32x32x16_bf16(emu_v8s a, emu_v8s b, emu_v16f c);
emu_v4f emu_mfma_f32_16x16x32_fp8_fp8(long a, long b, emu_v4f c);
int emu_cvt_pk_fp8_f32(float a, float b, int old, bool word_sel);
#define __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a, b, c, x, y, z) emu_mfma_f32_16x16x32_fp8_fp8((a), (b), (c))
#define __builtin_amdgcn_cvt_pk_fp8_f32(a, b, old, sel) emu_cvt_pk_fp8_f32((a), (b), (old), (sel))
#define __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, x, y, z) emu_mfma_f32_16x16x4f32((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, x, y, z) emu_mfma_f32_32x32x2f32((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, x, y, z) emu_mfma_f32_16x16x32_bf16(__builtin_bit_cast(emu_v8s, (a)), __builtin_bit_cast(emu_v8s, (b)), (c))
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z) emu_mfma_f32_32x32x16_bf16(__builtin_bit_cast(emu_v8s, (a)), __builtin_bit_cast(emu_v8s, (b)), (c))
// DPP row controls used by the kernels (gfx9 encodings): quad_perm 0x00-0xFF, row_half_mirror 0x141, row_mirror 0x140
static inline int emu_update_dpp(int src, int ctrl) {
    int lane = emu::lane_id(), from;
    if (ctrl < 0x100) from = (lane & ~3) | ((ctrl >> (2 * (lane & 3))) & 3);
    else if (ctrl == 0x141) from = (lane & ~7) | (7 - (lane & 7));
    else if (ctrl == 0x140) from = (lane & ~15) | (15 - (lane & 15));
    else { fprintf(stderr, "emu: unsupported dpp_ctrl 0x%x\n", ctrl); abort(); }
    return __shfl(src, from);
}
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rm, bm, bc) emu_update_dpp((src), (ctrl))
// buffer resources: (base, byte size); loads return 0 for every dword that is not entirely inside [0, size)
struct emu_rsrc { const char* base; unsigned bytes; };
typedef emu_rsrc __amdgpu_buffer_rsrc_t;
typedef unsigned emu_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned emu_u32x2 __attribute__((ext_vector_type(2)));
static inline emu_rsrc emu_make_rsrc(void* p, unsigned bytes) { return emu_rsrc{(const char*)p, bytes}; }
static inline unsigned emu_buf_dword(emu_rsrc rs, unsigned off) { unsigned v = 0; if (off <= rs.bytes && rs.bytes - off >= 4) memcpy(&v, rs.base + off, 4); return v; }
static inline emu_u32x4 emu_buf_load_b128(emu_rsrc rs, int voff, int soff) {
    const unsigned o = (unsigned)voff + (unsigned)soff;
    return emu_u32x4{emu_buf_dword(rs, o), emu_buf_dword(rs, o + 4), emu_buf_dword(rs, o + 8), emu_buf_dword(rs, o + 12)};
}
static inline emu_u32x2 emu_buf_load_b64(emu_rsrc rs, int voff, int soff) {
    const unsigned o = (unsigned)voff + (unsigned)soff;
    return emu_u32x2{emu_buf_dword(rs, o), emu_buf_dword(rs, o + 4)};
}
#define __builtin_amdgcn_make_buffer_rsrc(p, stride, bytes, flags) emu_make_rsrc((p), (bytes))
#define __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, aux) emu_buf_load_b128((rs), (voff), (soff))
#define __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, aux) emu_buf_load_b64((rs), (voff), (soff))
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))
#define __builtin_amdgcn_readfirstlane(x) (x)      /* callers pass wave-uniform values (also inside divergent regions, where a fibre rendezvous would not complete) */
/* global_load_lds_dwordx4: LDS destination = wave-uniform base + lane * 16 */
#define CV_GLDS16(gptr, lds_wave_base) memcpy((char*)(lds_wave_base) + 16 * emu::lane_id(), (const void*)(gptr), 16)
#define CV_GLDS16S(sbase, voff, lds_wave_base) memcpy((char*)(lds_wave_base) + 16 * emu::lane_id(), (const char*)(sbase) + (voff), 16)
#define CV_VMCNT0() ((void)0)
#define CV_VMCNT(n) ((void)0)
/* v_permlane32_swap: the upper half of the first operand and the lower half of the second trade places; returns {new first, new second} */
static inline emu_u32x2 emu_permlane32_swap(unsigned a, unsigned b) {
    const int l = emu::lane_id();
    const unsigned a_other = __shfl(a, l ^ 32), b_other = __shfl(b, l ^ 32);
    return l < 32 ? emu_u32x2{a, a_other} : emu_u32x2{b_other, b};
}
#define __builtin_amdgcn_permlane32_swap(a, b, fi, bc) emu_permlane32_swap((a), (b))
#define CV_OPAQUE_ZERO 1
#define __builtin_amdgcn_exp2f(x) exp2f(x)                              /* v_exp_f32: callers stay out of the range where it flushes (results below 2^-126) */
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_wave_barrier() ((void)__shfl(0, 0))            /* the emulator's lanes are fibers: a wave-wide rendezvous */
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_s_sleep(x) ((void)0)
#define __builtin_amdgcn_sched_group_barrier(mask, size, id) ((void)0)
#define __builtin_amdgcn_s_getreg(x) 0      /* hardware id registers (dev-tool stamps only) */
#define __builtin_nontemporal_load(p) (*(p))
#define __builtin_nontemporal_store(v, p) (*(p) = (v))
