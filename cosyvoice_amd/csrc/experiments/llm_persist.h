// Prototype (round 3): two dependent decode GEMVs of a Qwen2 layer - gate/up (+RMSNorm, SiLU*up) and down (+residual) - in
// ONE launch, the all-to-all seam between them crossed inside the launch instead of by a kernel boundary.
//
// Structure (one workgroup per CU's worth of rows, all workgroups co-resident - 256 x 320 threads, a few KB of LDS):
//   * run-ahead weights: a workgroup requests its share of BOTH matrices at kernel entry - 19 (gate, up) row pairs (68 KB) and 3-4 rows of
//     down (39 KB) - as non-temporal 16-byte loads straight into registers (56 + 32 VGPRs per lane; with one workgroup per CU the whole
//     512-register file is available, so the "ring" of the LDS-DMA engine of MI355X_MICROARCH.md's price list is simply the register
//     file here: 107 KB per CU in flight from the first cycle).  The down stream arrives while gate/up is being reduced and handed over.
//   * hand-off: the 19 activations a workgroup produces are published as 8-byte {value, tag = epoch} granules, ONE sc1 (write-through) store
//     each - the data is the flag, no fence, no separate flag (guide recipe R2); every workgroup gathers all 4864 granules with GW gather
//     waves sweeping 8 KB chunks with sc1 loads (L1-bypassing, L2-served) until every tag equals this replay's epoch, and parks the values
//     in LDS.  Epoch = a device word that changes on every replay (the decode loop: KV length + 1), so nothing is re-zeroed between replays and
//     a stale granule of the previous token can never be taken for a fresh one.  Spins are bounded (a give-up sets *fail and the kernel
//     ends with garbage instead of hanging the GPU).
//   * the kernel boundary this removes costs 1.2-1.45 us plus the consumer's cold start (first-byte latency of its weight stream);
//     the hand-off costs an allgather (price-list row `allgather`: 2.9-4.2 us for 32 KB with ONE gather wave next to a live stream).
// Measured against the two-launch chain by tools/ubench/persist_probe.hip (profiles/r3_persist_probe.txt); go / no-go in DESIGN.md.
#pragma once
#include "llm_kernels.h"

namespace cv {

typedef unsigned long long u64_t;

struct MlpPairArgs {
    const bf16_t* Wgu;        // [2 I][H], rows interleaved (gate_j, up_j)  (cosyvoice_amd/weights.py layout)
    const bf16_t* Wd;         // [H][I]
    const float* gamma; float eps;
    float* h;                 // [H] residual stream, updated in place
    u64_t* gran;              // [I] granules of THIS launch site (one buffer per layer)
    const int* epoch;         // device word, unique per replay, >= 1
    int* fail;                // set when a bounded spin gave up
    int H, I;
    long long* stamps;        // optional [gridDim][8] wall-clock stamps (100 MHz): entry, act published, gather done, exit
    int mode;                 // 0 = full; 1 = skip the gather (reads stale granule VALUES without checking tags: timing decomposition only)
};

__device__ __forceinline__ u64_t gran_load(const u64_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void gran_store(u64_t* p, u64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// STEPS = H / 128 (7 for 896), WAVES x 4 sixteen-lane groups >= pairs per workgroup, RMAX = max down rows per workgroup, GW = gather waves.
// ONE_PER_CU: an 84 KB LDS reservation makes the hardware place at most one workgroup per CU (two fit by registers and real LDS use).
template <int STEPS, int WAVES, int RMAX, int GW, bool ONE_PER_CU = false>
__global__ __launch_bounds__(WAVES * 64) void mlp_pair_kernel(MlpPairArgs p) {
    constexpr int NT = WAVES * 64;
    __shared__ char reserve[ONE_PER_CU ? 84 * 1024 : 16];
    if (p.mode == 99) reserve[threadIdx.x] = 1;                  // never true: keeps the reservation alive
    constexpr int IMAX = 5120;                                   // LDS room for the gathered activations
    __shared__ __attribute__((aligned(16))) float xs[STEPS * 128];
    __shared__ __attribute__((aligned(16))) float act_s[IMAX];
    __shared__ float red[WAVES];
    __shared__ float part[WAVES][RMAX];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, grp = lane >> 4, sub = lane & 15;
    const int G = gridDim.x, b = blockIdx.x;
    const int pp = (p.I + G - 1) / G;                            // (gate, up) pairs per workgroup (19)
    const int unit = wave * 4 + grp;
    const int pair = b * pp + unit;
    const bool have_pair = unit < pp && pair < p.I;
    const int r0 = (int)((long long)b * p.H / G), r1 = (int)((long long)(b + 1) * p.H / G);     // down rows [r0, r1): 3 or 4
    long long t_entry = 0;
    if (p.stamps && tid == 0) t_entry = wall_clock64();

    // 0. the small L2-resident operands FIRST: vmcnt retires loads in issue order, so whatever is requested behind the weight stream is only
    //    usable once the whole stream has landed - requested ahead of it, x is normalised while the weights are still in flight
    const unsigned epoch = (unsigned)*p.epoch;
    float hres = 0.f;
    if (tid < RMAX && r0 + tid < r1) hres = p.h[r0 + tid];
    const bool have_x = tid * 4 < p.H;
    float4 xv = *reinterpret_cast<const float4*>(p.h + (have_x ? tid * 4 : 0)), gv = *reinterpret_cast<const float4*>(p.gamma + (have_x ? tid * 4 : 0));
    if (!have_x) xv = make_float4(0.f, 0.f, 0.f, 0.f);

    // 1. every weight byte of both phases
    u32x4 w[2][STEPS];
    {
        const int row0 = 2 * (have_pair ? pair : 0);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const bf16_t* wr = p.Wgu + (long long)(row0 + r) * p.H + sub * 8;
#pragma unroll
            for (int s = 0; s < STEPS; ++s) w[r][s] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wr + s * 128));
        }
    }
    const int ppr = p.I / 8;                                      // 16-byte pieces per down row (608)
    constexpr int PPL = 2;                                        // pieces per lane and row: ppr <= PPL * NT (host check)
    u32x4 wd[RMAX][PPL];
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
        const int row = min(r0 + r, p.H - 1);
#pragma unroll
        for (int i = 0; i < PPL; ++i) {
            const int pc = tid + NT * i;
            const bool ok = pc < ppr && r0 + r < r1;
            u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p.Wd + (long long)row * p.I + (ok ? pc : 0) * 8));
            if (!ok) v = (u32x4){0u, 0u, 0u, 0u};
            wd[r][i] = v;
        }
    }

    // 2. RMSNorm of x, shared through LDS (gemv_norm_kernel's prologue)
    {
        const bool have = have_x;
        const int k = have ? tid * 4 : 0;
        float ss = wave_sum(xv.x * xv.x + xv.y * xv.y + xv.z * xv.z + xv.w * xv.w);
        if (lane == 0) red[wave] = ss;
        __syncthreads();
        float tot = 0.f;
#pragma unroll
        for (int i = 0; i < WAVES; ++i) tot += red[i];
        const float rstd = rsqrtf(tot / (float)p.H + p.eps);
        if (have) *reinterpret_cast<float4*>(&xs[k]) = make_float4(xv.x * rstd * gv.x, xv.y * rstd * gv.y, xv.z * rstd * gv.z, xv.w * rstd * gv.w);
        __syncthreads();
    }
    // 3. gate / up dot products of this group's pair, SiLU(gate) * up, published as ONE granule
    {
        float acc[2] = {0.f, 0.f};
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            const float4 xa = *reinterpret_cast<const float4*>(&xs[s * 128 + sub * 8]), xb = *reinterpret_cast<const float4*>(&xs[s * 128 + sub * 8 + 4]);
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const u32x4 u = w[r][s];
                float a = acc[r];
                a += __uint_as_float(u[0] << 16) * xa.x;          a += __uint_as_float(u[0] & 0xffff0000u) * xa.y;
                a += __uint_as_float(u[1] << 16) * xa.z;          a += __uint_as_float(u[1] & 0xffff0000u) * xa.w;
                a += __uint_as_float(u[2] << 16) * xb.x;          a += __uint_as_float(u[2] & 0xffff0000u) * xb.y;
                a += __uint_as_float(u[3] << 16) * xb.z;          a += __uint_as_float(u[3] & 0xffff0000u) * xb.w;
                acc[r] = a;
            }
        }
        const float g = group16_sum(acc[0]), up = group16_sum(acc[1]);
        if (have_pair && sub == 0) {
            const float a = (g / (1.f + expf(-g))) * up;
            gran_store(p.gran + pair, ((u64_t)epoch << 32) | (u64_t)__float_as_uint(a));
        }
    }
    if (p.stamps && tid == 0) p.stamps[b * 8 + 1] = wall_clock64() - t_entry;
    // 4. gather: GW waves sweep 8 KB chunks (16 granules per lane) until every tag is this replay's epoch
    if (wave < GW) {
        const int nchunk = (p.I + 1023) / 1024;
        for (int c = wave; c < nchunk; c += GW) {                 // wave-uniform
            u64_t v[16];
            unsigned spins = 0;
            while (true) {
                bool ok = true;
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int idx = c * 1024 + k * 64 + lane;
                    v[k] = gran_load(p.gran + min(idx, p.I - 1));
                    ok &= (unsigned)(v[k] >> 32) == epoch || idx >= p.I;
                }
                if (__all(ok) || p.mode == 1) break;
                if (++spins > (1u << 17)) { if (lane == 0) atomicExch(p.fail, 1); break; }
                __builtin_amdgcn_s_sleep(2);
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int idx = c * 1024 + k * 64 + lane;
                if (idx < p.I) act_s[idx] = __uint_as_float((unsigned)v[k]);
            }
        }
    }
    __syncthreads();
    if (p.stamps && tid == 0) p.stamps[b * 8 + 2] = wall_clock64() - t_entry;
    // 5. down rows of this workgroup out of the registers requested at entry, + residual
    {
        float acc[RMAX];
#pragma unroll
        for (int r = 0; r < RMAX; ++r) acc[r] = 0.f;
#pragma unroll
        for (int i = 0; i < PPL; ++i) {
            const int pc = min(tid + NT * i, ppr - 1);
            const float4 xa = *reinterpret_cast<const float4*>(&act_s[pc * 8]), xb = *reinterpret_cast<const float4*>(&act_s[pc * 8 + 4]);
#pragma unroll
            for (int r = 0; r < RMAX; ++r) {
                const u32x4 u = wd[r][i];
                float a = acc[r];
                a += __uint_as_float(u[0] << 16) * xa.x;          a += __uint_as_float(u[0] & 0xffff0000u) * xa.y;
                a += __uint_as_float(u[1] << 16) * xa.z;          a += __uint_as_float(u[1] & 0xffff0000u) * xa.w;
                a += __uint_as_float(u[2] << 16) * xb.x;          a += __uint_as_float(u[2] & 0xffff0000u) * xb.y;
                a += __uint_as_float(u[3] << 16) * xb.z;          a += __uint_as_float(u[3] & 0xffff0000u) * xb.w;
                acc[r] = a;
            }
        }
#pragma unroll
        for (int r = 0; r < RMAX; ++r) acc[r] = wave_sum(acc[r]);
        if (lane == 0) {
#pragma unroll
            for (int r = 0; r < RMAX; ++r) part[wave][r] = acc[r];
        }
        __syncthreads();
        if (tid < RMAX && r0 + tid < r1) {
            float t = 0.f;
#pragma unroll
            for (int i = 0; i < WAVES; ++i) t += part[i][tid];
            p.h[r0 + tid] = hres + t;
        }
    }
    if (p.stamps && tid == 0) { p.stamps[b * 8 + 3] = wall_clock64() - t_entry; p.stamps[b * 8 + 0] = t_entry; }
}

}  // namespace cv
