// Dev tool (round 5): can the LAST workgroup of a group to arrive combine the group's partials inside the launch WITHOUT agent-scope fences?
// Round 2 measured the textbook form (partials, __threadfence(), arrival counter, __threadfence(), combine) at +50 % of a batched decode step: each
// release / acquire fence at agent scope writes back / invalidates an XCD's whole L2.  Here the partials themselves travel as RELAXED AGENT-SCOPE atomic
// stores / loads (sc1 accesses: coherent at the agent's coherence point by the memory model, no cache-wide maintenance), ordered against the arrival
// counter by "s_waitcnt vmcnt(0)" (the stores have been acknowledged) instead of a release fence.
//   A  two launches: producers store partials, a second launch combines                                    (what llm.hip does today)
//   B  one launch, relaxed agent-scope atomics + vmcnt(0) + arrival counter, last arriver combines         (the candidate)
//   C  one launch, plain stores + __threadfence() + counter + __threadfence()                              (round 2's form, for reference)
// Every variant is checked against the host's expectation over many launches with data that changes per launch (a stale read is a wrong sum).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <functional>

constexpr int PARTF = 448 + 64;       // floats per partial (7 heads x 64 + padding)

struct Args { const float* data; float* part; float* out; unsigned* ctr; int S; int iter; int same_xcd; int pairs; };

__device__ __forceinline__ void wg_ids(const Args& a, int& pair, int& slice) {
    const int id = blockIdx.x;
    if (a.same_xcd && (a.pairs & 7) == 0) { const int x = id & 7, q = id >> 3; pair = (q / a.S) * 8 + x; slice = q % a.S; }    // the S workgroups of a pair: ids equal mod 8
    else { pair = id / a.S; slice = id % a.S; }
}

// the "work": every thread pulls 6 float4 of a per-(pair, slice) region (24 KB per workgroup) and reduces them to its 2 floats of the partial
__device__ __forceinline__ float2 work(const Args& a, int pair, int slice) {
    const float4* d = reinterpret_cast<const float4*>(a.data) + ((long long)(pair * a.S + slice) * 6) * 256 + threadIdx.x;
    float4 v[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) v[i] = d[i * 256];
    float s0 = (float)a.iter, s1 = (float)(a.iter & 3);
#pragma unroll
    for (int i = 0; i < 6; ++i) { s0 += v[i].x + v[i].z; s1 += v[i].y + v[i].w; }
    return make_float2(s0, s1);
}

__global__ __launch_bounds__(256) void produce_kernel(Args a) {
    int pair, slice; wg_ids(a, pair, slice);
    const float2 p = work(a, pair, slice);
    *reinterpret_cast<float2*>(a.part + ((long long)pair * a.S + slice) * PARTF + 2 * threadIdx.x) = p;
}
__global__ __launch_bounds__(256) void combine_kernel(Args a) {
    const int pair = blockIdx.x;
    float2 acc = make_float2(0.f, 0.f);
    float2 v[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) v[s] = *reinterpret_cast<const float2*>(a.part + ((long long)pair * a.S + min(s, a.S - 1)) * PARTF + 2 * threadIdx.x);
#pragma unroll
    for (int s = 0; s < 8; ++s) if (s < a.S) { acc.x += v[s].x; acc.y += v[s].y; }
    *reinterpret_cast<float2*>(a.out + (long long)pair * PARTF + 2 * threadIdx.x) = acc;
}

template <int MODE>      // 1: relaxed agent-scope atomics + vmcnt(0)   2: plain + __threadfence()
__global__ __launch_bounds__(256) void fused_kernel(Args a) {
    __shared__ int last;
    int pair, slice; wg_ids(a, pair, slice);
    const float2 p = work(a, pair, slice);
    float* mine = a.part + ((long long)pair * a.S + slice) * PARTF + 2 * threadIdx.x;
    if (MODE == 1) {
        __hip_atomic_store(mine, p.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(mine + 1, p.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // this wave's partial stores have been acknowledged at the coherence point
    } else {
        *reinterpret_cast<float2*>(mine) = p;
        __threadfence();
    }
    __syncthreads();
    if (threadIdx.x == 0) last = (__hip_atomic_fetch_add(a.ctr + pair, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(a.S - 1));
    __syncthreads();
    if (!last) return;
    if (MODE == 2) __threadfence();
    float2 v[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const float* q = a.part + ((long long)pair * a.S + min(s, a.S - 1)) * PARTF + 2 * threadIdx.x;
        if (MODE == 1) { v[s].x = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); v[s].y = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        else v[s] = *reinterpret_cast<const float2*>(q);
    }
    float2 acc = make_float2(0.f, 0.f);
#pragma unroll
    for (int s = 0; s < 8; ++s) if (s < a.S) { acc.x += v[s].x; acc.y += v[s].y; }
    *reinterpret_cast<float2*>(a.out + (long long)pair * PARTF + 2 * threadIdx.x) = acc;
    if (threadIdx.x == 0) __hip_atomic_store(a.ctr + pair, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch (kernel boundaries order it)
}

__global__ void check_kernel(const float* out, const float* expect, int n, int iter, int S, unsigned* errors) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int pair = i / 512, e = i % 512;
    if (e >= 512) return;
    const float add = (e & 1) ? (float)(iter & 3) : (float)iter;
    (void)pair;
    const float want = expect[i] + (float)S * add;
    if (out[i] != want) atomicAdd(errors, 1u);
}

static float time_graph(const char* name, int n, const std::function<void(int, hipStream_t)>& launch) {
    hipStream_t s; (void)hipStreamCreate(&s);
    hipGraph_t g; hipGraphExec_t ge;
    (void)hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < n; ++i) launch(i, s);
    (void)hipStreamEndCapture(s, &g);
    (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int i = 0; i < 3; ++i) (void)hipGraphLaunch(ge, s);
    (void)hipStreamSynchronize(s);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int reps = 20;
    (void)hipEventRecord(e0, s);
    for (int i = 0; i < reps; ++i) (void)hipGraphLaunch(ge, s);
    (void)hipEventRecord(e1, s); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const float us = ms * 1e3f / (reps * n);
    printf("%-72s %7.2f us per step\n", name, us); fflush(stdout);
    (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g); (void)hipStreamDestroy(s);
    return us;
}

int main() {
    const int pairs = 64;
    for (int S : {4, 8}) {
        const size_t nd = (size_t)pairs * S * 6 * 256 * 4;
        std::vector<float> hd(nd);
        for (size_t i = 0; i < nd; ++i) hd[i] = (float)((i * 2654435761u >> 20) & 7);            // small integers: every sum is exact in fp32
        float *data, *part, *out, *expect; unsigned *ctr, *errors;
        (void)hipMalloc(&data, nd * 4); (void)hipMemcpy(data, hd.data(), nd * 4, hipMemcpyHostToDevice);
        (void)hipMalloc(&part, (size_t)pairs * S * PARTF * 4); (void)hipMemset(part, 0, (size_t)pairs * S * PARTF * 4);
        (void)hipMalloc(&out, (size_t)pairs * PARTF * 4); (void)hipMalloc(&expect, (size_t)pairs * PARTF * 4);
        (void)hipMalloc(&ctr, pairs * 4); (void)hipMemset(ctr, 0, pairs * 4);
        (void)hipMalloc(&errors, 4); (void)hipMemset(errors, 0, 4);
        // expectation without the per-launch addend: variant A at iter = 0
        Args a{data, part, out, nullptr, S, 0, 0, pairs};
        hipLaunchKernelGGL(produce_kernel, dim3(pairs * S), dim3(256), 0, 0, a);
        hipLaunchKernelGGL(combine_kernel, dim3(pairs), dim3(256), 0, 0, a);
        (void)hipMemcpy(expect, out, (size_t)pairs * PARTF * 4, hipMemcpyDeviceToDevice);
        (void)hipDeviceSynchronize();
        printf("---- %d pairs x %d slices (%d workgroups of 256 threads, 24 KB read per workgroup)\n", pairs, S, pairs * S);
        for (int same = 0; same < 2; ++same) {
            for (int mode = 0; mode < 3; ++mode) {
                // correctness: 3000 launches, the data term changes with the launch index
                (void)hipMemset(errors, 0, 4); (void)hipMemset(ctr, 0, pairs * 4);
                for (int it = 1; it <= 3000; ++it) {
                    Args b{data, part, out, ctr, S, it, same, pairs};
                    if (mode == 0) { hipLaunchKernelGGL(produce_kernel, dim3(pairs * S), dim3(256), 0, 0, b); hipLaunchKernelGGL(combine_kernel, dim3(pairs), dim3(256), 0, 0, b); }
                    else if (mode == 1) hipLaunchKernelGGL(fused_kernel<1>, dim3(pairs * S), dim3(256), 0, 0, b);
                    else hipLaunchKernelGGL(fused_kernel<2>, dim3(pairs * S), dim3(256), 0, 0, b);
                    hipLaunchKernelGGL(check_kernel, dim3((pairs * 512 + 255) / 256), dim3(256), 0, 0, out, expect, pairs * 512, it, S, errors);
                }
                unsigned herr = 0; (void)hipMemcpy(&herr, errors, 4, hipMemcpyDeviceToHost);
                char name[160];
                snprintf(name, sizeof name, "%s, %s: %u wrong elements in 3000 launches;", mode == 0 ? "A two launches" : mode == 1 ? "B relaxed agent atomics + vmcnt(0)" : "C __threadfence()",
                         same ? "slices of a pair on one XCD" : "slices spread over XCDs", herr);
                (void)hipMemset(ctr, 0, pairs * 4);
                time_graph(name, 96, [&](int i, hipStream_t s) {
                    Args b{data, part, out, ctr, S, i, same, pairs};
                    if (mode == 0) { hipLaunchKernelGGL(produce_kernel, dim3(pairs * S), dim3(256), 0, s, b); hipLaunchKernelGGL(combine_kernel, dim3(pairs), dim3(256), 0, s, b); }
                    else if (mode == 1) hipLaunchKernelGGL(fused_kernel<1>, dim3(pairs * S), dim3(256), 0, s, b);
                    else hipLaunchKernelGGL(fused_kernel<2>, dim3(pairs * S), dim3(256), 0, s, b);
                });
            }
        }
        (void)hipFree(data); (void)hipFree(part); (void)hipFree(out); (void)hipFree(expect); (void)hipFree(ctr); (void)hipFree(errors);
    }
    return 0;
}
