// Dev tool: can a batch-1 decode chain hide its kernel boundaries by launching dependent GEMVs EARLY?
//
// Mode A (baseline): the five weight-streaming kernels of a decode layer as one dependent chain in a hipGraph (what llm.hip does).
// Mode B ("early launch"): the same kernels alternate between two capture streams, so kernel k+1 is dispatched while kernel k runs.
//   Each wave first requests ALL of its weight rows (they do not depend on the predecessor), then waits for the predecessor's
//   completion counter (one relaxed agent-scope poller wave per workgroup), then reads x with sc1 loads, computes, stores y with sc1
//   (write-through) stores and arrives on a sharded counter.  The boundary + dispatch + first HBM round trip of kernel k+1 overlap
//   kernel k.  Protocol per MI355X_MICROARCH.md "Workgroup dispatch ... inter-workgroup visibility": {sc1 stores -> vmcnt(0) ->
//   relaxed agent atomic} on the producer, {relaxed poll -> sc1 loads} on the consumer.  Every spin is bounded.
// Prints us per layer for both modes and checks mode B's result bit-for-bit against mode A's on every replay.
#include <hip/hip_runtime.h>
#include <vector>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>

typedef unsigned short bf16_t;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int NSHARD = 8, SHARD_PAD = 16;          // 8 counters per kernel instance, 64 B apart

struct Args {
    const bf16_t* W; const float* x; float* y; const float* res; int N, K;
    int* wait_cnt; int wait_target; int* arrive_cnt; int flags; int* err;
};

__device__ __forceinline__ float group16_sum(float v) {
    v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
    return v;
}

// 256 threads.  SPLITK == 1: wave w, group g own ROWS rows each (16 * ROWS rows per workgroup), K = STEPS * 128.
// SPLITK == 4: the 4 waves split K (STEPS steps each) over the same 4 * ROWS rows.
template <int STEPS, int ROWS, int SPLITK>
__global__ __launch_bounds__(256) void gemv_flag(Args p) {
    __shared__ float part[4][4][ROWS];
    __shared__ int ok_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, grp = lane >> 4, sub = lane & 15;
    const int steps = p.K / 128;
    const int s0 = SPLITK == 1 ? 0 : wave * steps / 4, s1 = SPLITK == 1 ? steps : (wave + 1) * steps / 4;
    const int row0 = SPLITK == 1 ? ((blockIdx.x * 4 + wave) * 4 + grp) * ROWS : (blockIdx.x * 4 + grp) * ROWS;
    u32x4 w[ROWS][STEPS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const int row = min(row0 + r, p.N - 1);
        const bf16_t* wr = p.W + (long long)row * p.K + sub * 8;
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            const bool ok = s0 + s < s1;
            u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wr + (ok ? (s0 + s) : s0) * 128));
            if (!ok) t = (u32x4){0u, 0u, 0u, 0u};
            w[r][s] = t;
        }
    }
    if (p.flags && p.wait_target > 0) {
        if (wave == 0) {                                     // one poller wave per workgroup; lanes 0..7 read one shard each
            int spins = 0, sum = 0;
            do {
                int v = 0;
                if (lane < NSHARD) v = __hip_atomic_load(p.wait_cnt + lane * SHARD_PAD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4);
                sum = __shfl(v, 0);
                if (sum >= p.wait_target) break;
                __builtin_amdgcn_s_sleep(2);
            } while (++spins < (1 << 20));
            if (sum < p.wait_target && lane == 0) atomicAdd(p.err, 1);      // gave up: never hang the box
        }
        __syncthreads();
    }
    // x: sc1 loads in flag mode (the producer stored write-through; this CU's L1 / this XCD's L2 may hold last token's lines)
    float4 xa[STEPS], xb[STEPS];
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (unsigned)p.K * 4u, 0x00020000);
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        const bool ok = s0 + s < s1;
        const int so = ok ? (s0 + s) : s0;
        const int off = (so * 128 + sub * 8) * 4;
        u32x4 a, b;
        if (p.flags) { a = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16); b = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 16, 0, 16); }
        else { a = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0); b = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 16, 0, 0); }
        xa[s] = make_float4(__uint_as_float(a[0]), __uint_as_float(a[1]), __uint_as_float(a[2]), __uint_as_float(a[3]));
        xb[s] = make_float4(__uint_as_float(b[0]), __uint_as_float(b[1]), __uint_as_float(b[2]), __uint_as_float(b[3]));
        if (!ok) { xa[s] = make_float4(0.f, 0.f, 0.f, 0.f); xb[s] = xa[s]; }
    }
    float acc[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        float a = 0.f;
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            const u32x4 u = w[r][s];
            a += __uint_as_float(u[0] << 16) * xa[s].x; a += __uint_as_float(u[0] & 0xffff0000u) * xa[s].y;
            a += __uint_as_float(u[1] << 16) * xa[s].z; a += __uint_as_float(u[1] & 0xffff0000u) * xa[s].w;
            a += __uint_as_float(u[2] << 16) * xb[s].x; a += __uint_as_float(u[2] & 0xffff0000u) * xb[s].y;
            a += __uint_as_float(u[3] << 16) * xb[s].z; a += __uint_as_float(u[3] & 0xffff0000u) * xb[s].w;
        }
        acc[r] = group16_sum(a);
    }
    bool writer = sub == 0;
    if (SPLITK > 1) {
        if (sub == 0) {
#pragma unroll
            for (int r = 0; r < ROWS; ++r) part[wave][grp][r] = acc[r];
        }
        __syncthreads();
        writer = writer && wave == 0;
        if (writer) {
#pragma unroll
            for (int r = 0; r < ROWS; ++r) { float t = 0.f; for (int ww = 0; ww < 4; ++ww) t += part[ww][grp][r]; acc[r] = t; }
        }
    }
    if (writer) {
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int row = row0 + r;
            if (row < p.N) {
                float v = acc[r];
                if (p.res) v += (p.flags ? __hip_atomic_load(p.res + row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : p.res[row]) * 0.5f;
                v = v / (1.f + fabsf(v));                    // keeps the chain bounded over hundreds of kernels
                if (p.flags) __hip_atomic_store(p.y + row, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else p.y[row] = v;
            }
        }
    }
    if (p.flags) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wave's write-through stores have left
        __syncthreads();                                     // ... and every other wave's
        if (tid == 0) __hip_atomic_fetch_add(p.arrive_cnt + (blockIdx.x % NSHARD) * SHARD_PAD, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

struct KDesc { int kind; int N, K; int wgs; };

int main(int argc, char** argv) {
    const int NL = 24, H = 896, QKV = 1152, I = 4864, KPL = 5, NK = NL * KPL;
    const int reps = argc > 1 ? atoi(argv[1]) : 50;
    auto dmalloc = [](size_t b) { void* p; if (hipMalloc(&p, b) != hipSuccess) { printf("alloc failed\n"); exit(1); } (void)hipMemset(p, 0, b); return p; };
    auto fill_w = [&](size_t n, int K, unsigned seed) {
        std::vector<bf16_t> h(n); unsigned s = seed * 2654435761u + 12345u;
        const float sc = 1.7f / sqrtf((float)K);
        for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; const float v = ((int)(s >> 8) % 2001 - 1000) * 1e-3f * sc; unsigned u; memcpy(&u, &v, 4); h[i] = (bf16_t)(u >> 16); }
        bf16_t* d = (bf16_t*)dmalloc(n * 2); (void)hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice); return d;
    };
    std::vector<bf16_t*> wq(NL), wa(NL), wo(NL), wg(NL), wd(NL);
    for (int l = 0; l < NL; ++l) {
        wq[l] = fill_w((size_t)QKV * H, H, 5 * l); wa[l] = fill_w((size_t)H * QKV, QKV, 5 * l + 1); wo[l] = fill_w((size_t)H * H, H, 5 * l + 2);
        wg[l] = fill_w((size_t)2 * I * H, H, 5 * l + 3); wd[l] = fill_w((size_t)H * I, I, 5 * l + 4);
    }
    float* x = (float*)dmalloc(H * 4); float* q = (float*)dmalloc(QKV * 4); float* a = (float*)dmalloc(H * 4); float* h2 = (float*)dmalloc(H * 4);
    float* act = (float*)dmalloc(2 * I * 4);
    std::vector<float> hx(H); for (int i = 0; i < H; ++i) hx[i] = sinf(0.37f * i);
    int* cnt = (int*)dmalloc((size_t)(NK + 1) * NSHARD * SHARD_PAD * 4); int* err = (int*)dmalloc(4);
    auto cslot = [&](int k) { return cnt + (size_t)k * NSHARD * SHARD_PAD; };

    auto enqueue = [&](int k, hipStream_t s, int flags) {
        const int l = k / KPL, j = k % KPL;
        Args p{}; p.flags = flags; p.err = err; p.arrive_cnt = cslot(k); p.wait_cnt = k > 0 ? cslot(k - 1) : cslot(NK); p.wait_target = 0;
        auto wgs = [&](int jj) { return jj == 0 ? QKV / 16 : jj == 1 ? H / 16 : jj == 2 ? H / 16 : jj == 3 ? 2 * I / 32 : H / 4; };
        if (k > 0) p.wait_target = wgs((k - 1) % KPL);
        if (j == 0) { p.W = wq[l]; p.x = x; p.y = q; p.N = QKV; p.K = H; hipLaunchKernelGGL((gemv_flag<7, 1, 1>), dim3(wgs(0)), dim3(256), 0, s, p); }
        if (j == 1) { p.W = wa[l]; p.x = q; p.y = a; p.N = H; p.K = QKV; hipLaunchKernelGGL((gemv_flag<9, 1, 1>), dim3(wgs(1)), dim3(256), 0, s, p); }
        if (j == 2) { p.W = wo[l]; p.x = a; p.y = h2; p.res = x; p.N = H; p.K = H; hipLaunchKernelGGL((gemv_flag<7, 1, 1>), dim3(wgs(2)), dim3(256), 0, s, p); }
        if (j == 3) { p.W = wg[l]; p.x = h2; p.y = act; p.N = 2 * I; p.K = H; hipLaunchKernelGGL((gemv_flag<7, 2, 1>), dim3(wgs(3)), dim3(256), 0, s, p); }
        if (j == 4) { p.W = wd[l]; p.x = act; p.y = x; p.res = h2; p.N = H; p.K = I; hipLaunchKernelGGL((gemv_flag<10, 1, 4>), dim3(wgs(4)), dim3(256), 0, s, p); }
    };
    // NOTE: down reads only the first 4864 of act's 9728 entries - the probe measures the chain, not the model.

    hipStream_t s0, s1; (void)hipStreamCreate(&s0); (void)hipStreamCreate(&s1);
    hipEvent_t fork, join; (void)hipEventCreateWithFlags(&fork, hipEventDisableTiming); (void)hipEventCreateWithFlags(&join, hipEventDisableTiming);
    hipGraph_t gA, gB; hipGraphExec_t eA, eB;
    (void)hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal);
    for (int k = 0; k < NK; ++k) enqueue(k, s0, 0);
    (void)hipStreamEndCapture(s0, &gA); (void)hipGraphInstantiate(&eA, gA, nullptr, nullptr, 0);

    (void)hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal);
    (void)hipMemsetAsync(cnt, 0, (size_t)(NK + 1) * NSHARD * SHARD_PAD * 4, s0);
    (void)hipEventRecord(fork, s0); (void)hipStreamWaitEvent(s1, fork, 0);
    for (int k = 0; k < NK; ++k) enqueue(k, (k & 1) ? s1 : s0, 1);
    (void)hipEventRecord(join, s1); (void)hipStreamWaitEvent(s0, join, 0);
    (void)hipStreamEndCapture(s0, &gB);
    if (hipGraphInstantiate(&eB, gB, nullptr, nullptr, 0) != hipSuccess) { printf("graph B instantiate failed\n"); return 1; }

    std::vector<float> refx(H), gotx(H);
    auto run = [&](hipGraphExec_t e, int n, float* us_per_layer) {
        (void)hipMemcpy(x, hx.data(), H * 4, hipMemcpyHostToDevice);
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0, s0);
        for (int i = 0; i < n; ++i) (void)hipGraphLaunch(e, s0);
        (void)hipEventRecord(e1, s0);
        if (hipEventSynchronize(e1) != hipSuccess) { printf("sync failed\n"); exit(2); }
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); *us_per_layer = ms * 1e3f / (n * NL);
    };
    float usA, usB;
    run(eA, 3, &usA); run(eA, reps, &usA);
    (void)hipMemcpy(refx.data(), x, H * 4, hipMemcpyDeviceToHost);
    printf("mode A (one chain, kernel boundaries)          %7.2f us/layer  (%d kernels per layer)\n", usA, KPL); fflush(stdout);
    run(eB, 3, &usB); run(eB, reps, &usB);
    (void)hipMemcpy(gotx.data(), x, H * 4, hipMemcpyDeviceToHost);
    int herr = 0; (void)hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < H; ++i) bad += memcmp(&refx[i], &gotx[i], 4) != 0;
    printf("mode B (two chains, early launch + flags)      %7.2f us/layer  mismatching outputs %d / %d, spin give-ups %d\n", usB, bad, H, herr);
    // replay-by-replay check: any stale read shows up as a mismatch against mode A's state after the same number of replays
    int bad_runs = 0;
    for (int r = 0; r < 20; ++r) {
        float t; run(eA, 1 + r % 3, &t); (void)hipMemcpy(refx.data(), x, H * 4, hipMemcpyDeviceToHost);
        run(eB, 1 + r % 3, &t); (void)hipMemcpy(gotx.data(), x, H * 4, hipMemcpyDeviceToHost);
        bad_runs += memcmp(refx.data(), gotx.data(), H * 4) != 0;
    }
    (void)hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
    printf("replay check: %d / 20 runs differ, total give-ups %d, |x| sample %.5f %.5f\n", bad_runs, herr, refx[0], refx[500]);
    return 0;
}
