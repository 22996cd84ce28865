// LLM (Qwen2 backbone of cosyvoice/llm/llm.py Qwen2LM) decode-path kernels for gfx950.
// Batch-1 autoregressive decode is HBM-bandwidth bound (727.6 MB of bf16 weights per token): the GEMV streams every
// weight row once with 16B/lane coalesced loads, activations are fp32 in LDS, accumulation is fp32 in a fixed order.
// Every kernel reads the loop state (KV length, "done" flag, last token) from device memory so that one captured
// hipGraph can be replayed for every token without host involvement.
#pragma once
#include "common.h"

namespace cv {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct DecodeState {       // lives in device memory (one per cv_llm handle)
    int pos;               // number of positions already in the KV cache
    int step;              // index i of the reference loop `for i in range(max_len)` (llm/llm.py:538)
    int done;              // set when a stop token was sampled or max_len reached
    int n_tokens;          // tokens emitted so far
    int last_token;        // last emitted token (input of the next backbone step)
    int stop_token;        // the id that set `done` (eos / fill / any other special id), -1 while running; inference_bistream needs to
                           // tell a fill token (more text is due) from eos
    int pad[2];
};

// ---------------------------------------------------------------------------------------------------------------
// Next-kernel weight prefetch (round 3).  A decode layer is a chain of five dependent launches; three of them (qkv, attention, o_proj) move
// 3.6 MB in ~11 us - they are bound by the launch boundary and one memory round trip, and HBM idles while they run - and the two that follow
// (gate / up 17.4 MB, down 8.7 MB) then wait for HBM.  A forked graph branch that warms L2 costs more than it saves (+21 us per layer for the
// fork / join edges, profiles/r2_batch_decode_ab.txt); extra WORKGROUPS appended to the short kernels cost no boundary at all: workgroups
// [first, first + groups * stride) of the host launch only read (default cache policy: the lines are left in the XCD's L2 and in the memory-side
// cache) the bytes a LATER kernel of the chain will stream, and exit.  Consumer workgroup b of that kernel reads the contiguous range
// [b * cons_bytes, (b + 1) * cons_bytes); it is fetched by host workgroups first + j * stride + b, j = 0 .. groups - 1 (`first` and `stride` are
// multiples of 8: with workgroups dealt round-robin to the 8 XCDs, fetcher and consumer share an L2).  Purely a hint: results do not depend on it.
// ---------------------------------------------------------------------------------------------------------------
struct PrefetchArgs {
    const char* p = nullptr;          // start of the region (null = off)
    long long bytes = 0;              // its size (loads are clamped to it)
    int cons_bytes = 0, n_cons = 0;   // bytes per consumer workgroup, number of consumer workgroups
    int stride = 0, first = 0;        // round_up(n_cons, 8); index of the first prefetch workgroup in the host launch (multiple of 8)
    int shift = 0;                    // dev knob: fetch for consumer b + shift instead (breaks the XCD match on purpose, A/B runs)
    unsigned* sink = nullptr;         // never written in practice: keeps the loads alive
};
// U wave-loads (64 lanes x 16 B = 1 KB each) per wave; `wave` of `nw` waves in the workgroup, prefetch-workgroup index pw (0-based)
template <int U>
__device__ __forceinline__ void prefetch_role(const PrefetchArgs& f, int pw, int wave, int nw, int lane) {
    const int j = pw / f.stride;
    int b = pw % f.stride;
    if (b >= f.n_cons) return;
    b = (b + f.shift) % f.n_cons;
    const long long base = (long long)b * f.cons_bytes;
    const int kb0 = (j * nw + wave) * U;                                   // first 1 KB unit of this wave inside the consumer's range
    if (kb0 * 1024 >= f.cons_bytes) return;
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        long long o = base + min((kb0 + u) * 1024 + lane * 16, f.cons_bytes - 16);
        o = o < f.bytes - 16 ? o : f.bytes - 16;
        v[u] = *reinterpret_cast<const u32x4*>(f.p + o);
    }
    unsigned a = 0u;
#pragma unroll
    for (int u = 0; u < U; ++u) a |= v[u][0] ^ v[u][3];
    if (a == 0x9e3779b9u && f.sink) f.sink[0] = a;                         // bf16 weight words never form this pattern in all lanes' OR; a hit would only touch the sink
}
inline int prefetch_groups(int cons_bytes, int nw, int U) { return ((cons_bytes + 1023) / 1024 + nw * U - 1) / (nw * U); }

// ---------------------------------------------------------------------------------------------------------------
// y[n] = epi( sum_k W[n][k] * xn[k] ),  W bf16 [N][K] row-major, K % 128 == 0.
//   xn = x                      (gamma == nullptr)
//   xn = rmsnorm(x) * gamma     (Qwen2RMSNorm fused as a prologue: every workgroup recomputes the 896-wide norm)
// A 16-lane group owns ROWS consecutive output rows: per step the group reads 256 contiguous bytes of a row (16B per lane).
// A workgroup = WAVES waves; wave w covers k-steps [w*S/WAVES, (w+1)*S/WAVES) of the same 4*ROWS rows (split-K inside the
// workgroup, combined through LDS in fixed order -> deterministic).
//   mode 0: y[n] = acc + bias[n] (+ res[n])
//   mode 1: rows are interleaved (gate_j, up_j); y[j] = silu(gate_j) * up_j          (Qwen2MLP)
// ---------------------------------------------------------------------------------------------------------------
struct GemvArgs {
    const bf16_t* W; const float* bias; const float* x; float* y; int N, K;
    const float* gamma; float eps; const float* res; int mode; const DecodeState* st;
    const float* part = nullptr;   // NSP > 0: x is the combination of NSP split-attention partials, [K/64][NSP][ATTN_PART] (see attn_decode_kernel)
    // qkv_attn_kernel leaves the NEW token out of its partials (its k / v are computed by other workgroups of the same launch); the merge adds
    // it as one more partial: score = q_new . k_new / 8, value = v_new.  qnew [K/64][64], knew / vnew [kv heads][64]; null = not used.
    const float* qnew = nullptr; const float* knew = nullptr; const float* vnew = nullptr; int kv_group = 1;
    DecodeState* advance = nullptr;   // last GEMV of a backbone step: also advances the KV length (no kernel of the step reads `pos` after it)
    PrefetchArgs pf;                  // workgroups >= pf.first of the launch only prefetch a later kernel's weights (prefetch_role)
    // gemv_norm_kernel only (attn_oproj_kernel upstream): the input is x + sum_{j < n_opart} opart[j][K] (per-head o_proj contributions added to
    // the residual in a fixed order); workgroup 0 also stores that sum to x_out (the residual the down projection adds to).
    const float* opart = nullptr; int n_opart = 0; float* x_out = nullptr;
};

constexpr int ATTN_PART = 68;      // 64 unnormalised numerators + running max + denominator (+2 pad: rows stay 16-byte aligned)


// Latency structure (batch-1 decode is a chain of ~125 short kernels, each bounded by ONE memory round trip if written so):
// every lane first issues ALL of its weight loads (ROWS x STEPS x 16 B, non-temporal: each byte is read once per token),
// then its slice of x (and gamma) straight from L2 into registers — no LDS staging, no barrier before the weights are in
// flight.  RMSNorm statistics come from a 16-lane shuffle reduction (each group covers the whole row when WAVES == 1).
template <int STEPS, int ROWS, int WAVES, int NSP = 0>
__global__ __launch_bounds__(WAVES * 64) void gemv_kernel(GemvArgs p) {
    __shared__ float part[WAVES][4][ROWS];
    // no `done` test here: it would put a dependent load in front of the weight stream; a finished request simply recomputes
    // into buffers nobody reads (sample / embed / advance are the kernels that honour `done`).
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, grp = lane >> 4, sub = lane & 15;
    if constexpr (NSP > 0) {                                 // o_proj hosts the prefetch of the down projection's weights
        if (p.pf.p && (int)blockIdx.x >= p.pf.first) { prefetch_role<10>(p.pf, blockIdx.x - p.pf.first, wave, WAVES, lane); return; }
    }
    if (p.advance && blockIdx.x == 0 && tid == 0 && !p.advance->done) p.advance->pos += 1;
    const int steps = p.K / 128;
    const int s0 = wave * steps / WAVES, s1 = (wave + 1) * steps / WAVES;
    const int row0 = (blockIdx.x * 4 + grp) * ROWS;

    // NSP > 0 (o_proj over split-attention partials): the partials are requested BEFORE the weight rows (round 3).  vmcnt retires in issue order, so
    // behind the weight stream the merge below could only start once the workgroup's last weight byte had landed; ahead of it, the merge and its
    // barrier run while the weights are in flight.  Unconditional, clamped loads (a load in a divergent branch costs the compiler its vmcnt count).
    constexpr int NQ = NSP > 0 ? NSP : 1;
    float4 pa[NQ]; float2 ml[NQ];
    if constexpr (NSP > 0) {
        const int t4 = min(tid, p.K / 4 - 1);                 // thread t owns dims 4t .. 4t+3 of x (head t / 16)
        const float* ph = p.part + (long long)(t4 >> 4) * NSP * ATTN_PART;
#pragma unroll
        for (int q = 0; q < NSP; ++q) {
            pa[q] = *reinterpret_cast<const float4*>(ph + q * ATTN_PART + (t4 & 15) * 4);
            ml[q] = *reinterpret_cast<const float2*>(ph + q * ATTN_PART + 64);
        }
    }
    u32x4 w[ROWS][STEPS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const int row = min(row0 + r, p.N - 1);              // clamp: the reductions below are wave collectives
        const bf16_t* wr = p.W + (long long)row * p.K + sub * 8;
#pragma unroll
        for (int s = 0; s < STEPS; ++s)
        {
            const bool ok = s0 + s < s1;
            u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wr + (ok ? (s0 + s) : s0) * 128));
            if (!ok) t = (u32x4){0u, 0u, 0u, 0u};
            w[r][s] = t;
        }
    }
    float4 xa[STEPS], xb[STEPS];
    if constexpr (NSP == 0) {
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            const bool ok = s0 + s < s1;
            const int so = ok ? (s0 + s) : s0;
            xa[s] = *reinterpret_cast<const float4*>(p.x + so * 128 + sub * 8);
            xb[s] = *reinterpret_cast<const float4*>(p.x + so * 128 + sub * 8 + 4);
            if (!ok) { xa[s] = make_float4(0.f, 0.f, 0.f, 0.f); xb[s] = xa[s]; }
        }
    } else {
        // x = softmax-combination of the NSP partial attention results of each head (flash-decoding merge, fixed order).  The
        // workgroup merges ONCE: thread t owns dims 4t..4t+3 (head t / 16), i.e. NSP float4 + NSP (max, denominator) pairs per
        // thread, and publishes x through LDS - the 4 row-groups of every wave would otherwise each fetch the same partials
        // (64 load instructions per lane measured +2 us on the o_proj launch: address-unit bound, not bandwidth bound).
        static_assert(WAVES == 4, "partial-combine prologue: 256 threads cover K <= 1024");
        __shared__ __attribute__((aligned(16))) float xs[1024];
        // the new token's own (score, value) when the attention left it out (qkv_attn_kernel): 16 threads of a head share the dot product
        // (computed by every thread, clamped, so that the cross-lane reduction runs in uniform control flow)
        float s_new = 0.f; float4 vn = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.qnew) {
            const int hh = min(tid >> 4, p.K / 64 - 1), g = hh / p.kv_group, d0 = (tid & 15) * 4;
            const float4 qn = *reinterpret_cast<const float4*>(p.qnew + hh * 64 + d0), kn = *reinterpret_cast<const float4*>(p.knew + g * 64 + d0);
            vn = *reinterpret_cast<const float4*>(p.vnew + g * 64 + d0);
            s_new = group16_sum(qn.x * kn.x + qn.y * kn.y + qn.z * kn.z + qn.w * kn.w) * 0.125f;
        }
        if (tid * 4 < p.K) {
            float M = p.qnew ? s_new : ml[0].x;
#pragma unroll
            for (int q = 0; q < NSP; ++q) if (ml[q].y > 0.f || !p.qnew) M = fmaxf(M, ml[q].x);
            float den = 0.f; float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int q = 0; q < NSP; ++q) {
                const float w = (ml[q].y > 0.f) ? expf(ml[q].x - M) : 0.f;       // an empty slice carries l = 0
                den += w * ml[q].y;
                a.x += w * pa[q].x; a.y += w * pa[q].y; a.z += w * pa[q].z; a.w += w * pa[q].w;
            }
            if (p.qnew) { const float w = expf(s_new - M); den += w; a.x += w * vn.x; a.y += w * vn.y; a.z += w * vn.z; a.w += w * vn.w; }
            const float inv = 1.f / den;
            *reinterpret_cast<float4*>(&xs[tid * 4]) = make_float4(a.x * inv, a.y * inv, a.z * inv, a.w * inv);
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            const bool ok = s0 + s < s1;
            const int so = ok ? (s0 + s) : s0;
            xa[s] = *reinterpret_cast<const float4*>(&xs[so * 128 + sub * 8]);
            xb[s] = *reinterpret_cast<const float4*>(&xs[so * 128 + sub * 8 + 4]);
            if (!ok) { xa[s] = make_float4(0.f, 0.f, 0.f, 0.f); xb[s] = xa[s]; }
        }
    }
    if (p.gamma) {                                           // fused Qwen2RMSNorm (host guarantees WAVES == 1 here)
        // gamma is requested BEFORE the reduction so its round trip overlaps the weight stream instead of following the shuffles
        float4 ga[STEPS], gb[STEPS];
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            const int so = (s0 + s < s1) ? (s0 + s) : s0;
            ga[s] = *reinterpret_cast<const float4*>(p.gamma + so * 128 + sub * 8);
            gb[s] = *reinterpret_cast<const float4*>(p.gamma + so * 128 + sub * 8 + 4);
        }
        float ss = 0.f;
#pragma unroll
        for (int s = 0; s < STEPS; ++s)
            ss += xa[s].x * xa[s].x + xa[s].y * xa[s].y + xa[s].z * xa[s].z + xa[s].w * xa[s].w +
                  xb[s].x * xb[s].x + xb[s].y * xb[s].y + xb[s].z * xb[s].z + xb[s].w * xb[s].w;
        ss = group16_sum(ss);
        const float rstd = rsqrtf(ss / (float)p.K + p.eps);
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            xa[s].x = xa[s].x * rstd * ga[s].x; xa[s].y = xa[s].y * rstd * ga[s].y; xa[s].z = xa[s].z * rstd * ga[s].z; xa[s].w = xa[s].w * rstd * ga[s].w;
            xb[s].x = xb[s].x * rstd * gb[s].x; xb[s].y = xb[s].y * rstd * gb[s].y; xb[s].z = xb[s].z * rstd * gb[s].z; xb[s].w = xb[s].w * rstd * gb[s].w;
        }
    }
    float acc[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        float a = 0.f;
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            const u32x4 u = w[r][s];
            a += __uint_as_float(u[0] << 16) * xa[s].x;          a += __uint_as_float(u[0] & 0xffff0000u) * xa[s].y;
            a += __uint_as_float(u[1] << 16) * xa[s].z;          a += __uint_as_float(u[1] & 0xffff0000u) * xa[s].w;
            a += __uint_as_float(u[2] << 16) * xb[s].x;          a += __uint_as_float(u[2] & 0xffff0000u) * xb[s].y;
            a += __uint_as_float(u[3] << 16) * xb[s].z;          a += __uint_as_float(u[3] & 0xffff0000u) * xb[s].w;
        }
        acc[r] = group16_sum(a);
    }
    if (WAVES > 1) {                                         // split-K inside the workgroup, combined in fixed order
        if (sub == 0) {
#pragma unroll
            for (int r = 0; r < ROWS; ++r) part[wave][grp][r] = acc[r];
        }
        __syncthreads();
        if (wave != 0) return;
#pragma unroll
        for (int r = 0; r < ROWS; ++r) { float t = 0.f; for (int ww = 0; ww < WAVES; ++ww) t += part[ww][grp][r]; acc[r] = t; }
    }
    if (sub != 0) return;
    if (p.mode == 1) {                                       // ROWS == 2: (gate_j, up_j)
        const int j = blockIdx.x * 4 + grp;
        if (row0 + 1 < p.N) { const float g = acc[0]; p.y[j] = (g / (1.f + expf(-g))) * acc[ROWS - 1]; }
    } else {
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int row = row0 + r;
            if (row < p.N) {
                float v = acc[r];
                if (p.bias) v += p.bias[row];
                if (p.res) v += p.res[row];
                p.y[row] = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The RMSNorm-prologue GEMVs (qkv, gate / up, head: K = hidden <= 896) with the normalised input SHARED by the 4 waves of a workgroup.
// In gemv_kernel every wave fetches x AND gamma for itself: 2 x 3.5 KB of L2 reads against 7 - 14 KB of weights - a third to a half of a
// CU's load instructions went to re-reading the same two vectors (a CU ingests a few tens of bytes per cycle, the binding resource of
// these kernels).  Here each wave still requests all of its weight rows first; then 224 threads fetch x / gamma once (float4 each), the
// sum of squares is reduced over the workgroup in a fixed order, x * rstd * gamma is parked in LDS (3.5 KB) and every lane reads its 14
// float4 from there.  The weight loads stay in flight across both barriers.  Same row -> lane mapping, same FMA order as gemv_kernel.
// ---------------------------------------------------------------------------------------------------------------
// WAVES = 5 for gate / up: 4864 row pairs over 20 groups per workgroup = 244 workgroups, one per CU in a single balanced round (with 4 waves
// the 304 workgroups leave 48 of the 256 CUs with two each - the tail of the launch).
// XFIRST (round 3): x and gamma are requested BEFORE the weight rows.  vmcnt retires loads in issue order, so with x behind the weight stream
// (round 2) the normalisation - and both of its barriers - could only start once the workgroup's last weight byte had landed; requested ahead of
// the stream (two L2 hits), x is normalised and parked in LDS while the weights are still in flight and the FMAs start on the first row that lands.
template <int STEPS, int ROWS, int WAVES = 4, bool XFIRST = true, bool OPART = false>
__global__ __launch_bounds__(WAVES * 64) void gemv_norm_kernel(GemvArgs p) {
    __shared__ __attribute__((aligned(16))) float xs[STEPS * 128];
    __shared__ float red[WAVES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, grp = lane >> 4, sub = lane & 15;
    if constexpr (ROWS == 1) {                               // qkv (and head) can host a prefetch of later weights
        if (p.pf.p && (int)blockIdx.x >= p.pf.first) { prefetch_role<9>(p.pf, blockIdx.x - p.pf.first, wave, WAVES, lane); return; }
    }
    const int steps = p.K / 128;
    const int unit = (blockIdx.x * WAVES + wave) * 4 + grp;  // 16-lane group index: ROWS consecutive rows
    const int row0 = unit * ROWS;
    const bool have = tid * 4 < p.K;
    float4 xv, gv;
    if constexpr (XFIRST) {
        xv = *reinterpret_cast<const float4*>(p.x + (have ? tid * 4 : 0)); gv = *reinterpret_cast<const float4*>(p.gamma + (have ? tid * 4 : 0));
    }
    constexpr int MAXP = 16;                                 // per-head o_proj contributions (attn_oproj_kernel), requested ahead of the weight stream like x
    float4 op[OPART ? MAXP : 1];
    if constexpr (OPART) {
        static_assert(XFIRST, "the o_proj contributions are requested with x");
#pragma unroll
        for (int j = 0; j < MAXP; ++j) op[j] = *reinterpret_cast<const float4*>(p.opart + (long long)min(j, p.n_opart - 1) * p.K + (have ? tid * 4 : 0));
    }

    u32x4 w[ROWS][STEPS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const int row = min(row0 + r, p.N - 1);
        const bf16_t* wr = p.W + (long long)row * p.K + sub * 8;
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            const bool ok = s < steps;
            u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wr + (ok ? s : 0) * 128));
            if (!ok) t = (u32x4){0u, 0u, 0u, 0u};
            w[r][s] = t;
        }
    }
    {
        const int k = have ? tid * 4 : 0;
        if constexpr (!XFIRST) { xv = *reinterpret_cast<const float4*>(p.x + k); gv = *reinterpret_cast<const float4*>(p.gamma + k); }
        if constexpr (OPART) {
#pragma unroll
            for (int j = 0; j < MAXP; ++j) if (j < p.n_opart) { xv.x += op[j].x; xv.y += op[j].y; xv.z += op[j].z; xv.w += op[j].w; }
            if (blockIdx.x == 0 && have && p.x_out) *reinterpret_cast<float4*>(p.x_out + k) = xv;
        }
        if (!have) xv = make_float4(0.f, 0.f, 0.f, 0.f);
        float ss = wave_sum(xv.x * xv.x + xv.y * xv.y + xv.z * xv.z + xv.w * xv.w);
        if (lane == 0) red[wave] = ss;
        __syncthreads();
        ss = (red[0] + red[1]) + (red[2] + red[3]);            // K <= 896: the elements live in waves 0..3 (a fifth wave holds zeros)
        const float rstd = rsqrtf(ss / (float)p.K + p.eps);
        if (have) *reinterpret_cast<float4*>(&xs[k]) = make_float4(xv.x * rstd * gv.x, xv.y * rstd * gv.y, xv.z * rstd * gv.z, xv.w * rstd * gv.w);
        __syncthreads();
    }
    float acc[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) acc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        if (s < steps) {
            const float4 xa = *reinterpret_cast<const float4*>(&xs[s * 128 + sub * 8]), xb = *reinterpret_cast<const float4*>(&xs[s * 128 + sub * 8 + 4]);
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
                const u32x4 u = w[r][s];
                float a = acc[r];
                a += __uint_as_float(u[0] << 16) * xa.x;          a += __uint_as_float(u[0] & 0xffff0000u) * xa.y;
                a += __uint_as_float(u[1] << 16) * xa.z;          a += __uint_as_float(u[1] & 0xffff0000u) * xa.w;
                a += __uint_as_float(u[2] << 16) * xb.x;          a += __uint_as_float(u[2] & 0xffff0000u) * xb.y;
                a += __uint_as_float(u[3] << 16) * xb.z;          a += __uint_as_float(u[3] & 0xffff0000u) * xb.w;
                acc[r] = a;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) acc[r] = group16_sum(acc[r]);
    if (sub != 0) return;
    if (p.mode == 1) {                                       // ROWS == 2: (gate_j, up_j)
        if (row0 + 1 < p.N) { const float g = acc[0]; p.y[unit] = (g / (1.f + expf(-g))) * acc[ROWS - 1]; }
    } else {
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int row = row0 + r;
            if (row < p.N) {
                float v = acc[r];
                if (p.bias) v += p.bias[row];
                if (p.res) v += p.res[row];
                p.y[row] = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Decode attention for one new position (GQA, head_dim 64), fused with rotate-half RoPE on q/k and the KV-cache append.
// qkv = [q(H*64) | k(Hkv*64) | v(Hkv*64)] raw projections (+bias) of the new token.
// cache layout: K,V [Hkv][max_len][64] fp32.  rope table: cos/sin [max_len][32].
// ---------------------------------------------------------------------------------------------------------------
struct AttnDecodeArgs {
    const float* qkv; float* kcache; float* vcache; const float* rope_cos; const float* rope_sin;
    int heads, kv_heads, max_len; const DecodeState* st;
    float* part; int nsplit;      // wave (h, s) covers the s-th contiguous slice of the keys and writes part[h][s][ATTN_PART]
    PrefetchArgs pf;              // workgroups >= pf.first only prefetch a later kernel's weights (prefetch_role)
};

// ONE WAVE per (query head, key slice) - the structure the decode GEMVs proved on this chip: single-wave workgroups, no LDS, no
// barrier, every load of a pass requested before anything is consumed.  A 16-lane group covers one key row with float4 loads,
// the 4 groups of the wave take 4 consecutive keys, NS = 12 slots -> 48 keys per pass (the 381-key context of the benchmark
// utterance over 8 slices is exactly one pass).  q and the new k get their rotate-half RoPE in registers straight from the
// qkv vector; scores stay in registers; softmax is online across passes with shuffle-only reductions.  Each wave leaves an
// un-normalised partial (64 numerators, running max, denominator); the o_proj GEMV merges the nsplit partials of every head
// in its prologue (gemv_kernel<..., NSP>): no atomics, no extra launch, fixed summation order.
// (A workgroup-per-head version with the K/V slice in 2 x 24 float4 registers per lane and LDS reductions measured 14-17 us per
// launch on MI355X - register-array spills and ~60 full vmcnt waits - against 2.8-7 us for the GEMVs around it.)
static __global__ __launch_bounds__(64) void attn_decode_kernel(AttnDecodeArgs p) {
    constexpr int NS = 12, PASS = 4 * NS;
    const int lane = threadIdx.x, sub = lane & 15, grp = lane >> 4;
    if (p.pf.p && (int)blockIdx.x >= p.pf.first) { prefetch_role<14>(p.pf, blockIdx.x - p.pf.first, 0, 1, lane); return; }
    if ((int)blockIdx.x >= p.heads * p.nsplit) return;                     // padding up to pf.first
    const int h = blockIdx.x / p.nsplit, sp = blockIdx.x % p.nsplit, gsz = p.heads / p.kv_heads, g = h / gsz;
    const int pos = p.st->pos;                       // the new token sits at index `pos`
    const int L = pos + 1;
    const int per = ((L + p.nsplit * 4 - 1) / (p.nsplit * 4)) * 4;           // keys per slice (multiple of 4)
    const int kb = sp * per, ke = min(L, kb + per);
    const float* kc = p.kcache + (long long)g * p.max_len * 64;
    const float* vc = p.vcache + (long long)g * p.max_len * 64;
    float4 k4[NS], v4[NS];
    auto load_pass = [&](int base) {
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) {
            const int j = base + sl * 4 + grp;
            const long long o = (long long)((j < pos && j < ke) ? j : 0) * 64 + sub * 4;      // unconditional, clamped
            k4[sl] = *reinterpret_cast<const float4*>(kc + o);
            v4[sl] = *reinterpret_cast<const float4*>(vc + o);
        }
    };
    load_pass(kb);
    // rotate-half RoPE in registers: lane dims d = sub*4 .. +3, partner dims (d + 32) % 64, sign - for d < 32
    const float* qraw = p.qkv + h * 64;
    const float* kq = p.qkv + p.heads * 64 + g * 64;
    const float* vq = p.qkv + (p.heads + p.kv_heads) * 64 + g * 64;
    const int d0 = sub * 4, dp = (d0 + 32) & 63;
    const float4 c4 = *reinterpret_cast<const float4*>(p.rope_cos + pos * 32 + (d0 & 31));
    const float4 s4 = *reinterpret_cast<const float4*>(p.rope_sin + pos * 32 + (d0 & 31));
    const float4 qa = *reinterpret_cast<const float4*>(qraw + d0), qb = *reinterpret_cast<const float4*>(qraw + dp);
    const float4 ka = *reinterpret_cast<const float4*>(kq + d0), kp = *reinterpret_cast<const float4*>(kq + dp);
    const float4 vn4 = *reinterpret_cast<const float4*>(vq + d0);
    const float sg = d0 < 32 ? -1.f : 1.f;
    const float4 q4 = make_float4(qa.x * c4.x + sg * qb.x * s4.x, qa.y * c4.y + sg * qb.y * s4.y, qa.z * c4.z + sg * qb.z * s4.z, qa.w * c4.w + sg * qb.w * s4.w);
    const float4 kn4 = make_float4(ka.x * c4.x + sg * kp.x * s4.x, ka.y * c4.y + sg * kp.y * s4.y, ka.z * c4.z + sg * kp.z * s4.z, ka.w * c4.w + sg * kp.w * s4.w);
    if (!p.st->done && sp == 0 && h % gsz == 0 && grp == 0) {               // KV-cache append: one wave per kv head
        *reinterpret_cast<float4*>(p.kcache + ((long long)g * p.max_len + pos) * 64 + d0) = kn4;
        *reinterpret_cast<float4*>(p.vcache + ((long long)g * p.max_len + pos) * 64 + d0) = vn4;
    }
    const float NEG = -__builtin_huge_valf();
    float m_run = NEG, l_run = 0.f;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int base = kb; base < ke; base += PASS) {   // wave-uniform
        if (base != kb) load_pass(base);
        float sc[NS];
        float mt = NEG;
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) {
            const int j = base + sl * 4 + grp;
            const float4 kk = (j == pos) ? kn4 : k4[sl];
            float a = q4.x * kk.x + q4.y * kk.y + q4.z * kk.z + q4.w * kk.w;
            a = group16_sum(a) * 0.125f;
            sc[sl] = j < ke ? a : NEG;
            mt = fmaxf(mt, sc[sl]);
        }
        mt = fmaxf(mt, __shfl_xor(mt, 16)); mt = fmaxf(mt, __shfl_xor(mt, 32));
        const float m_new = fmaxf(m_run, mt);        // finite: the pass holds at least one key
        const float scale = (m_run == NEG) ? 0.f : expf(m_run - m_new);
        acc.x *= scale; acc.y *= scale; acc.z *= scale; acc.w *= scale;
        float lt = 0.f;
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) {
            const int j = base + sl * 4 + grp;
            const float e = (sc[sl] == NEG) ? 0.f : expf(sc[sl] - m_new);
            const float4 vv = (j == pos) ? vn4 : v4[sl];
            acc.x += e * vv.x; acc.y += e * vv.y; acc.z += e * vv.z; acc.w += e * vv.w;
            lt += e;
        }
        l_run = l_run * scale + lt;                 // per-group partial (identical on the 16 lanes of a group)
        m_run = m_new;
    }
    acc.x += __shfl_xor(acc.x, 16); acc.y += __shfl_xor(acc.y, 16); acc.z += __shfl_xor(acc.z, 16); acc.w += __shfl_xor(acc.w, 16);
    acc.x += __shfl_xor(acc.x, 32); acc.y += __shfl_xor(acc.y, 32); acc.z += __shfl_xor(acc.z, 32); acc.w += __shfl_xor(acc.w, 32);
    l_run += __shfl_xor(l_run, 16); l_run += __shfl_xor(l_run, 32);
    float* pr = p.part + (long long)blockIdx.x * ATTN_PART;
    if (grp == 0) *reinterpret_cast<float4*>(pr + d0) = acc;
    if (lane == 0) { pr[64] = (l_run > 0.f) ? m_run : 0.f; pr[65] = l_run; }
}

// ---------------------------------------------------------------------------------------------------------------
// Decode attention + o_proj in ONE launch (round 3): the o_proj GEMV split over K by HEAD instead of over rows only.
//   y = h + Wo . a,  a = [a_0 .. a_{H-1}]   =>   y = h + sum_h  Wo[:, 64 h : 64 h + 64] . a_h
// Workgroup (head h, row block rb) runs the whole attention of head h - NW waves, wave w over the w-th key slice exactly like
// attn_decode_kernel's single wave (RoPE in registers, online softmax, KV append by slice 0), merged through LDS - and then multiplies its
// rows of Wo's 64-column slice of head h with a_h: an un-summed per-head contribution  opart[h][row].  The H contributions are added to the
// residual by the NEXT kernel's prologue (gemv_norm_kernel, GemvArgs::opart: 14 more float4 per thread, requested ahead of its weight
// stream), in a fixed order.  One launch boundary and one global round trip (attention partials -> o_proj) less per layer; the price is that
// the R row blocks of a head each read the head's whole K / V range (R x 2 x L x 256 B through one CU instead of 1 / 8 of it per wave).
// The Wo fragment of a workgroup is requested first (ITERS x 16 B per lane), so the GEMV needs no further round trip after the merge.
// ---------------------------------------------------------------------------------------------------------------
struct AttnOprojArgs {
    const float* qkv; float* kcache; float* vcache; const float* rope_cos; const float* rope_sin;
    int heads, kv_heads, max_len; const DecodeState* st;
    const bf16_t* wo; int hidden;        // Wo [hidden][heads * 64] bf16 row-major
    float* opart;                        // [heads][hidden] fp32
    int rblocks, rows_per_block;         // gridDim.x = heads * rblocks; rows_per_block <= 8 * NW * ITERS
};

template <int NW, int ITERS, int NS>                // NS key slots per lane group and pass: a wave takes 4 * NS keys per pass (<8, 4, 10>: 224 registers, <16, 2, 6>: 128 - no scratch)
__global__ __launch_bounds__(NW * 64) void attn_oproj_kernel(AttnOprojArgs p) {
    constexpr int PASS = 4 * NS;
    __shared__ __attribute__((aligned(16))) float pw[NW][ATTN_PART];
    __shared__ __attribute__((aligned(16))) float as[64];
    __shared__ u32x4 wlds[ITERS][NW * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, sub = lane & 15, grp = lane >> 4;
    const int h = blockIdx.x / p.rblocks, rb = blockIdx.x % p.rblocks, gsz = p.heads / p.kv_heads, g = h / gsz;
    // this workgroup's rows of Wo[:, 64 h .. 64 h + 63]: 8 lanes per row (16 B each), 8 rows per wave and iteration - requested before anything else
    const int sub8 = lane & 7, rgrp = lane >> 3;
    const int row_lo = rb * p.rows_per_block, row_hi = min(p.hidden, row_lo + p.rows_per_block);
    u32x4 wv[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int row = min(row_lo + (it * NW + wave) * 8 + rgrp, p.hidden - 1);
        wv[it] = *reinterpret_cast<const u32x4*>(p.wo + (long long)row * (p.heads * 64) + h * 64 + sub8 * 8);
    }
    const int pos = p.st->pos;                       // the new token sits at index `pos`
    const int L = pos + 1;
    const int per = ((L + NW * 4 - 1) / (NW * 4)) * 4;                       // keys per slice (multiple of 4)
    const int kb = wave * per, ke = min(L, kb + per);
    const float* kc = p.kcache + (long long)g * p.max_len * 64;
    const float* vc = p.vcache + (long long)g * p.max_len * 64;
    float4 k4[NS], v4[NS];
    auto load_pass = [&](int base) {
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) {
            const int j = base + sl * 4 + grp;
            const unsigned o = (unsigned)(((j < pos && j < ke) ? j : 0) * 64 + sub * 4);      // unconditional, clamped; 32-bit offset from a uniform base (saddr form: no 64-bit address pair per load)
            k4[sl] = *reinterpret_cast<const float4*>(kc + o);
            v4[sl] = *reinterpret_cast<const float4*>(vc + o);
        }
    };
    load_pass(kb);
    // the Wo fragment was requested first, so it lands first (vmcnt retires in order): park it in LDS (each lane its own slots, no barrier) - 16
    // registers less across the attention, and the K / V loads stay in flight behind the counted wait
#pragma unroll
    for (int it = 0; it < ITERS; ++it) wlds[it][tid] = wv[it];
    const float* qraw = p.qkv + h * 64;
    const float* kq = p.qkv + p.heads * 64 + g * 64;
    const float* vq = p.qkv + (p.heads + p.kv_heads) * 64 + g * 64;
    const int d0 = sub * 4, dp = (d0 + 32) & 63;
    const float4 c4 = *reinterpret_cast<const float4*>(p.rope_cos + pos * 32 + (d0 & 31));
    const float4 s4 = *reinterpret_cast<const float4*>(p.rope_sin + pos * 32 + (d0 & 31));
    const float4 qa = *reinterpret_cast<const float4*>(qraw + d0), qb = *reinterpret_cast<const float4*>(qraw + dp);
    const float4 ka = *reinterpret_cast<const float4*>(kq + d0), kp = *reinterpret_cast<const float4*>(kq + dp);
    const float4 vn4 = *reinterpret_cast<const float4*>(vq + d0);
    const float sg = d0 < 32 ? -1.f : 1.f;
    const float4 q4 = make_float4(qa.x * c4.x + sg * qb.x * s4.x, qa.y * c4.y + sg * qb.y * s4.y, qa.z * c4.z + sg * qb.z * s4.z, qa.w * c4.w + sg * qb.w * s4.w);
    const float4 kn4 = make_float4(ka.x * c4.x + sg * kp.x * s4.x, ka.y * c4.y + sg * kp.y * s4.y, ka.z * c4.z + sg * kp.z * s4.z, ka.w * c4.w + sg * kp.w * s4.w);
    if (!p.st->done && wave == 0 && rb == 0 && h % gsz == 0 && grp == 0) {  // KV-cache append: one wave per kv head
        *reinterpret_cast<float4*>(p.kcache + ((long long)g * p.max_len + pos) * 64 + d0) = kn4;
        *reinterpret_cast<float4*>(p.vcache + ((long long)g * p.max_len + pos) * 64 + d0) = vn4;
    }
    const float NEG = -__builtin_huge_valf();
    float m_run = NEG, l_run = 0.f;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int base = kb; base < ke; base += PASS) {   // wave-uniform
        if (base != kb) load_pass(base);
        float sc[NS];
        float mt = NEG;
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) {
            const int j = base + sl * 4 + grp;
            const float4 kk = (j == pos) ? kn4 : k4[sl];
            float a = q4.x * kk.x + q4.y * kk.y + q4.z * kk.z + q4.w * kk.w;
            a = group16_sum(a) * 0.125f;
            sc[sl] = j < ke ? a : NEG;
            mt = fmaxf(mt, sc[sl]);
        }
        mt = fmaxf(mt, __shfl_xor(mt, 16)); mt = fmaxf(mt, __shfl_xor(mt, 32));
        const float m_new = fmaxf(m_run, mt);
        const float scale = (m_run == NEG) ? 0.f : expf(m_run - m_new);
        acc.x *= scale; acc.y *= scale; acc.z *= scale; acc.w *= scale;
        float lt = 0.f;
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) {
            const int j = base + sl * 4 + grp;
            const float e = (sc[sl] == NEG) ? 0.f : expf(sc[sl] - m_new);
            const float4 vv = (j == pos) ? vn4 : v4[sl];
            acc.x += e * vv.x; acc.y += e * vv.y; acc.z += e * vv.z; acc.w += e * vv.w;
            lt += e;
        }
        l_run = l_run * scale + lt;
        m_run = m_new;
    }
    acc.x += __shfl_xor(acc.x, 16); acc.y += __shfl_xor(acc.y, 16); acc.z += __shfl_xor(acc.z, 16); acc.w += __shfl_xor(acc.w, 16);
    acc.x += __shfl_xor(acc.x, 32); acc.y += __shfl_xor(acc.y, 32); acc.z += __shfl_xor(acc.z, 32); acc.w += __shfl_xor(acc.w, 32);
    l_run += __shfl_xor(l_run, 16); l_run += __shfl_xor(l_run, 32);
    if (grp == 0) *reinterpret_cast<float4*>(&pw[wave][d0]) = acc;
    if (lane == 0) { pw[wave][64] = (l_run > 0.f) ? m_run : 0.f; pw[wave][65] = l_run; }
    __syncthreads();
    if (tid < 16) {                                  // flash-decoding merge of the NW slices (fixed order), normalised: a_h
        float M = NEG;
#pragma unroll
        for (int w = 0; w < NW; ++w) if (pw[w][65] > 0.f) M = fmaxf(M, pw[w][64]);
        float den = 0.f; float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const float wgt = (pw[w][65] > 0.f) ? expf(pw[w][64] - M) : 0.f;
            const float4 t = *reinterpret_cast<const float4*>(&pw[w][tid * 4]);
            den += wgt * pw[w][65];
            a.x += wgt * t.x; a.y += wgt * t.y; a.z += wgt * t.z; a.w += wgt * t.w;
        }
        const float inv = 1.f / den;
        *reinterpret_cast<float4*>(&as[tid * 4]) = make_float4(a.x * inv, a.y * inv, a.z * inv, a.w * inv);
    }
    __syncthreads();
    const float4 xa = *reinterpret_cast<const float4*>(&as[sub8 * 8]), xb = *reinterpret_cast<const float4*>(&as[sub8 * 8 + 4]);
    float* out = p.opart + (long long)h * p.hidden;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const u32x4 u = wlds[it][tid];
        float a = 0.f;
        a += __uint_as_float(u[0] << 16) * xa.x;          a += __uint_as_float(u[0] & 0xffff0000u) * xa.y;
        a += __uint_as_float(u[1] << 16) * xa.z;          a += __uint_as_float(u[1] & 0xffff0000u) * xa.w;
        a += __uint_as_float(u[2] << 16) * xb.x;          a += __uint_as_float(u[2] & 0xffff0000u) * xb.y;
        a += __uint_as_float(u[3] << 16) * xb.z;          a += __uint_as_float(u[3] & 0xffff0000u) * xb.w;
        a += dpp_row<0xB1>(a); a += dpp_row<0x4E>(a); a += dpp_row<0x141>(a);           // sum over the 8 lanes of a row
        const int row = row_lo + (it * NW + wave) * 8 + rgrp;
        if (sub8 == 0 && row < row_hi) out[row] = a;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// QKV projection + decode attention in ONE launch (one kernel boundary less per layer).
//   workgroups [0, heads * nsplit):   (head h, key slice sp): RMSNorm (shared through LDS) -> the 64 q rows of head h as a 4-wave GEMV ->
//       RoPE -> attention over the CACHED keys of the slice (4 waves x 12 keys per pass, online softmax, merged through LDS) -> one
//       un-normalised partial (64 numerators, max, denominator).  The q GEMV is recomputed by the nsplit workgroups of a head (its 115 KB of
//       weights come from L2 after the first) - cheaper than a kernel boundary on this chip.
//   workgroups [heads * nsplit, + 2 * kv_heads):   the new token's k (RoPE'd) and v rows of one kv head: appended to the cache and left in
//       knew / vnew.  The new token's own score / value cannot be used by the attention workgroups of the same launch, so the o_proj merge
//       (gemv_kernel<..., NSP>, GemvArgs::qnew) adds it as one more partial.
// ---------------------------------------------------------------------------------------------------------------
struct QkvAttnArgs {
    const bf16_t* W; const float* bias; const float* x; const float* gamma; float eps; int K;
    float* kcache; float* vcache; const float* rope_cos; const float* rope_sin;
    int heads, kv_heads, max_len; const DecodeState* st;
    float* part; int nsplit; float* qnew; float* knew; float* vnew;
};

template <int STEPS>
__global__ __launch_bounds__(256) void qkv_attn_kernel(QkvAttnArgs p) {
    constexpr int NS = 3, WPASS = 4 * NS, PASS = 4 * WPASS;           // keys per wave / workgroup and pass
    __shared__ __attribute__((aligned(16))) float xs[STEPS * 128];
    __shared__ __attribute__((aligned(16))) float vec[64];            // raw projection rows of this workgroup's head
    __shared__ __attribute__((aligned(16))) float rot[64];            // after RoPE
    __shared__ __attribute__((aligned(16))) float pw[4][ATTN_PART];
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, grp = lane >> 4, sub = lane & 15;
    const int steps = p.K / 128;
    const int n_attn = p.heads * p.nsplit;
    const bool is_attn = (int)blockIdx.x < n_attn;
    const int h = is_attn ? blockIdx.x / p.nsplit : 0, sp = is_attn ? blockIdx.x % p.nsplit : 0;
    const int kvj = is_attn ? 0 : blockIdx.x - n_attn;                 // 0 .. kv_heads-1: k heads, then v heads
    const int head_row = is_attn ? h * 64 : (p.heads + kvj) * 64;      // first of this workgroup's 64 projection rows
    const int gsz = p.heads / p.kv_heads, g = is_attn ? h / gsz : kvj % p.kv_heads;
    const int pos = p.st->pos;

    // 1. every weight row of this wave (16 rows: 4 per 16-lane group)
    u32x4 w[4][STEPS];
    const int row_in_head = (wave * 4 + grp) * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const bf16_t* wr = p.W + (long long)(head_row + row_in_head + r) * p.K + sub * 8;
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            const bool ok = s < steps;
            u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wr + (ok ? s : 0) * 128));
            if (!ok) t = (u32x4){0u, 0u, 0u, 0u};
            w[r][s] = t;
        }
    }
    // 2. attention workgroups: the first pass of cached keys of this wave, requested before anything is consumed
    const int per = ((pos + p.nsplit * 4 - 1) / (p.nsplit * 4)) * 4;   // cached keys per slice (multiple of 4); the new token is not among them
    const int kb = sp * per, ke = min(pos, kb + per);
    const float* kc = p.kcache + (long long)g * p.max_len * 64;
    const float* vc = p.vcache + (long long)g * p.max_len * 64;
    float4 k4[NS], v4[NS];
    auto load_pass = [&](int base) {
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) {
            const int j = base + wave * WPASS + sl * 4 + grp;
            const long long o = (long long)(j < ke ? j : 0) * 64 + sub * 4;   // unconditional, clamped
            k4[sl] = *reinterpret_cast<const float4*>(kc + o);
            v4[sl] = *reinterpret_cast<const float4*>(vc + o);
        }
    };
    if (is_attn) load_pass(kb);
    // 3. RMSNorm of x, shared through LDS (see gemv_norm_kernel)
    {
        const bool have = tid * 4 < p.K;
        const int k = have ? tid * 4 : 0;
        float4 xv = *reinterpret_cast<const float4*>(p.x + k), gv = *reinterpret_cast<const float4*>(p.gamma + k);
        if (!have) xv = make_float4(0.f, 0.f, 0.f, 0.f);
        float ss = wave_sum(xv.x * xv.x + xv.y * xv.y + xv.z * xv.z + xv.w * xv.w);
        if (lane == 0) red[wave] = ss;
        __syncthreads();
        ss = (red[0] + red[1]) + (red[2] + red[3]);
        const float rstd = rsqrtf(ss / (float)p.K + p.eps);
        if (have) *reinterpret_cast<float4*>(&xs[k]) = make_float4(xv.x * rstd * gv.x, xv.y * rstd * gv.y, xv.z * rstd * gv.z, xv.w * rstd * gv.w);
        __syncthreads();
    }
    // 4. the 64 projection rows
    float acc4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        if (s < steps) {
            const float4 xa = *reinterpret_cast<const float4*>(&xs[s * 128 + sub * 8]), xb = *reinterpret_cast<const float4*>(&xs[s * 128 + sub * 8 + 4]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const u32x4 u = w[r][s];
                float a = acc4[r];
                a += __uint_as_float(u[0] << 16) * xa.x;          a += __uint_as_float(u[0] & 0xffff0000u) * xa.y;
                a += __uint_as_float(u[1] << 16) * xa.z;          a += __uint_as_float(u[1] & 0xffff0000u) * xa.w;
                a += __uint_as_float(u[2] << 16) * xb.x;          a += __uint_as_float(u[2] & 0xffff0000u) * xb.y;
                a += __uint_as_float(u[3] << 16) * xb.z;          a += __uint_as_float(u[3] & 0xffff0000u) * xb.w;
                acc4[r] = a;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) acc4[r] = group16_sum(acc4[r]);
    if (sub == 0) {
        const float4 b = p.bias ? *reinterpret_cast<const float4*>(p.bias + head_row + row_in_head) : make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(&vec[row_in_head]) = make_float4(acc4[0] + b.x, acc4[1] + b.y, acc4[2] + b.z, acc4[3] + b.w);
    }
    __syncthreads();
    // 5. rotate-half RoPE (q and k heads; v passes through)
    const bool is_v = !is_attn && kvj >= p.kv_heads;
    if (tid < 64) {
        float v = vec[tid];
        if (!is_v) {
            const int f = tid & 31;
            const float c = p.rope_cos[pos * 32 + f], sn = p.rope_sin[pos * 32 + f];
            v = tid < 32 ? v * c - vec[tid + 32] * sn : v * c + vec[tid - 32] * sn;
        }
        rot[tid] = v;
        if (is_attn) { if (sp == 0) p.qnew[h * 64 + tid] = v; }
        else {
            float* dst = is_v ? p.vnew : p.knew;
            dst[g * 64 + tid] = v;
            if (!p.st->done) (is_v ? p.vcache : p.kcache)[((long long)g * p.max_len + pos) * 64 + tid] = v;      // KV-cache append
        }
    }
    if (!is_attn) return;
    __syncthreads();
    // 6. attention of this wave over its share of the slice
    const float4 q4 = *reinterpret_cast<const float4*>(&rot[sub * 4]);
    const float NEG = -__builtin_huge_valf();
    float m_run = NEG, l_run = 0.f;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int base = kb; base < ke; base += PASS) {      // workgroup-uniform
        if (base != kb) load_pass(base);
        float sc[NS];
        float mt = NEG;
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) {
            const int j = base + wave * WPASS + sl * 4 + grp;
            const float4 kk = k4[sl];
            float a = q4.x * kk.x + q4.y * kk.y + q4.z * kk.z + q4.w * kk.w;
            a = group16_sum(a) * 0.125f;
            sc[sl] = j < ke ? a : NEG;
            mt = fmaxf(mt, sc[sl]);
        }
        mt = fmaxf(mt, __shfl_xor(mt, 16)); mt = fmaxf(mt, __shfl_xor(mt, 32));
        if (mt == NEG) continue;                         // wave-uniform: this wave holds no key of the pass
        const float m_new = fmaxf(m_run, mt);
        const float scale = (m_run == NEG) ? 0.f : expf(m_run - m_new);
        acc.x *= scale; acc.y *= scale; acc.z *= scale; acc.w *= scale;
        float lt = 0.f;
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) {
            const float e = (sc[sl] == NEG) ? 0.f : expf(sc[sl] - m_new);
            const float4 vv = v4[sl];
            acc.x += e * vv.x; acc.y += e * vv.y; acc.z += e * vv.z; acc.w += e * vv.w;
            lt += e;
        }
        l_run = l_run * scale + lt;                     // per-group partial (identical on the 16 lanes of a group)
        m_run = m_new;
    }
    acc.x += __shfl_xor(acc.x, 16); acc.y += __shfl_xor(acc.y, 16); acc.z += __shfl_xor(acc.z, 16); acc.w += __shfl_xor(acc.w, 16);
    acc.x += __shfl_xor(acc.x, 32); acc.y += __shfl_xor(acc.y, 32); acc.z += __shfl_xor(acc.z, 32); acc.w += __shfl_xor(acc.w, 32);
    l_run += __shfl_xor(l_run, 16); l_run += __shfl_xor(l_run, 32);
    // 7. merge the 4 waves (fixed order) and leave one partial per (head, slice)
    if (grp == 0) *reinterpret_cast<float4*>(&pw[wave][sub * 4]) = acc;
    if (lane == 0) { pw[wave][64] = (l_run > 0.f) ? m_run : 0.f; pw[wave][65] = l_run; }
    __syncthreads();
    if (tid < 16) {
        float M = NEG;
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) if (pw[ww][65] > 0.f) M = fmaxf(M, pw[ww][64]);
        float den = 0.f; float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) {
            const float wgt = (pw[ww][65] > 0.f) ? expf(pw[ww][64] - M) : 0.f;
            const float4 t = *reinterpret_cast<const float4*>(&pw[ww][tid * 4]);
            den += wgt * pw[ww][65];
            a.x += wgt * t.x; a.y += wgt * t.y; a.z += wgt * t.z; a.w += wgt * t.w;
        }
        float* pr = p.part + (long long)blockIdx.x * ATTN_PART;
        *reinterpret_cast<float4*>(pr + tid * 4) = a;
        if (tid == 0) { pr[64] = den > 0.f ? M : 0.f; pr[65] = den; }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Prefill helpers
// ---------------------------------------------------------------------------------------------------------------
// RoPE on q (in place) and k of L rows, and append k,v to the cache at positions pos0..pos0+L-1.
static __global__ __launch_bounds__(256) void rope_store_kernel(float* qkv, int L, int heads, int kv_heads, int pos0,
                                                          const float* rope_cos, const float* rope_sin,
                                                          float* kcache, float* vcache, int max_len) {
    const int row = blockIdx.x, tid = threadIdx.x;
    const int width = (heads + 2 * kv_heads) * 64;
    float* r = qkv + (long long)row * width;
    const int pos = pos0 + row;
    const int nrot = (heads + kv_heads) * 32;                 // (head, f) pairs to rotate
    for (int i = tid; i < nrot; i += 256) {
        const int hh = i >> 5, f = i & 31;
        const float c = rope_cos[pos * 32 + f], s = rope_sin[pos * 32 + f];
        const float a = r[hh * 64 + f], b = r[hh * 64 + f + 32];
        const float ra = a * c - b * s, rb = b * c + a * s;
        if (hh < heads) { r[hh * 64 + f] = ra; r[hh * 64 + f + 32] = rb; }
        else {
            float* kc = kcache + ((long long)(hh - heads) * max_len + pos) * 64;
            kc[f] = ra; kc[f + 32] = rb;
        }
    }
    for (int i = tid; i < kv_heads * 64; i += 256) {
        const int g = i >> 6, d = i & 63;
        vcache[((long long)g * max_len + pos) * 64 + d] = r[(heads + kv_heads) * 64 + i];
    }
}

// act[m][j] = silu(gu[m][2j]) * gu[m][2j+1]
static __global__ __launch_bounds__(256) void silu_mul_kernel(const float* gu, float* act, long long n_out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_out) return;
    const float2 v = *reinterpret_cast<const float2*>(gu + 2 * i);
    act[i] = (v.x / (1.f + expf(-v.x))) * v.y;
}

// rows of a bf16 or fp32 table -> fp32 matrix: out[r][:] = table[ids[r]][:]
static __global__ __launch_bounds__(256) void gather_rows_kernel(const void* table, int bf16, const int* ids, int n_rows, int dim, float* out,
                                                          long long table_rows, float scale) {
    const int r = blockIdx.x;
    long long id = ids[r];
    if (id < 0) id = 0;
    if (id >= table_rows) id = table_rows - 1;
    for (int c = threadIdx.x; c < dim; c += 256) {
        const float v = bf16 ? bf16_to_f32(reinterpret_cast<const bf16_t*>(table)[id * dim + c]) : reinterpret_cast<const float*>(table)[id * dim + c];
        out[(long long)r * dim + c] = v * scale;
    }
}

// decode: h = speech_embedding[last_token]
static __global__ __launch_bounds__(256) void embed_last_token_kernel(const bf16_t* table, int dim, float* h, const DecodeState* st) {
    if (st->done) return;
    const long long id = st->last_token;
    for (int c = threadIdx.x; c < dim; c += 256) h[c] = bf16_to_f32(table[id * dim + c]);
}

// ---------------------------------------------------------------------------------------------------------------
// Sampling on device (one workgroup of 1024 threads), llm/llm.py:150-160 + :542-549 and utils/common.py:138-167.
//   mode 0: greedy argmax (first index on ties)            mode 1: repetition-aware sampling (RAS)
// ---------------------------------------------------------------------------------------------------------------
// Per-request sampling parameters live in DEVICE memory (written once per request), so that the captured decode graph does
// not depend on the request and is never re-captured when the text length (min_len / max_len) or the seed changes.
struct SampleParams {
    int mode; int eos; int n_stop; int min_len; int max_len; float top_p; int top_k; int win; float tau_r; int use_uniforms;
    unsigned long long seed;
};
struct SampleArgs {
    const float* logits; int V; const SampleParams* sp;
    const float* uniforms;                                    // explicit uniforms [2 per step] when sp->use_uniforms (parity tests)
    DecodeState* st; int* tokens; int max_tokens;
    const bf16_t* emb_table = nullptr; int emb_dim = 0; float* h_out = nullptr;   // when set: h_out = speech_embedding[token] for an emitted token (was embed_last_token_kernel)
    const float* emb_table_f32 = nullptr;                     // the same with an fp32 table (CosyVoice-300M's speech_embedding, lm1.hip)
    // batched decode: workgroup b = blockIdx.x samples slot b; element strides between the slots (single sequence: one workgroup, strides unused)
    long long slot_logits = 0, slot_uniforms = 0, slot_tokens = 0, slot_h = 0;
};

__device__ __forceinline__ float uniform01(unsigned long long seed, unsigned step, unsigned draw) {
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (2ull * step + draw + 1ull);      // splitmix64
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
    return (float)(z >> 40) * (1.0f / 16777216.0f);
}

static __global__ __launch_bounds__(1024) void sample_kernel(SampleArgs a) {
    {   // slot of a batched decode (blockIdx.x == 0 for the single-sequence path)
        const long long sl = blockIdx.x;
        a.logits += sl * a.slot_logits; a.sp += sl; a.st += sl; a.tokens += sl * a.slot_tokens;
        if (a.uniforms) a.uniforms += sl * a.slot_uniforms;
        if (a.h_out) a.h_out += sl * a.slot_h;
    }
    // flatten (static args + device-resident request parameters) into the names the body uses
    struct { const float* logits; int V, eos, n_stop, min_len, max_len, mode; float top_p; int top_k, win; float tau_r;
             unsigned long long seed; const float* uniforms; DecodeState* st; int* tokens; int max_tokens; } p;
    p.logits = a.logits; p.V = a.V; p.eos = a.sp->eos; p.n_stop = a.sp->n_stop; p.min_len = a.sp->min_len; p.max_len = a.sp->max_len;
    p.mode = a.sp->mode; p.top_p = a.sp->top_p; p.top_k = a.sp->top_k; p.win = a.sp->win; p.tau_r = a.sp->tau_r; p.seed = a.sp->seed;
    p.uniforms = a.sp->use_uniforms ? a.uniforms : nullptr; p.st = a.st; p.tokens = a.tokens; p.max_tokens = a.max_tokens;
    __shared__ float pr[8192];
    __shared__ float redv[16];
    __shared__ int redi[16];
    __shared__ float chunk[1024];
    __shared__ float selp[64];
    __shared__ int seli[64];
    __shared__ int s_tok;
    DecodeState* st = p.st;
    if (st->done) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int step = st->step;
    if (step >= p.max_len) { if (tid == 0) st->done = 1; return; }
    const float NEG = -__builtin_huge_valf();
    const bool ignore_eos = step < p.min_len;
    for (int i = tid; i < p.V; i += 1024) pr[i] = (ignore_eos && i == p.eos) ? NEG : p.logits[i];
    __syncthreads();

    // block arg-max with lowest-index tie-break
    auto block_argmax = [&](float& bv, int& bi) {
        float v = NEG; int ix = 0x7fffffff;
        for (int i = tid; i < p.V; i += 1024) { const float x = pr[i]; if (x > v || (x == v && i < ix)) { v = x; ix = i; } }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_xor(v, off); const int oi = __shfl_xor(ix, off);
            if (ov > v || (ov == v && oi < ix)) { v = ov; ix = oi; }
        }
        __syncthreads();
        if (lane == 0) { redv[wave] = v; redi[wave] = ix; }
        __syncthreads();
        v = redv[0]; ix = redi[0];
        for (int w = 1; w < 16; ++w) if (redv[w] > v || (redv[w] == v && redi[w] < ix)) { v = redv[w]; ix = redi[w]; }
        bv = v; bi = ix;
    };

    int tok;
    if (p.mode == 0) {
        float bv; block_argmax(bv, tok);
    } else {
        // softmax probabilities
        float mx = NEG;
        for (int i = tid; i < p.V; i += 1024) mx = fmaxf(mx, pr[i]);
        mx = block_max(mx, redv);
        float sm = 0.f;
        for (int i = tid; i < p.V; i += 1024) { const float e = pr[i] == NEG ? 0.f : expf(pr[i] - mx); pr[i] = e; sm += e; }
        sm = block_sum(sm, redv);
        __syncthreads();
        const float inv = 1.f / sm;
        for (int i = tid; i < p.V; i += 1024) pr[i] *= inv;
        __syncthreads();
        // nucleus: stable descending order, add while cum < top_p and count < top_k   (common.py:151-158)
        int cnt = 0; float cum = 0.f;
        while (cum < p.top_p && cnt < p.top_k) {
            float bv; int bi; block_argmax(bv, bi);
            if (tid == 0) { selp[cnt] = bv; seli[cnt] = bi; pr[bi] = -1.f; }       // -1 marks "taken" (restored below)
            cum += bv; ++cnt;
            __syncthreads();
        }
        if (tid == 0) {
            for (int c = 0; c < cnt; ++c) pr[seli[c]] = selp[c];
            const float u = p.uniforms ? p.uniforms[2 * step] : uniform01(p.seed, step, 0);
            float tot = 0.f; for (int c = 0; c < cnt; ++c) tot += selp[c];
            float acc = 0.f; int pick = cnt - 1;
            for (int c = 0; c < cnt; ++c) { acc += selp[c] / tot; if (u < acc) { pick = c; break; } }
            int t = seli[pick];
            int rep = 0;
            const int n = st->n_tokens;
            for (int w = 0; w < p.win && w < n; ++w) rep += p.tokens[n - 1 - w] == t;
            s_tok = (rep >= p.win * p.tau_r) ? -(t + 1) : t;      // negative: needs the full-distribution fallback
        }
        __syncthreads();
        tok = s_tok;
        if (tok < 0) {
            const int banned = -tok - 1;
            if (tid == 0) pr[banned] = 0.f;
            __syncthreads();
            // inverse CDF over the full (renormalised) distribution: per-thread contiguous chunks + serial scan of chunk sums
            const int per = (p.V + 1023) / 1024, b0 = tid * per, b1 = min(p.V, b0 + per);
            float cs = 0.f;
            for (int i = b0; i < b1; ++i) cs += pr[i];
            chunk[tid] = cs;
            __syncthreads();
            if (tid == 0) {
                float tot = 0.f; for (int c = 0; c < 1024; ++c) tot += chunk[c];
                const float u = (p.uniforms ? p.uniforms[2 * step + 1] : uniform01(p.seed, step, 1)) * tot;
                float acc = 0.f; int c = 0;
                for (; c < 1023; ++c) { if (u < acc + chunk[c]) break; acc += chunk[c]; }
                int i = c * per, last = i;
                const int e = min(p.V, i + per);
                for (; i < e; ++i) { if (pr[i] > 0.f) last = i; acc += pr[i]; if (u < acc) break; }
                s_tok = i < e ? i : last;
            }
            __syncthreads();
            tok = s_tok;
        }
    }
    const bool stop = tok >= p.eos && tok < p.eos + p.n_stop;
    if (tid == 0) {
        if (stop) { st->done = 1; st->stop_token = tok; }
        else {
            if (st->n_tokens < p.max_tokens) p.tokens[st->n_tokens] = tok;
            st->n_tokens += 1; st->last_token = tok; st->step = step + 1;
            if (step + 1 >= p.max_len) { /* loop ends after this token; the next replay marks done */ }
        }
    }
    if (a.h_out && !stop)                                     // input of the backbone step that follows in the same graph replay
        for (int c = tid; c < a.emb_dim; c += 1024)
            a.h_out[c] = a.emb_table_f32 ? a.emb_table_f32[(long long)tok * a.emb_dim + c] : bf16_to_f32(a.emb_table[(long long)tok * a.emb_dim + c]);
}

// advance the KV length after a backbone step
static __global__ void advance_pos_kernel(DecodeState* st) { if (!st->done) st->pos += 1; }

}  // namespace cv
