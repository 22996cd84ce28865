// CosyVoice-300M TransformerLM decode step behind ONE C entry point (reference: cosyvoice/llm/llm.py:162-223 TransformerLM.inference ->
// cosyvoice/transformer/encoder.py:267-327 forward_chunk -> encoder_layer.py:60-119 -> attention.py:200-330 RelPositionMultiHeadedAttention with its key / value cache).
//
// The step is one output row through 14 pre-norm layers over fp32 weights: 734 MB of weights per token, nothing else of size.  Until round 4 the host side
// (cosyvoice1_hip.py) replayed it as 115 launches of the general operators (LayerNorm, tiled GEMM -> GEMV, the [H][1][n] matrix_bd GEMM, the MFMA attention
// with one useful query row); what bounded it was the NUMBER of dependent launches, not their bytes.  Here the step is 73 launches (optionally one hipGraph):
//     per layer   [LayerNorm + (q+u | q+v | k | v) GEMV -> the layer's cache row]  [one-query relative-position attention, keys split over 8 workgroups per head]
//                 [merge of the 8 partial softmaxes + output GEMV + residual]  [LayerNorm + w_1 GEMV + ReLU]  [w_2 GEMV + residual]
//     around it   [embed GEMV] [LayerNorm + ReLU, * sqrt(d)] ... [after_norm + decoder GEMV -> logits]
// The GEMV is gemm_conv.h's gemv_f32_kernel (same lane / wave / k order, same four-way combine) with its input vector staged in LDS by a prologue; the LayerNorm
// prologue repeats norm_rows_kernel's register path arithmetic, so every product of the step has the bits of the launch-per-operator path.  The attention is a
// new summation order (exact fp32 dot products per key instead of the MFMA chain): results agree to fp32 rounding.
// What changes from token to token - the position, i.e. the cache row written, the key count and the first row of the relative-position table - is read by
// the kernels from a device block (Lm1Dyn) that the step's FIRST kernel (launched outside the graph, it also takes the input row's address) updates; what
// changes from request to request - the cache and table buffers - is rebound there by cv_lm1_bind.  One captured graph therefore serves every step of every request
// (option "graph", off by default: on the MI355X the replay costs 0.607 ms per token against 0.591 ms for the same launches issued one by one).
#include "api_common.h"
#include "common.h"
#include "llm_kernels.h"
#include <cstdlib>
#include <type_traits>
#include <vector>

namespace cv {

constexpr int LM1_MAX_LAYERS = 32;
constexpr int LM1_SPLITS = 8;                 // key ranges per head in the decode attention
constexpr int LM1_MAX_KEYS_PER_SPLIT = 4096;  // (a bound on the cache a handle binds: 32768 positions)
#define LM1_NEG_INF (-__builtin_huge_valf())
constexpr int LM1_MAX_K = 4096;               // longest GEMV input (the feed-forward width)

struct Lm1Dyn {
    int pos, n_tab, pad0, pad1;
    float* rows[LM1_MAX_LAYERS];              // per layer [cap][4 d]: (q + u | q + v | k | v) of every position so far
    const float* tabs[LM1_MAX_LAYERS];        // per layer [2 n_tab - 1][d]: linear_pos(pe), row m = relative position n_tab - 1 - m
};

enum { LM1_PRO_NONE = 0, LM1_PRO_LN = 1, LM1_PRO_MERGE = 2 };

struct Lm1GemvArgs {
    const float* x;                            // [K] (NONE / LN) or the attention partials [heads][LM1_SPLITS][66] (MERGE)
    const float* g; const float* b; float eps; // LN prologue
    const float* W; long long ldw; const float* bias; const float* res; float* y;
    Lm1Dyn* dyn; int layer;                    // layer >= 0: y = dyn->rows[layer] + pos * N   (the cache row of this position)
    int set_pos;                               // >= 0: block 0 publishes the step's position (first kernel of a step); -2: the position of the device loop (st->pos)
    const DecodeState* st;                     // set_pos == -2 (cv_lm1_decode): the loop state the sampler and advance_pos_kernel keep on the device
    int N, K, Kp, act, pro;
};

// The prologue of the decode GEMVs: the input vector into LDS (xs, zero beyond K up to steps * 64 floats), by one of three rules; `issue_weights()` requests the caller's
// first weight loads at the point where they hide the most (before everything - or, under the LayerNorm, right behind the row's own loads).
template <int NW, typename F>
static __device__ __forceinline__ void lm1_gemv_prologue(const Lm1GemvArgs& p, float* xs, float* mw, int steps, F&& issue_weights) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // (every prologue requests its own small operands BEFORE the caller's weight rows: vmcnt retires in issue order, what the prologue waits for must not queue behind the weight stream)
    // ---- prologue: the input vector into LDS, zero beyond K up to the last 64-float step
    if (p.pro == LM1_PRO_LN) {
        // norm_rows_kernel's register path (C <= 1024, C % 4 == 0), every wave on the whole row; wave w parks chunk w
        float4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = lane * 4 + k * 256;
            const float4 t = *reinterpret_cast<const float4*>(p.x + min(c, p.K - 4));        // unconditional (clamped) loads: all four in flight together
            const bool ok = c < p.K;
            v[k] = make_float4(ok ? t.x : 0.f, ok ? t.y : 0.f, ok ? t.z : 0.f, ok ? t.w : 0.f);
        }
        const int cw = lane * 4 + wave * 256;                   // gamma / beta of the wave's chunk travel with the row, not after the statistics
        const float4 gw = *reinterpret_cast<const float4*>(p.g + min(cw, p.K - 4)), bw = *reinterpret_cast<const float4*>(p.b + min(cw, p.K - 4));
        issue_weights();
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (lane * 4 + k * 256 < p.K) s += v[k].x + v[k].y + v[k].z + v[k].w;
        s = wave_sum(s);
        const float mean = s / (float)p.K;
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (lane * 4 + k * 256 < p.K) { const float a = v[k].x - mean, b = v[k].y - mean, cc = v[k].z - mean, d = v[k].w - mean; q += a * a + b * b + cc * cc + d * d; }
        q = wave_sum(q);
        const float rstd = rsqrtf(q / (float)p.K + p.eps);
        {
            // the chunk this wave parks (same expressions as norm_rows_kernel's epilogue)
            const float4 vw = make_float4(wave == 0 ? v[0].x : wave == 1 ? v[1].x : wave == 2 ? v[2].x : v[3].x, wave == 0 ? v[0].y : wave == 1 ? v[1].y : wave == 2 ? v[2].y : v[3].y,
                                          wave == 0 ? v[0].z : wave == 1 ? v[1].z : wave == 2 ? v[2].z : v[3].z, wave == 0 ? v[0].w : wave == 1 ? v[1].w : wave == 2 ? v[2].w : v[3].w);
            float o[4] = {0.f, 0.f, 0.f, 0.f};
            if (cw < p.K) {
                o[0] = (vw.x - mean) * rstd; o[1] = (vw.y - mean) * rstd; o[2] = (vw.z - mean) * rstd; o[3] = (vw.w - mean) * rstd;
                o[0] *= gw.x; o[1] *= gw.y; o[2] *= gw.z; o[3] *= gw.w;
                o[0] += bw.x; o[1] += bw.y; o[2] += bw.z; o[3] += bw.w;
            }
            if (cw < steps * 64) *reinterpret_cast<float4*>(&xs[cw]) = make_float4(o[0], o[1], o[2], o[3]);
        }
    } else if (p.pro == LM1_PRO_MERGE) {
        // x[h * 64 + d] = sum_s acc_s[d] * (e^(m_s - M) / sum_s l_s e^(m_s - M)): the key ranges of one head's softmax, in range order.  The LM1_SPLITS weights of a
        // head are made ONCE per workgroup (one lane per (head, range), the head's lanes combine through shuffles) and read back from LDS - not once per element.
        // Round 6: a lane's partial accumulators (up to 4 elements x LM1_SPLITS ranges) are REQUESTED before the weights are made - they do not depend on them, and
        // behind the barrier they were a second global round trip on the critical path of the launch (5.5 us for a 2 - 4 MB matrix).  Same sums, same order.
        constexpr int EPT = 4;
        const int hs = min(tid, (p.K >> 6) * LM1_SPLITS - 1);
        const float* ph = p.x + (long long)hs * 66;
        const float ms = ph[0], ls = ph[1];                      // first in the queue: the weights' shuffles start when these two land
        float as[EPT][LM1_SPLITS];
#pragma unroll
        for (int i = 0; i < EPT; ++i) {
            const int e = min(tid + i * (int)blockDim.x, p.K - 1);
            const float* pa = p.x + (long long)(e >> 6) * LM1_SPLITS * 66 + 2 + (e & 63);
#pragma unroll
            for (int s = 0; s < LM1_SPLITS; ++s) as[i][s] = pa[s * 66];
        }
        issue_weights();
        {
            float M = ms;
#pragma unroll
            for (int o = 1; o < LM1_SPLITS; o <<= 1) M = fmaxf(M, __shfl_xor(M, o));
            const float w = (ms == LM1_NEG_INF) ? 0.f : expf(ms - M);
            float L = ls * w;
#pragma unroll
            for (int o = 1; o < LM1_SPLITS; o <<= 1) L += __shfl_xor(L, o);
            if (tid < (p.K >> 6) * LM1_SPLITS) mw[tid] = w / L;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < EPT; ++i) {
            const int e = tid + i * (int)blockDim.x;
            if (e < steps * 64) {
                float o = 0.f;
                if (e < p.K) {
#pragma unroll
                    for (int s = 0; s < LM1_SPLITS; ++s) o += as[i][s] * mw[(e >> 6) * LM1_SPLITS + s];
                }
                xs[e] = o;
            }
        }
        for (int e = tid + EPT * (int)blockDim.x; e < steps * 64; e += blockDim.x) {      // (beyond 4 elements per lane: not reached at the supported widths, d <= 1024)
            float o = 0.f;
            if (e < p.K) {
                const float* pa = p.x + (long long)(e >> 6) * LM1_SPLITS * 66 + 2 + (e & 63);
                float a2[LM1_SPLITS];
#pragma unroll
                for (int s = 0; s < LM1_SPLITS; ++s) a2[s] = pa[s * 66];
#pragma unroll
                for (int s = 0; s < LM1_SPLITS; ++s) o += a2[s] * mw[(e >> 6) * LM1_SPLITS + s];
            }
            xs[e] = o;
        }
    } else {
        // (requesting the vector BEFORE the weight rows - llm_kernels.h XFIRST - measured the same on this model: 327 vs 321 us per token, round 6; the weights go first)
        issue_weights();
        for (int e = tid * 4; e < steps * 64; e += 256 * NW) {
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < p.K) t = *reinterpret_cast<const float4*>(p.x + e);           // K % 4 == 0
            *reinterpret_cast<float4*>(&xs[e]) = t;
        }
    }
}

// y[N] = act(pro(x)[K] . W[N][K]^T + bias) (+ res).  4 output rows per workgroup, a 16-lane group per row, the 4 waves split K (gemv_f32_kernel).
// U: 64-float steps a lane group has in flight (a wave's share of K = 1024 is 4 steps).  The wave's FIRST group of weight loads is issued before the prologue:
// the weights do not depend on the previous kernel's output, the input vector does, so the prologue's dependent loads (x, gamma, beta) wait under them.
// NW: waves per workgroup (they split K): 4, or 8 for the K = 4096 product whose 1024 rows are only 256 workgroups - 16 steps per wave were two dependent
// round trips per wave, 8 steps are one.
// W16 (round 6, the model's fp16 mode: cv_lm1_use_bf16): the matrix is stored as bf16 [N][Kp]; a lane takes the SAME four k-values of a step as 8 bytes and widens them when
// the step is consumed - the products and their order are those of the fp32 kernel on the bf16-rounded weights, half the bytes per token.
typedef unsigned lm1_u32x2 __attribute__((ext_vector_type(2)));
// RPG: output rows per 16-lane group (4 RPG rows per workgroup).  A row's products and their order do not depend on it; what does is how much of the prologue (the input
// vector, its LayerNorm, the barrier) a workgroup pays per byte of weights and how many weight bytes a CU has in flight (option "gemv_rows").
template <int U, int NW, bool W16 = false, int RPG = 1>
static __global__ __launch_bounds__(64 * NW) void lm1_gemv_kernel(Lm1GemvArgs p) {
    __shared__ float xs[LM1_MAX_K + 64];
    __shared__ float part[NW][4 * RPG];
    __shared__ float mw[256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, grp = lane >> 4, sub = lane & 15;
    const int steps = (p.Kp + 63) / 64;
    if (p.set_pos >= 0 && blockIdx.x == 0 && tid == 0) p.dyn->pos = p.set_pos;
    if (p.set_pos == -2 && blockIdx.x == 0 && tid == 0) p.dyn->pos = p.st->pos;
    const int row0 = ((int)blockIdx.x * 4 + grp) * RPG;           // this group's rows row0 .. row0 + RPG - 1 (clamped below: the reductions are wave collectives)
    const int s0 = wave * steps / NW, s1 = (wave + 1) * steps / NW;
    using raw_t = typename std::conditional<W16, lm1_u32x2, v4f>::type;
    raw_t w[RPG][U], wn[RPG][U];
    auto load_w = [&](raw_t (&dst)[RPG][U], int sb) {
#pragma unroll
        for (int i = 0; i < RPG; ++i) {
            const long long ro = (long long)min(row0 + i, p.N - 1) * p.ldw;
#pragma unroll
            for (int u = 0; u < U; ++u) {                       // unconditional loads (clamped step) keep the vmcnt bookkeeping exact
                const int k = min(min(sb + u, s1 - 1) * 64 + sub * 4, p.Kp - 4);
                if constexpr (W16) dst[i][u] = __builtin_nontemporal_load(reinterpret_cast<const lm1_u32x2*>(reinterpret_cast<const unsigned short*>(p.W) + ro + k));
                else dst[i][u] = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p.W + ro + k));
            }
        }
    };
    auto wide = [](const raw_t& r) -> v4f {
        if constexpr (W16) return (v4f){__uint_as_float(r[0] << 16), __uint_as_float(r[0] & 0xffff0000u), __uint_as_float(r[1] << 16), __uint_as_float(r[1] & 0xffff0000u)};
        else return r;
    };
    lm1_gemv_prologue<NW>(p, xs, mw, steps, [&] { if (s0 < s1) load_w(w, s0); });
    __syncthreads();
    // ---- the rows' dot products (gemv_f32_kernel: same loads, same order)
    float acc[RPG];
#pragma unroll
    for (int i = 0; i < RPG; ++i) acc[i] = 0.f;
    for (int sb = s0; sb < s1; sb += U) {
        const bool more = sb + U < s1;
        if (more) load_w(wn, sb + U);                           // the next group is requested before this one is consumed
        float4 x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = min(sb + u, s1 - 1) * 64 + sub * 4;
            x[u] = *reinterpret_cast<const float4*>(&xs[k]);
            if (sb + u >= s1) x[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < RPG; ++i)
#pragma unroll
            for (int u = 0; u < U; ++u) { const v4f wf = wide(w[i][u]); acc[i] += wf[0] * x[u].x; acc[i] += wf[1] * x[u].y; acc[i] += wf[2] * x[u].z; acc[i] += wf[3] * x[u].w; }
        if (more) {
#pragma unroll
            for (int i = 0; i < RPG; ++i)
#pragma unroll
                for (int u = 0; u < U; ++u) w[i][u] = wn[i][u];
        }
    }
#pragma unroll
    for (int i = 0; i < RPG; ++i) {
        const float a = group16_sum(acc[i]);
        if (sub == 0) part[wave][grp * RPG + i] = a;
    }
    __syncthreads();
    if (wave != 0 || sub >= RPG) return;
    const int gi = grp * RPG + sub, n = row0 + sub;              // lane `sub` of the group finishes its row `sub`
    if (n >= p.N) return;
    float v = ((part[0][gi] + part[1][gi]) + part[2][gi]) + part[3][gi];
    if (NW == 8) v = (((v + part[4 % NW][gi]) + part[5 % NW][gi]) + part[6 % NW][gi]) + part[7 % NW][gi];
    if (p.bias) v += p.bias[n];
    v = apply_act(p.act, v, 0.f);
    if (p.res) v += p.res[n];
    float* y = p.layer >= 0 ? p.dyn->rows[p.layer] + (long long)p.dyn->pos * p.N : p.y;
    y[n] = v;
}

// The bf16 matrices again, laid out over the lanes the way the CosyVoice2 decode GEMVs are (llm_kernels.h gemv_norm_kernel), round 6: a 16-lane group owns ROWS whole rows
// and a lane 8 consecutive k-values per 128-float step as ONE 16-byte load, every load of the workgroup's share requested before the prologue's barrier.
//   KSPLIT = false: a wave's four groups keep their rows to themselves over the whole K (K <= 128 STEPS) - 4 NW ROWS rows per workgroup, e.g. 256 workgroups for the
//                   [4096][1024] products: the input row and its LayerNorm are made once per 16 rows instead of once per 4 (lm1_gemv_kernel: 1024 workgroups, each
//                   re-reading 16 KB of x / gamma / beta against 8 KB of weights);
//   KSPLIT = true:  the NW waves share 4 ROWS rows and split K (the [1024][4096] product and the products with few rows), partial sums combined in wave order.
// The k order of a row's sum differs from lm1_gemv_kernel's: results agree to fp32 rounding, the bf16 weights are the same values (option "gemv16_wide", default 1).
typedef unsigned lm1_u32x4 __attribute__((ext_vector_type(4)));
// W16 = false: the same layout over fp32 matrices (a step = 16 lanes x 4 floats = 64 k-values; KSPLIT = false only: the split forms ARE lm1_gemv_kernel's) - option "gemv_wide".
template <bool W16, int STEPS, int ROWS, int NW, bool KSPLIT>
static __global__ __launch_bounds__(64 * NW) void lm1_gemv16_kernel(Lm1GemvArgs p) {
    constexpr int SK = W16 ? 128 : 64, LK = W16 ? 8 : 4;        // k-values per step / per lane and step
    __shared__ __attribute__((aligned(16))) float xs[LM1_MAX_K + 64];
    __shared__ float part[NW][4 * ROWS];
    __shared__ float mw[256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, grp = lane >> 4, sub = lane & 15;
    const int steps = p.Kp / SK;                                 // steps of a row (the launcher checks Kp % SK == 0)
    if (p.set_pos >= 0 && blockIdx.x == 0 && tid == 0) p.dyn->pos = p.set_pos;
    if (p.set_pos == -2 && blockIdx.x == 0 && tid == 0) p.dyn->pos = p.st->pos;
    const int unit = KSPLIT ? (int)blockIdx.x * 4 + grp : ((int)blockIdx.x * NW + wave) * 4 + grp;
    const int row0 = unit * ROWS;
    const int s0 = KSPLIT ? wave * steps / NW : 0, s1 = KSPLIT ? (wave + 1) * steps / NW : steps;
    lm1_u32x4 w[ROWS][STEPS];
    lm1_gemv_prologue<NW>(p, xs, mw, steps * SK / 64, [&] {
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const long long ro = (long long)min(row0 + r, p.N - 1) * p.ldw + sub * LK;
#pragma unroll
            for (int s = 0; s < STEPS; ++s) {                     // unconditional loads (clamped step): the vmcnt bookkeeping stays exact
                const long long k = ro + (long long)min(s0 + s, max(s1 - 1, 0)) * SK;
                if constexpr (W16) w[r][s] = __builtin_nontemporal_load(reinterpret_cast<const lm1_u32x4*>(reinterpret_cast<const unsigned short*>(p.W) + k));
                else w[r][s] = __builtin_nontemporal_load(reinterpret_cast<const lm1_u32x4*>(p.W + k));
            }
        }
    });
    __syncthreads();
    float acc[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) acc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        if (s0 + s < s1) {
            const float4 xa = *reinterpret_cast<const float4*>(&xs[(s0 + s) * SK + sub * LK]);
            float4 xb = xa;
            if constexpr (W16) xb = *reinterpret_cast<const float4*>(&xs[(s0 + s) * SK + sub * LK + 4]);
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
                const lm1_u32x4 u = w[r][s];
                float a = acc[r];
                if constexpr (W16) {
                    a += __uint_as_float(u[0] << 16) * xa.x;          a += __uint_as_float(u[0] & 0xffff0000u) * xa.y;
                    a += __uint_as_float(u[1] << 16) * xa.z;          a += __uint_as_float(u[1] & 0xffff0000u) * xa.w;
                    a += __uint_as_float(u[2] << 16) * xb.x;          a += __uint_as_float(u[2] & 0xffff0000u) * xb.y;
                    a += __uint_as_float(u[3] << 16) * xb.z;          a += __uint_as_float(u[3] & 0xffff0000u) * xb.w;
                } else {
                    a += __uint_as_float(u[0]) * xa.x; a += __uint_as_float(u[1]) * xa.y; a += __uint_as_float(u[2]) * xa.z; a += __uint_as_float(u[3]) * xa.w;
                }
                acc[r] = a;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) acc[r] = group16_sum(acc[r]);
    if constexpr (KSPLIT) {
#pragma unroll
        for (int r = 0; r < ROWS; ++r) if (sub == 0) part[wave][grp * ROWS + r] = acc[r];
        __syncthreads();
        if (wave != 0) return;
#pragma unroll
        for (int r = 0; r < ROWS; ++r) { float t = 0.f; for (int ww = 0; ww < NW; ++ww) t += part[ww][grp * ROWS + r]; acc[r] = t; }
    }
    if (sub != 0) return;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const int n = row0 + r;
        if (n < p.N) {
            float v = acc[r];
            if (p.bias) v += p.bias[n];
            v = apply_act(p.act, v, 0.f);
            if (p.res) v += p.res[n];
            float* y = p.layer >= 0 ? p.dyn->rows[p.layer] + (long long)p.dyn->pos * p.N : p.y;
            y[n] = v;
        }
    }
}

// LayerNorm -> activation -> * scale of one row (the input layer's norm; its output is the residual stream, so it is materialised)
struct Lm1NormArgs { const float* x; float* y; const float* g; const float* b; float eps, scale; int C, act; };
static __global__ __launch_bounds__(64) void lm1_norm_kernel(Lm1NormArgs p) {
    const int lane = threadIdx.x;
    float4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = lane * 4 + k * 256;
        v[k] = c < p.C ? *reinterpret_cast<const float4*>(p.x + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (lane * 4 + k * 256 < p.C) s += v[k].x + v[k].y + v[k].z + v[k].w;
    s = wave_sum(s);
    const float mean = s / (float)p.C;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (lane * 4 + k * 256 < p.C) { const float a = v[k].x - mean, b = v[k].y - mean, cc = v[k].z - mean, d = v[k].w - mean; q += a * a + b * b + cc * cc + d * d; }
    q = wave_sum(q);
    const float rstd = rsqrtf(q / (float)p.C + p.eps);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = lane * 4 + k * 256;
        if (c >= p.C) continue;
        float o[4] = {(v[k].x - mean) * rstd, (v[k].y - mean) * rstd, (v[k].z - mean) * rstd, (v[k].w - mean) * rstd};
        const float4 g = *reinterpret_cast<const float4*>(p.g + c); o[0] *= g.x; o[1] *= g.y; o[2] *= g.z; o[3] *= g.w;
        const float4 b = *reinterpret_cast<const float4*>(p.b + c); o[0] += b.x; o[1] += b.y; o[2] += b.z; o[3] += b.w;
        const float4 t = apply_act4(p.act, make_float4(o[0], o[1], o[2], o[3]), 0.f);
        *reinterpret_cast<float4*>(p.y + c) = make_float4(t.x * p.scale, t.y * p.scale, t.z * p.scale, t.w * p.scale);
    }
}

// One query (the row at `pos`), head h, key range s of LM1_SPLITS:  score_j = ((q + u) . k_j + (q + v) . p_(n - 1 - j)) * scale  (attention.py:300-326; for the last
// query rel_shift leaves column j = relative position n - 1 - j), partial softmax (m, l, acc[64]) over the range -> part[h][s][66].
// ONE pass: a 16-lane group owns a key (4 of the 64 dimensions per lane) and requests its k, p AND v rows together, LM1_AU keys per group in flight - up to
// 16 * LM1_AU * LM1_SPLITS = 512 keys the whole context is requested before the first score exists (the kernel is a chain of latencies, not of bytes); every
// group keeps a running (m, l, acc) over its keys, the 16 groups merge through LDS.
constexpr int LM1_AU = 4;
struct Lm1AttnArgs { const Lm1Dyn* dyn; int layer, d, heads; float scale; float* part; };
static __global__ __launch_bounds__(256) void lm1_attn_kernel(Lm1AttnArgs p) {
    __shared__ float gm[16], gl[16];
    __shared__ float gacc[16][64];
    const int h = blockIdx.x, s = blockIdx.y, tid = threadIdx.x, sub = tid & 15, kg = tid >> 4;
    const int pos = p.dyn->pos, n = pos + 1, d = p.d;
    const int per = (n + LM1_SPLITS - 1) / LM1_SPLITS, j0 = min(n, s * per), j1 = min(n, j0 + per), cnt = j1 - j0;
    const float* rows = p.dyn->rows[p.layer];
    const float* qrow = rows + (long long)pos * 4 * d + h * 64;
    const float* tab = p.dyn->tabs[p.layer] + (long long)(p.dyn->n_tab - n) * d + h * 64;
    float* out = p.part + ((long long)h * LM1_SPLITS + s) * 66;
    float4 kx[LM1_AU], px[LM1_AU], vx[LM1_AU];
    auto issue = [&](int i0) {
#pragma unroll
        for (int u = 0; u < LM1_AU; ++u) {                      // unconditional (clamped) loads: all 3 * LM1_AU in flight together
            const int j = min(j0 + min(i0 + u * 16 + kg, max(cnt - 1, 0)), pos);      // (an empty range still reads a valid row: no branch around the loads)
            const float* r = rows + (long long)j * 4 * d + h * 64 + sub * 4;
            kx[u] = *reinterpret_cast<const float4*>(r + 2 * d);
            vx[u] = *reinterpret_cast<const float4*>(r + 3 * d);
            px[u] = *reinterpret_cast<const float4*>(tab + (long long)j * d + sub * 4);
        }
    };
    float m_run = LM1_NEG_INF, l_run = 0.f;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i0 = 0;; i0 += 16 * LM1_AU) {
        issue(i0);
        // the query is requested WITH the sweep's rows (an offset the compiler cannot see through keeps these two loads inside the loop: hoisted, they are waited
        // for before the rows are requested - two round trips instead of one)
        const int zero = cv_opaque_zero();
        const float4 qu = *reinterpret_cast<const float4*>(qrow + sub * 4 + zero), qv = *reinterpret_cast<const float4*>(qrow + d + sub * 4 + zero);
        float sc[LM1_AU];
        float mt = m_run;
#pragma unroll
        for (int u = 0; u < LM1_AU; ++u) {
            float a = qu.x * kx[u].x; a += qu.y * kx[u].y; a += qu.z * kx[u].z; a += qu.w * kx[u].w;
            float b = qv.x * px[u].x; b += qv.y * px[u].y; b += qv.z * px[u].z; b += qv.w * px[u].w;
            const float sj = (group16_sum(a) + group16_sum(b)) * p.scale;         // (collectives outside the select)
            sc[u] = (i0 + u * 16 + kg < cnt) ? sj : LM1_NEG_INF;
            mt = fmaxf(mt, sc[u]);
        }
        if (mt != LM1_NEG_INF) {                                // (a group with no key in this sweep and none before keeps its empty state)
            const float alpha = (m_run == LM1_NEG_INF) ? 0.f : expf(m_run - mt);
            l_run *= alpha; acc.x *= alpha; acc.y *= alpha; acc.z *= alpha; acc.w *= alpha;
#pragma unroll
            for (int u = 0; u < LM1_AU; ++u) {
                const float e = (sc[u] == LM1_NEG_INF) ? 0.f : expf(sc[u] - mt);
                l_run += e; acc.x += e * vx[u].x; acc.y += e * vx[u].y; acc.z += e * vx[u].z; acc.w += e * vx[u].w;
            }
            m_run = mt;
        }
        if (i0 + 16 * LM1_AU >= cnt) break;
    }
    if (sub == 0) { gm[kg] = m_run; gl[kg] = l_run; }
    *reinterpret_cast<float4*>(&gacc[kg][sub * 4]) = acc;
    __syncthreads();
    if (tid >= 64) return;
    float M = LM1_NEG_INF;
#pragma unroll
    for (int g = 0; g < 16; ++g) M = fmaxf(M, gm[g]);
    float L = 0.f, a = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) {                              // fixed group order
        const float w = (gm[g] == LM1_NEG_INF) ? 0.f : expf(gm[g] - M);
        L += gl[g] * w; a += gacc[g][tid] * w;
    }
    out[2 + tid] = a;
    if (tid == 0) { out[0] = M; out[1] = L; }
}

static __global__ void lm1_bind_kernel(Lm1Dyn* dyn, Lm1Dyn v) { if (threadIdx.x == 0 && blockIdx.x == 0) *dyn = v; }

struct Lm1Layer {
    const float *ln1_g, *ln1_b, *w_qkv, *b_qkv, *w_out, *b_out, *ln2_g, *ln2_b, *w1, *b1, *w2, *b2;
};

}  // namespace cv

using namespace cv;

struct cv_lm1 {
    int n_layers = 0, d = 0, heads = 0, ffn = 0, d_in = 0, n_out = 0, act = 0, cap = 0;
    float xscale = 1.f;
    std::vector<Lm1Layer> L;
    const float *embed_w = nullptr, *embed_b = nullptr, *embed_g = nullptr, *embed_beta = nullptr, *after_g = nullptr, *after_b = nullptr, *dec_w = nullptr, *dec_b = nullptr;
    // fp16 mode of this model (cv_lm1_use_bf16): the same matrices as bf16 [N][Kp]; null = fp32
    struct L16 { const void *w_qkv = nullptr, *w_out = nullptr, *w1 = nullptr, *w2 = nullptr; };
    std::vector<L16> L16s; const void* embed_w16 = nullptr; const void* dec_w16 = nullptr; bool w16 = false;
    int wide16 = 1;                    // bf16 matrices on lm1_gemv16_kernel (16-byte loads, whole rows per lane group; option "gemv16_wide", env CV_LM1_WIDE16); 0: lm1_gemv_kernel<.., true>
    int wide32 = 1;                    // the same for the fp32 matrices of the [N >= 2048][K <= 1024] products (option "gemv_wide", env CV_LM1_WIDE)
    int rows32 = 1, rows16 = 1;        // output rows per 16-lane group of the decode GEMVs over fp32 / bf16 matrices (option "gemv_rows" / "gemv_rows16": 1 | 2 | 4; the same bits)
    DevBuf dyn, x0, x1, h, ff, part;
    // the device-resident decode loop (cv_lm1_decode_begin / cv_lm1_decode): loop state, sampling parameters, emitted tokens, the next input row, logits, injected uniforms
    DevBuf dstate, dsp, dtokens, xin, dlogits, duniforms;
    int max_tokens = 0; const float* emb_table = nullptr; DecodeState host_state{}; std::vector<int> host_tokens; int loop_open = 0;
    hipGraphExec_t graph = nullptr; float* graph_logits = nullptr; hipStream_t graph_stream = nullptr;
    hipStream_t own_stream = nullptr;          // a NULL stream argument means this (blocking) stream: it orders itself against the legacy default stream, and it can be captured
    int use_graph = 0, bound = 0;              // measured on the MI355X (profiles/r4_cv1_fused_step_timing.txt): the replayed graph is 3 % SLOWER per token than the same 72 launches issued eagerly
    long long steps = 0, graph_replays = 0;
    ~cv_lm1() { if (graph) (void)hipGraphExecDestroy(graph); if (own_stream) (void)hipStreamDestroy(own_stream); }
};

namespace {

inline int kp_of(int K) { return (K + 31) / 32 * 32; }

hipStream_t resolve(cv_lm1* m, void* s) {
    if (s) return as_stream(s);
    if (!m->own_stream) CV_HIP(hipStreamCreate(&m->own_stream));
    return m->own_stream;
}

// W16: the bf16 copy of W (cv_lm1_use_bf16) or null
void gemv(cv_lm1* m, int pro, const float* x, const float* g, const float* b, float eps, const float* W, const void* W16, const float* bias, const float* res, float* y, int layer,
          int set_pos, int N, int K, int act, hipStream_t s) {
    CV_CHECK(K % 4 == 0 && K <= LM1_MAX_K && (pro != LM1_PRO_LN || K <= 1024), "cv_lm1: a GEMV input of up to 4096 floats (1024 under the LayerNorm prologue), K % 4 == 0");
    const bool h16 = m->w16 && W16 != nullptr;
    Lm1GemvArgs a{x, g, b, eps, h16 ? reinterpret_cast<const float*>(W16) : W, (long long)kp_of(K), bias, res, y, m->dyn.as<Lm1Dyn>(), layer, set_pos, m->dstate.as<DecodeState>(), N, K, kp_of(K), act, pro};
    const int steps = (kp_of(K) + 63) / 64;
    const int rpg = h16 ? m->rows16 : m->rows32;
    const dim3 grid((unsigned)((N + 4 * rpg - 1) / (4 * rpg)));
#define LM1_GEMV(U_, NW_, W16_, RPG_) hipLaunchKernelGGL((lm1_gemv_kernel<U_, NW_, W16_, RPG_>), grid, dim3(64 * NW_), 0, s, a)
#define LM1_GEMV_R(U_, NW_, W16_) do { if (rpg == 4) LM1_GEMV(U_, NW_, W16_, 4); else if (rpg == 2) LM1_GEMV(U_, NW_, W16_, 2); else LM1_GEMV(U_, NW_, W16_, 1); } while (0)
    if (h16 && m->wide16 && kp_of(K) % 128 == 0) {             // 16-byte loads, whole rows per lane group (lm1_gemv16_kernel)
        const int st = kp_of(K) / 128;
        if (st <= 8 && N >= 2048) { hipLaunchKernelGGL((lm1_gemv16_kernel<true, 8, 1, 4, false>), dim3((unsigned)((N + 15) / 16)), dim3(256), 0, s, a); return; }
        if (st <= 8) { hipLaunchKernelGGL((lm1_gemv16_kernel<true, 2, 1, 4, true>), dim3((unsigned)((N + 3) / 4)), dim3(256), 0, s, a); return; }
        if (st <= 32 && pro == LM1_PRO_NONE) { hipLaunchKernelGGL((lm1_gemv16_kernel<true, 4, 1, 8, true>), dim3((unsigned)((N + 3) / 4)), dim3(512), 0, s, a); return; }
    }
    if (!h16 && m->wide32 && kp_of(K) % 64 == 0 && kp_of(K) <= 1024 && N >= 2048) {      // fp32 matrices, whole rows per lane group, 16 rows per workgroup
        hipLaunchKernelGGL((lm1_gemv16_kernel<false, 16, 1, 4, false>), dim3((unsigned)((N + 15) / 16)), dim3(256), 0, s, a); return;
    }
    if (h16) {
        if (steps <= 16) LM1_GEMV_R(4, 4, true);
        else if (pro == LM1_PRO_NONE && N <= 2048) LM1_GEMV_R(8, 8, true);
        else LM1_GEMV_R(8, 4, true);
        return;
    }
    if (steps <= 16) LM1_GEMV_R(4, 4, false);
    else if (pro == LM1_PRO_NONE && N <= 2048) LM1_GEMV_R(8, 8, false);
    else LM1_GEMV_R(8, 4, false);
#undef LM1_GEMV_R
#undef LM1_GEMV
}

// everything of a step after its first kernel
void step_body(cv_lm1* m, float* logits, hipStream_t s) {
    const int d = m->d;
    float* x0 = m->x0.as<float>(); float* x1 = m->x1.as<float>(); float* h = m->h.as<float>(); float* ff = m->ff.as<float>(); float* part = m->part.as<float>();
    hipLaunchKernelGGL(lm1_norm_kernel, dim3(1), dim3(64), 0, s, Lm1NormArgs{h, x0, m->embed_g, m->embed_beta, 1e-5f, m->xscale, d, m->act});
    for (int i = 0; i < m->n_layers; ++i) {
        const Lm1Layer& L = m->L[i];
        const cv_lm1::L16 H = m->w16 ? m->L16s[i] : cv_lm1::L16{};
        gemv(m, LM1_PRO_LN, x0, L.ln1_g, L.ln1_b, 1e-12f, L.w_qkv, H.w_qkv, L.b_qkv, nullptr, nullptr, i, -1, 4 * d, d, ACT_NONE, s);
        hipLaunchKernelGGL(lm1_attn_kernel, dim3(m->heads, LM1_SPLITS), dim3(256), 0, s, Lm1AttnArgs{m->dyn.as<Lm1Dyn>(), i, d, m->heads, 0.125f, part});
        gemv(m, LM1_PRO_MERGE, part, nullptr, nullptr, 0.f, L.w_out, H.w_out, L.b_out, x0, x1, -1, -1, d, d, ACT_NONE, s);
        gemv(m, LM1_PRO_LN, x1, L.ln2_g, L.ln2_b, 1e-12f, L.w1, H.w1, L.b1, nullptr, ff, -1, -1, m->ffn, d, m->act, s);
        gemv(m, LM1_PRO_NONE, ff, nullptr, nullptr, 0.f, L.w2, H.w2, L.b2, x1, x0, -1, -1, d, m->ffn, ACT_NONE, s);
    }
    gemv(m, LM1_PRO_LN, x0, m->after_g, m->after_b, 1e-5f, m->dec_w, m->dec_w16, m->dec_b, nullptr, logits, -1, -1, m->n_out, d, ACT_NONE, s);
}

}  // namespace

extern "C" {

cv_lm1* cv_lm1_create(const cv_lm1_config* c, const cv_lm1_layer_weights* layers) {
    cv_lm1* m = nullptr;
    int rc = guarded([&] {
        CV_CHECK(c && layers, "cv_lm1_create: null argument");
        CV_CHECK(c->n_layers >= 1 && c->n_layers <= LM1_MAX_LAYERS, "cv_lm1_create: 1..32 layers");
        CV_CHECK(c->d == c->heads * 64 && c->d <= 1024 && c->d % 4 == 0, "cv_lm1_create: 64-wide heads, model width up to 1024");
        CV_CHECK(c->ffn % 4 == 0 && c->ffn <= LM1_MAX_K && c->d_in % 4 == 0 && c->d_in <= LM1_MAX_K && c->n_out >= 1, "cv_lm1_create: widths");
        CV_CHECK(c->act == CV_ACT_RELU || c->act == CV_ACT_SILU || c->act == CV_ACT_NONE, "cv_lm1_create: activation");
        auto ok = [](const void* p) { return p && aligned16(p); };
        m = new cv_lm1();
        m->n_layers = c->n_layers; m->d = c->d; m->heads = c->heads; m->ffn = c->ffn; m->d_in = c->d_in; m->n_out = c->n_out; m->act = c->act; m->xscale = c->xscale;
        for (int i = 0; i < c->n_layers; ++i) {
            const cv_lm1_layer_weights& w = layers[i];
            CV_CHECK(ok(w.ln1_g) && ok(w.ln1_b) && ok(w.w_qkv) && ok(w.b_qkv) && ok(w.w_out) && ok(w.b_out) && ok(w.ln2_g) && ok(w.ln2_b) && ok(w.w1) && ok(w.b1) && ok(w.w2) && ok(w.b2),
                     "cv_lm1_create: every layer tensor present and 16B aligned");
            m->L.push_back(Lm1Layer{w.ln1_g, w.ln1_b, w.w_qkv, w.b_qkv, w.w_out, w.b_out, w.ln2_g, w.ln2_b, w.w1, w.b1, w.w2, w.b2});
        }
        CV_CHECK(ok(c->embed_w) && ok(c->embed_b) && ok(c->embed_g) && ok(c->embed_beta) && ok(c->after_g) && ok(c->after_b) && ok(c->dec_w) && ok(c->dec_b),
                 "cv_lm1_create: input layer / after_norm / decoder tensors present and 16B aligned");
        m->embed_w = c->embed_w; m->embed_b = c->embed_b; m->embed_g = c->embed_g; m->embed_beta = c->embed_beta;
        m->after_g = c->after_g; m->after_b = c->after_b; m->dec_w = c->dec_w; m->dec_b = c->dec_b;
        m->dyn.ensure(sizeof(Lm1Dyn)); m->x0.ensure((size_t)c->d * 4); m->x1.ensure((size_t)c->d * 4); m->h.ensure((size_t)c->d * 4);
        m->ff.ensure((size_t)c->ffn * 4); m->part.ensure((size_t)c->heads * LM1_SPLITS * 66 * 4);
        CV_HIP(hipMemset(m->dyn.p, 0, sizeof(Lm1Dyn)));
        if (const char* e = getenv("CV_LM1_GRAPH")) m->use_graph = atoi(e) != 0;
        if (const char* e = getenv("CV_LM1_WIDE16")) m->wide16 = atoi(e) != 0;
        if (const char* e = getenv("CV_LM1_WIDE")) m->wide32 = atoi(e) != 0;
        if (const char* e = getenv("CV_LM1_ROWS")) { const int v = atoi(e); if (v == 1 || v == 2 || v == 4) m->rows32 = v; }         // A/B knobs (options "gemv_rows" / "gemv_rows16")
        if (const char* e = getenv("CV_LM1_ROWS16")) { const int v = atoi(e); if (v == 1 || v == 2 || v == 4) m->rows16 = v; }
    });
    if (rc != 0) { delete m; return nullptr; }
    return m;
}

void cv_lm1_destroy(cv_lm1* m) { delete m; }

int cv_lm1_use_bf16(cv_lm1* m, const cv_lm1_layer_bf16* layers, const void* embed_w, const void* dec_w) {
    return guarded([&] {
        CV_CHECK(m, "cv_lm1_use_bf16: null handle");
        std::lock_guard<std::recursive_mutex> lk(runtime_lock());
        if (m->graph) { (void)hipGraphExecDestroy(m->graph); m->graph = nullptr; }       // a captured step holds the kernel choice and the matrix addresses
        if (!layers) { m->w16 = false; m->L16s.clear(); m->embed_w16 = m->dec_w16 = nullptr; return; }
        m->L16s.clear();
        for (int i = 0; i < m->n_layers; ++i) {
            const cv_lm1_layer_bf16& w = layers[i];
            CV_CHECK(w.w_qkv && w.w_out && w.w1 && w.w2 && aligned16(w.w_qkv) && aligned16(w.w_out) && aligned16(w.w1) && aligned16(w.w2), "cv_lm1_use_bf16: four 16B-aligned matrices per layer");
            cv_lm1::L16 h; h.w_qkv = w.w_qkv; h.w_out = w.w_out; h.w1 = w.w1; h.w2 = w.w2;
            m->L16s.push_back(h);
        }
        CV_CHECK((!embed_w || aligned16(embed_w)) && (!dec_w || aligned16(dec_w)), "cv_lm1_use_bf16: 16B-aligned matrices");
        m->embed_w16 = embed_w; m->dec_w16 = dec_w; m->w16 = true;
    });
}

int cv_lm1_bind(cv_lm1* m, float* const* rows, const float* const* tabs, int32_t n_tab, int32_t cap, void* stream) {
    return guarded([&] {
        CV_CHECK(m && rows && tabs, "cv_lm1_bind: null argument");
        CV_CHECK(cap >= 1 && cap <= LM1_SPLITS * LM1_MAX_KEYS_PER_SPLIT, "cv_lm1_bind: a cache of up to 32768 positions");
        CV_CHECK(n_tab >= cap, "cv_lm1_bind: the relative-position tables must cover the cache (n_tab >= cap)");
        Lm1Dyn v{};
        v.pos = 0; v.n_tab = n_tab;
        for (int i = 0; i < m->n_layers; ++i) {
            CV_CHECK(rows[i] && tabs[i] && aligned16(rows[i]) && aligned16(tabs[i]), "cv_lm1_bind: cache / table buffers present and 16B aligned");
            v.rows[i] = rows[i]; v.tabs[i] = tabs[i];
        }
        hipLaunchKernelGGL(lm1_bind_kernel, dim3(1), dim3(64), 0, resolve(m, stream), m->dyn.as<Lm1Dyn>(), v);
        m->cap = cap; m->bound = 1;
    });
}

int cv_lm1_step(cv_lm1* m, const float* x_row, int32_t pos, float* logits, void* stream) {
    return guarded([&] {
        CV_CHECK(m && x_row && logits && aligned16(x_row), "cv_lm1_step: null / unaligned argument");
        CV_CHECK(m->bound, "cv_lm1_step: cv_lm1_bind first");
        CV_CHECK(pos >= 0 && pos < m->cap, "cv_lm1_step: position beyond the bound cache");
        hipStream_t s = resolve(m, stream);
        // first kernel: the input layer's Linear on the caller's row; it publishes the position the rest of the step reads
        gemv(m, LM1_PRO_NONE, x_row, nullptr, nullptr, 0.f, m->embed_w, m->embed_w16, m->embed_b, nullptr, m->h.as<float>(), -1, pos, m->d, m->d_in, ACT_NONE, s);
        ++m->steps;
        if (!m->use_graph) { step_body(m, logits, s); return; }
        if (!m->graph || m->graph_logits != logits || m->graph_stream != s) {
            std::lock_guard<std::recursive_mutex> lk(runtime_lock());
            if (m->graph) { (void)hipGraphExecDestroy(m->graph); m->graph = nullptr; }
            hipGraph_t g = capture_graph(s, [&] { step_body(m, logits, s); });
            hipError_t e = hipGraphInstantiate(&m->graph, g, nullptr, nullptr, 0);
            (void)hipGraphDestroy(g);
            if (e != hipSuccess) { m->graph = nullptr; CV_HIP(e); }
            m->graph_logits = logits; m->graph_stream = s;
        }
        { std::lock_guard<std::recursive_mutex> lk(runtime_lock()); CV_HIP(hipGraphLaunch(m->graph, s)); }
        ++m->graph_replays;
    });
}

// ---- the decode loop on the device (round 5; reference: the python loop of TransformerLM.inference, llm/llm.py:196-223, one host round trip per token) ------------------
// cv_lm1_decode_begin parks the first input row, the position of the row it will write and the request's sampling parameters on the device; every step of
// cv_lm1_decode is then [input Linear on the parked row, publishing the position] [the 72 launches of cv_lm1_step] [sample_kernel: arg-max or repetition-aware
// sampling on the logits, eos masked below min_len, the sampled token's speech_embedding row parked as the next input] [advance]: no logits, no token and no embedding
// row cross the host boundary per step.  Tokens come back once per call (n_steps steps), like cv_llm_decode.
int cv_lm1_decode_begin(cv_lm1* m, const float* x_row, int32_t pos, const cv_sampling* sp, const float* emb_table, int32_t max_tokens, const float* host_uniforms, int32_t n_uniforms,
                        void* stream) {
    return guarded([&] {
        CV_CHECK(m && x_row && sp && emb_table && aligned16(x_row) && aligned16(emb_table), "cv_lm1_decode_begin: null / unaligned argument");
        CV_CHECK(m->bound, "cv_lm1_decode_begin: cv_lm1_bind first");
        CV_CHECK(pos >= 0 && pos < m->cap && max_tokens >= 1, "cv_lm1_decode_begin: position beyond the bound cache");
        CV_CHECK(m->n_out <= 8192, "cv_lm1_decode_begin: the device sampler holds at most 8192 logits");
        CV_CHECK(sp->max_len > 0 && sp->eos >= 0 && sp->eos + sp->n_stop <= m->n_out, "cv_lm1_decode_begin: bad sampling parameters");
        CV_CHECK(sp->mode == 0 || (sp->top_k > 0 && sp->top_k <= 64 && sp->win_size >= 0), "cv_lm1_decode_begin: bad RAS parameters");
        CV_CHECK(!sp->use_uniforms || (host_uniforms && n_uniforms >= 2), "cv_lm1_decode_begin: use_uniforms needs the uniforms");
        hipStream_t s = resolve(m, stream);
        m->dstate.ensure(sizeof(DecodeState)); m->dsp.ensure(sizeof(SampleParams)); m->dtokens.ensure((size_t)max_tokens * 4); m->xin.ensure((size_t)m->d_in * 4);
        m->dlogits.ensure((size_t)m->n_out * 4);
        m->max_tokens = max_tokens; m->emb_table = emb_table; m->host_tokens.assign(max_tokens, 0);
        DecodeState st{}; st.pos = pos; st.stop_token = -1;
        m->host_state = st;
        const SampleParams hp{sp->mode, sp->eos, sp->n_stop, sp->min_len, sp->max_len, sp->top_p, sp->top_k, sp->win_size, sp->tau_r, sp->use_uniforms, (unsigned long long)sp->seed};
        CV_HIP(hipMemcpyAsync(m->dstate.p, &st, sizeof(st), hipMemcpyHostToDevice, s));
        CV_HIP(hipMemcpyAsync(m->dsp.p, &hp, sizeof(hp), hipMemcpyHostToDevice, s));
        CV_HIP(hipMemcpyAsync(m->xin.p, x_row, (size_t)m->d_in * 4, hipMemcpyDeviceToDevice, s));
        if (sp->use_uniforms) {
            m->duniforms.ensure((size_t)2 * sp->max_len * 4);
            CV_HIP(hipMemcpyAsync(m->duniforms.p, host_uniforms, (size_t)std::min(n_uniforms, 2 * sp->max_len) * 4, hipMemcpyHostToDevice, s));
        }
        CV_HIP(hipStreamSynchronize(s));                        // the host copies above are stack / caller memory
        m->loop_open = 1;
    });
}

int cv_lm1_decode(cv_lm1* m, int32_t n_steps, int32_t* out_tokens, int32_t* n_out, int32_t* finished, void* stream) {
    return guarded([&] {
        CV_CHECK(m && out_tokens && n_out && finished && n_steps > 0, "cv_lm1_decode: bad arguments");
        CV_CHECK(m->loop_open, "cv_lm1_decode: cv_lm1_decode_begin first");
        CV_CHECK(m->host_state.done || m->host_state.pos + n_steps <= m->cap, "cv_lm1_decode: the bound cache ends before these steps do");
        hipStream_t s = resolve(m, stream);
        DecodeState* st = m->dstate.as<DecodeState>();
        const int before = m->host_state.n_tokens;
        for (int i = 0; i < n_steps; ++i) {
            gemv(m, LM1_PRO_NONE, m->xin.as<float>(), nullptr, nullptr, 0.f, m->embed_w, m->embed_w16, m->embed_b, nullptr, m->h.as<float>(), -1, -2, m->d, m->d_in, ACT_NONE, s);
            step_body(m, m->dlogits.as<float>(), s);
            SampleArgs sa{};
            sa.logits = m->dlogits.as<float>(); sa.V = m->n_out; sa.sp = m->dsp.as<SampleParams>(); sa.uniforms = m->duniforms.as<float>();
            sa.st = st; sa.tokens = m->dtokens.as<int>(); sa.max_tokens = m->max_tokens;
            sa.emb_table_f32 = m->emb_table; sa.emb_dim = m->d_in; sa.h_out = m->xin.as<float>();
            hipLaunchKernelGGL(sample_kernel, dim3(1), dim3(1024), 0, s, sa);
            hipLaunchKernelGGL(advance_pos_kernel, dim3(1), dim3(1), 0, s, st);
            ++m->steps;
        }
        CV_HIP(hipMemcpyAsync(&m->host_state, st, sizeof(DecodeState), hipMemcpyDeviceToHost, s));
        CV_HIP(hipMemcpyAsync(m->host_tokens.data(), m->dtokens.p, (size_t)m->max_tokens * 4, hipMemcpyDeviceToHost, s));
        CV_HIP(hipStreamSynchronize(s));
        CV_HIP(hipGetLastError());
        const int after = std::min(m->host_state.n_tokens, m->max_tokens);
        *n_out = std::max(0, after - before);
        for (int k = 0; k < *n_out; ++k) out_tokens[k] = m->host_tokens[before + k];
        *finished = m->host_state.done ? 1 : 0;
    });
}

int64_t cv_lm1_stat(const cv_lm1* m, const char* name) {
    if (!m || !name) return -1;
    const std::string n(name);
    if (n == "steps") return m->steps;
    if (n == "graph_replays") return m->graph_replays;
    if (n == "launches_per_step") return 3 + 5LL * m->n_layers;
    return -1;
}

int cv_lm1_set_option(cv_lm1* m, const char* name, int32_t value) {
    return guarded([&] {
        CV_CHECK(m && name, "cv_lm1_set_option: null argument");
        const std::string n(name);
        if (n == "graph") m->use_graph = value != 0;
        else if (n == "gemv16_wide" || n == "gemv_wide") {
            std::lock_guard<std::recursive_mutex> lk(runtime_lock());
            if (m->graph) { (void)hipGraphExecDestroy(m->graph); m->graph = nullptr; }
            (n == "gemv_wide" ? m->wide32 : m->wide16) = value != 0;
        }
        else if (n == "gemv_rows" || n == "gemv_rows16") {
            CV_CHECK(value == 1 || value == 2 || value == 4, "cv_lm1_set_option: gemv_rows must be 1, 2 or 4");
            std::lock_guard<std::recursive_mutex> lk(runtime_lock());
            if (m->graph) { (void)hipGraphExecDestroy(m->graph); m->graph = nullptr; }
            (n == "gemv_rows" ? m->rows32 : m->rows16) = value;
        }
        else CV_CHECK(false, "cv_lm1_set_option: unknown option");
    });
}

}  // extern "C"
