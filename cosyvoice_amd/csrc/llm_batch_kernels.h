// Lock-step batched decode (BASELINE.json configs[2]/[3], SURVEY.md §8e "LLM continuous batching"): NB sequences advance by one token
// per step and every weight matrix is streamed from HBM ONCE per step for all of them - the batch-1 GEMVs are bound by that stream
// (llm_kernels.h), so the cost of a step grows only by the per-sequence activations.
//
// The arithmetic of each sequence is the arithmetic of the single-sequence kernels (same lane mapping, same summation order), so a
// sequence decoded in a batch yields bit-identical logits - and therefore the same tokens - as the same sequence decoded alone.
// Per-sequence kernels that do not touch weights (sampler, embedding of the sampled token, position advance) are the single-sequence
// kernels launched per slot; the weight-streaming GEMVs and the attention get batched kernels here.
#pragma once
#include "llm_kernels.h"

namespace cv {

constexpr int MAX_NB = 8;

struct GemvBatchArgs {
    const bf16_t* W; const float* bias; const float* x; long long ldx; float* y; long long ldy; int N, K;
    const float* gamma; float eps; const float* res; long long ldres; int mode; int nb;
    const float* part; long long ldpart;      // NSP > 0: x of slot b = merge of its split-attention partials at part + b * ldpart
};

// gemv_kernel (llm_kernels.h) with a loop over the nb sequences around everything that depends on x: the weight registers are
// loaded once, x_b / partials_b come from L2 per sequence.
template <int STEPS, int ROWS, int WAVES, int NSP = 0>
__global__ __launch_bounds__(WAVES * 64) void gemv_batch_kernel(GemvBatchArgs p) {
    __shared__ float part[WAVES][4][ROWS][MAX_NB];
    __shared__ __attribute__((aligned(16))) float xs[NSP > 0 ? 1024 : 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, grp = lane >> 4, sub = lane & 15;
    const int steps = p.K / 128;
    const int s0 = wave * steps / WAVES, s1 = (wave + 1) * steps / WAVES;
    const int row0 = (blockIdx.x * 4 + grp) * ROWS;

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
    float4 ga[STEPS], gb[STEPS];
    if (p.gamma) {
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            const int so = (s0 + s < s1) ? (s0 + s) : s0;
            ga[s] = *reinterpret_cast<const float4*>(p.gamma + so * 128 + sub * 8);
            gb[s] = *reinterpret_cast<const float4*>(p.gamma + so * 128 + sub * 8 + 4);
        }
    }
    // the loops over the sequences are fully unrolled (compile-time b, uniform `b < nb` guards): acc[][] stays in registers
    float acc[MAX_NB][ROWS];
#pragma unroll
    for (int b = 0; b < MAX_NB; ++b) {
        if (b >= p.nb) break;
        float4 xa[STEPS], xb[STEPS];
        if constexpr (NSP == 0) {
            const float* xp = p.x + (long long)b * p.ldx;
#pragma unroll
            for (int s = 0; s < STEPS; ++s) {
                const bool ok = s0 + s < s1;
                const int so = ok ? (s0 + s) : s0;
                xa[s] = *reinterpret_cast<const float4*>(xp + so * 128 + sub * 8);
                xb[s] = *reinterpret_cast<const float4*>(xp + so * 128 + sub * 8 + 4);
                if (!ok) { xa[s] = make_float4(0.f, 0.f, 0.f, 0.f); xb[s] = xa[s]; }
            }
        } else {
            static_assert(NSP == 0 || WAVES == 4, "partial-combine prologue: 256 threads cover K <= 1024");
            if (b > 0) __syncthreads();                       // xs of the previous sequence has been consumed
            if (tid * 4 < p.K) {
                const float* ph = p.part + (long long)b * p.ldpart + (long long)(tid >> 4) * NSP * ATTN_PART;
                float4 pa[NSP > 0 ? NSP : 1]; float2 ml[NSP > 0 ? NSP : 1];
#pragma unroll
                for (int q = 0; q < NSP; ++q) {
                    pa[q] = *reinterpret_cast<const float4*>(ph + q * ATTN_PART + (tid & 15) * 4);
                    ml[q] = *reinterpret_cast<const float2*>(ph + q * ATTN_PART + 64);
                }
                float M = ml[0].x;
#pragma unroll
                for (int q = 1; q < NSP; ++q) M = fmaxf(M, ml[q].x);
                float den = 0.f; float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int q = 0; q < NSP; ++q) {
                    const float wq = (ml[q].y > 0.f) ? expf(ml[q].x - M) : 0.f;
                    den += wq * ml[q].y;
                    a.x += wq * pa[q].x; a.y += wq * pa[q].y; a.z += wq * pa[q].z; a.w += wq * pa[q].w;
                }
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
        if (p.gamma) {                                        // fused Qwen2RMSNorm of sequence b (WAVES == 1)
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
            acc[b][r] = group16_sum(a);
        }
    }
    if (WAVES > 1) {                                          // split-K inside the workgroup, combined in fixed order (as gemv_kernel)
        if (sub == 0) {
#pragma unroll
            for (int b = 0; b < MAX_NB; ++b)
                if (b < p.nb) {
#pragma unroll
                    for (int r = 0; r < ROWS; ++r) part[wave][grp][r][b] = acc[b][r];
                }
        }
        __syncthreads();
        if (wave != 0) return;
#pragma unroll
        for (int b = 0; b < MAX_NB; ++b)
            if (b < p.nb) {
#pragma unroll
                for (int r = 0; r < ROWS; ++r) { float t = 0.f; for (int ww = 0; ww < WAVES; ++ww) t += part[ww][grp][r][b]; acc[b][r] = t; }
            }
    }
    if (sub != 0) return;
#pragma unroll
    for (int b = 0; b < MAX_NB; ++b) {
        if (b >= p.nb) break;
        float* yb = p.y + (long long)b * p.ldy;
        if (p.mode == 1) {                                    // ROWS == 2: (gate_j, up_j)
            const int j = blockIdx.x * 4 + grp;
            if (row0 + 1 < p.N) { const float g = acc[b][0]; yb[j] = (g / (1.f + expf(-g))) * acc[b][ROWS - 1]; }
        } else {
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
                const int row = row0 + r;
                if (row < p.N) {
                    float v = acc[b][r];
                    if (p.bias) v += p.bias[row];
                    if (p.res) v += p.res[(long long)b * p.ldres + row];
                    yb[row] = v;
                }
            }
        }
    }
}

// attn_decode_kernel (llm_kernels.h) with the sequence index in blockIdx.y: own qkv row, own KV cache region, own state, own partials
struct AttnDecodeBatchArgs {
    const float* qkv; long long ldqkv; float* kcache; float* vcache; long long cache_stride;      // per-sequence strides
    const float* rope_cos; const float* rope_sin; int heads, kv_heads, max_len;
    const DecodeState* st; float* part; long long ldpart; int nsplit;
};

static __global__ __launch_bounds__(64) void attn_decode_batch_kernel(AttnDecodeBatchArgs p) {
    constexpr int NS = 12, PASS = 4 * NS;
    const int b = blockIdx.y;
    const int lane = threadIdx.x, sub = lane & 15, grp = lane >> 4;
    const int h = blockIdx.x / p.nsplit, sp = blockIdx.x % p.nsplit, gsz = p.heads / p.kv_heads, g = h / gsz;
    const DecodeState* st = p.st + b;
    const float* qkv = p.qkv + (long long)b * p.ldqkv;
    float* kcache = p.kcache + (long long)b * p.cache_stride;
    float* vcache = p.vcache + (long long)b * p.cache_stride;
    const int pos = st->pos;
    const int L = pos + 1;
    const int per = ((L + p.nsplit * 4 - 1) / (p.nsplit * 4)) * 4;
    const int kb = sp * per, ke = min(L, kb + per);
    const float* kc = kcache + (long long)g * p.max_len * 64;
    const float* vc = vcache + (long long)g * p.max_len * 64;
    float4 k4[NS], v4[NS];
    auto load_pass = [&](int base) {
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) {
            const int j = base + sl * 4 + grp;
            const long long o = (long long)((j < pos && j < ke) ? j : 0) * 64 + sub * 4;
            k4[sl] = *reinterpret_cast<const float4*>(kc + o);
            v4[sl] = *reinterpret_cast<const float4*>(vc + o);
        }
    };
    load_pass(kb);
    const float* qraw = qkv + h * 64;
    const float* kq = qkv + p.heads * 64 + g * 64;
    const float* vq = qkv + (p.heads + p.kv_heads) * 64 + g * 64;
    const int d0 = sub * 4, dp = (d0 + 32) & 63;
    const float4 c4 = *reinterpret_cast<const float4*>(p.rope_cos + pos * 32 + (d0 & 31));
    const float4 s4 = *reinterpret_cast<const float4*>(p.rope_sin + pos * 32 + (d0 & 31));
    const float4 qa = *reinterpret_cast<const float4*>(qraw + d0), qb = *reinterpret_cast<const float4*>(qraw + dp);
    const float4 ka = *reinterpret_cast<const float4*>(kq + d0), kp = *reinterpret_cast<const float4*>(kq + dp);
    const float4 vn4 = *reinterpret_cast<const float4*>(vq + d0);
    const float sg = d0 < 32 ? -1.f : 1.f;
    const float4 q4 = make_float4(qa.x * c4.x + sg * qb.x * s4.x, qa.y * c4.y + sg * qb.y * s4.y, qa.z * c4.z + sg * qb.z * s4.z, qa.w * c4.w + sg * qb.w * s4.w);
    const float4 kn4 = make_float4(ka.x * c4.x + sg * kp.x * s4.x, ka.y * c4.y + sg * kp.y * s4.y, ka.z * c4.z + sg * kp.z * s4.z, ka.w * c4.w + sg * kp.w * s4.w);
    if (!st->done && sp == 0 && h % gsz == 0 && grp == 0) {
        *reinterpret_cast<float4*>(kcache + ((long long)g * p.max_len + pos) * 64 + d0) = kn4;
        *reinterpret_cast<float4*>(vcache + ((long long)g * p.max_len + pos) * 64 + d0) = vn4;
    }
    const float NEG = -__builtin_huge_valf();
    float m_run = NEG, l_run = 0.f;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int base = kb; base < ke; base += PASS) {
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
    float* pr = p.part + (long long)b * p.ldpart + (long long)blockIdx.x * ATTN_PART;
    if (grp == 0) *reinterpret_cast<float4*>(pr + d0) = acc;
    if (lane == 0) { pr[64] = (l_run > 0.f) ? m_run : 0.f; pr[65] = l_run; }
}

}  // namespace cv
