// Small HBM-bound helper kernels of the flow stage (token -> mel): positional tables, layout packing, CFG+Euler update.
// All activations are channel-last ([time][channel]) so LayerNorm-over-channels is a row norm and every conv / linear
// is the implicit GEMM of gemm_conv.h; the reference's b c t <-> b t c rearranges (flow/decoder.py:438,450) disappear.
#pragma once
#include "common.h"

namespace cv {

// EspnetRelPositionalEncoding table for `T` positions (transformer/embedding.py:225-302, offset 0):
// row m in [0, 2T-1) encodes relative position p = T-1-m;  pe[m][2i] = sin(p*div_i), pe[m][2i+1] = cos(p*div_i).
static __global__ __launch_bounds__(256) void rel_pos_emb_kernel(float* pe, int T, int d) {
    const int m = blockIdx.x;
    const float p = (float)(T - 1 - m);
    for (int i = threadIdx.x; i < d / 2; i += 256) {
        const float div = expf((float)(2 * i) * -(9.210340371976184f / (float)d));      // ln(10000)
        pe[(long long)m * d + 2 * i] = sinf(p * div);
        pe[(long long)m * d + 2 * i + 1] = cosf(p * div);
    }
}

// q_u = q + pos_bias_u, q_v = q + pos_bias_v   (transformer/attention.py:305-308); q is the first d columns of qkv rows
static __global__ __launch_bounds__(256) void add_pos_bias_kernel(const float* qkv, int ld, const float* u, const float* v, float* qu, float* qv, int T, int d) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)T * d) return;
    const int t = (int)(i / d), c = (int)(i % d);
    const float q = qkv[(long long)t * ld + c];
    qu[i] = q + u[c]; qv[i] = q + v[c];
}

// nearest x2 upsample along time (upsample_encoder.py:60)
static __global__ __launch_bounds__(256) void upsample2x_kernel(const float* x, float* y, int T, int d) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= 2LL * T * d) return;
    const int t2 = (int)(i / d), c = (int)(i % d);
    y[i] = x[(long long)(t2 >> 1) * d + c];
}

// F.normalize(x, dim=1): x / max(||x||_2, 1e-12)   (flow/flow.py:248), one row
static __global__ __launch_bounds__(256) void l2_normalize_kernel(const float* x, float* y, int n) {
    __shared__ float red[16];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += x[i] * x[i];
    s = block_sum(s, red);
    const float inv = 1.f / fmaxf(sqrtf(s), 1e-12f);
    for (int i = threadIdx.x; i < n; i += 256) y[i] = x[i] * inv;
}

// matcha SinusoidalPosEmb(dim)(t, scale=1000): [sin(1000 t e_i) | cos(1000 t e_i)], e_i = exp(-i ln(1e4)/(half-1))
static __global__ __launch_bounds__(256) void time_sinusoid_kernel(const float* t, float* out, int n, int dim) {
    const int r = blockIdx.x, half = dim / 2;
    const float tv = t[r];
    for (int i = threadIdx.x; i < half; i += 256) {
        const float e = expf((float)i * -(9.210340371976184f / (float)(half - 1)));
        const float a = 1000.f * tv * e;
        out[(long long)r * dim + i] = sinf(a);
        out[(long long)r * dim + half + i] = cosf(a);
    }
}

// Estimator input  h[z][t][0:4*mel] = [x | mu | spks | cond]  (flow/decoder.py:425-431: pack([x, mu]), spks, cond)
//   cl = 1: internal solve_euler form - nu utterances of equal length: x, mu, cond channel-last [nu][T][mel], spks [nu][mel]; batch row
//           z = g * nu + u is utterance u of the conditioned (g = 0) or unconditioned (g = 1: zeros for mu / spks / cond) half of the
//           classifier-free guidance pair (flow_matching.py:101-108)
//   cl = 0: API form (boundary B3, nu = 1) - x, mu, cond [2][mel][T], spks [2][mel]
static __global__ __launch_bounds__(256) void pack_est_input_kernel(const float* x, const float* mu, const float* spks, const float* cond,
                                                                     float* h, int T, int mel, int cl, int nu) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long per = (long long)T * 4 * mel;
    if (i >= 2 * nu * per) return;
    const int z = (int)(i / per);
    const long long r = i % per;
    const int t = (int)(r / (4 * mel)), c4 = (int)(r % (4 * mel)), seg = c4 / mel, c = c4 % mel;
    float v;
    if (cl) {
        const int g = z / nu, u = z % nu;
        const long long tc = ((long long)u * T + t) * mel + c;
        if (seg == 0) v = x[tc];
        else if (g == 1) v = 0.f;
        else v = seg == 1 ? mu[tc] : (seg == 2 ? spks[u * mel + c] : cond[tc]);
    } else {
        const long long cf = ((long long)z * mel + c) * T + t;
        v = seg == 0 ? x[cf] : (seg == 1 ? mu[cf] : (seg == 2 ? spks[z * mel + c] : cond[cf]));
    }
    h[i] = v;
}

// x += dt * ((1 + r) * d[0] - r * d[1])      (flow_matching.py:116-118), x,d channel-last
static __global__ __launch_bounds__(256) void cfg_euler_kernel(float* x, const float* d, long long n, float dt, float rate) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float v = (1.0f + rate) * d[i] - rate * d[n + i];
    x[i] = x[i] + dt * v;
}

// fp32 -> bf16 (round to nearest even), 8 values per thread: the operand of the large-M convolutions (flow_big.h), rounded once where the small bf16 tiles
// round it every time they stage it
static __global__ __launch_bounds__(256) void cvt_bf16_kernel(const float* x, unsigned short* y, long long n8) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const float4 a = *reinterpret_cast<const float4*>(x + 8 * i), b = *reinterpret_cast<const float4*>(x + 8 * i + 4);
    *reinterpret_cast<uint4*>(y + 8 * i) = make_uint4(pack_bf16x2(a.x, a.y), pack_bf16x2(a.z, a.w), pack_bf16x2(b.x, b.y), pack_bf16x2(b.z, b.w));
}

// out[r][0:ca] = a[r], out[r][ca:ca+cb] = b[r]   (skip connection concat, flow/decoder.py:476)
static __global__ __launch_bounds__(256) void concat_cols_kernel(const float* a, int ca, const float* b, int cb, float* out, long long rows) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const int w = ca + cb;
    if (i >= rows * w) return;
    const long long r = i / w; const int c = (int)(i % w);
    out[i] = c < ca ? a[r * ca + c] : b[r * cb + (c - ca)];
}

// channel-last [B][T][C] (rows t0..T) -> channel-first [B][C][T - t0], optional per-(b,t) mask  (API layouts of B3 / B5)
static __global__ __launch_bounds__(256) void to_channel_first_kernel(const float* in, float* out, int B, int T, int C, int t0, const float* mask) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const int Tn = T - t0;
    if (i >= (long long)B * C * Tn) return;
    const int t = (int)(i % Tn), c = (int)((i / Tn) % C), b = (int)(i / ((long long)Tn * C));
    float v = in[((long long)b * T + t0 + t) * C + c];
    if (mask) v *= mask[(long long)b * T + t0 + t];
    out[i] = v;
}

// rows [0,n) of dst (row pitch C) <- src rows; rows [n, total) <- 0      (conds = [prompt_feat ; 0], flow/flow.py:266-268)
static __global__ __launch_bounds__(256) void copy_rows_zero_tail_kernel(const float* src, float* dst, long long n_copy, long long n_total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_total) return;
    dst[i] = i < n_copy ? src[i] : 0.f;
}

}  // namespace cv
