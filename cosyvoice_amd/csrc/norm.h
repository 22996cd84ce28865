// Row-wise LayerNorm / RMSNorm (fp32), one wave per row, 4 rows per workgroup. HBM/L2-bound:
// each row is read with 16B/lane coalesced loads; statistics are two-pass (mean, then centred variance)
// like torch.nn.LayerNorm, reduced with fixed-order wave shuffles.
// Rows of up to 1024 channels (every norm on the synthesis path: 256 / 512 / 896) stay in registers: ONE load pass, the
// second statistic and the affine + activation epilogue run on the registers, gamma / beta / col_add / y move as float4.
#pragma once
#include "common.h"

namespace cv {

struct NormArgs {
    const float* x; float* y; long long rows; int C;
    const float* gamma; const float* beta; float eps; int rms;
    int act; float scale; const float* row_scale; const float* col_add; long long rows_per_batch;
    long long gb_batch = 0;     // float offset of gamma / beta per batch of rows_per_batch rows (adaLN modulation: LN(x) * (1 + scale[b]) + shift[b]); 0 = shared
    unsigned short* y16 = nullptr;   // optional: the result rounded to bf16 (what a bf16-MFMA consumer would round it to when staging it), rows of C; y may then be null (C % 4 == 0, C <= 1024)
};

static __global__ __launch_bounds__(256) void norm_rows_kernel(NormArgs p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long row = (long long)blockIdx.x * 4 + wave;
    if (row >= p.rows) return;
    const float* xr = p.x + row * p.C;
    float* yr = p.y + row * p.C;
    const long long gbo = p.gb_batch ? (row / p.rows_per_batch) * p.gb_batch : 0;
    const float* gamma_p = p.gamma ? p.gamma + gbo : nullptr;
    const float* beta_p = p.beta ? p.beta + gbo : nullptr;
    const bool vec = (p.C & 3) == 0;
    if (vec && p.C <= 1024) {
        float4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = lane * 4 + k * 256;
            v[k] = c < p.C ? *reinterpret_cast<const float4*>(xr + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (lane * 4 + k * 256 < p.C) s += (p.rms ? v[k].x * v[k].x + v[k].y * v[k].y + v[k].z * v[k].z + v[k].w * v[k].w : v[k].x + v[k].y + v[k].z + v[k].w);
        s = wave_sum(s);
        float mean = 0.f, rstd;
        if (p.rms) {
            rstd = rsqrtf(s / (float)p.C + p.eps);
        } else {
            mean = s / (float)p.C;
            float q = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (lane * 4 + k * 256 < p.C) { const float a = v[k].x - mean, b = v[k].y - mean, cc = v[k].z - mean, d = v[k].w - mean; q += a * a + b * b + cc * cc + d * d; }
            q = wave_sum(q);
            rstd = rsqrtf(q / (float)p.C + p.eps);
        }
        const float rs = p.scale * (p.row_scale ? p.row_scale[row] : 1.f);
        const float* ca = p.col_add ? p.col_add + (row / p.rows_per_batch) * p.C : nullptr;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = lane * 4 + k * 256;
            if (c >= p.C) continue;
            float o[4] = {(v[k].x - mean) * rstd, (v[k].y - mean) * rstd, (v[k].z - mean) * rstd, (v[k].w - mean) * rstd};
            if (gamma_p) { const float4 g = *reinterpret_cast<const float4*>(gamma_p + c); o[0] *= g.x; o[1] *= g.y; o[2] *= g.z; o[3] *= g.w; }
            if (beta_p) { const float4 b = *reinterpret_cast<const float4*>(beta_p + c); o[0] += b.x; o[1] += b.y; o[2] += b.z; o[3] += b.w; }
            { const float4 t = apply_act4(p.act, make_float4(o[0], o[1], o[2], o[3]), 0.f); o[0] = t.x * rs; o[1] = t.y * rs; o[2] = t.z * rs; o[3] = t.w * rs; }
            if (ca) { const float4 a = *reinterpret_cast<const float4*>(ca + c); o[0] += a.x; o[1] += a.y; o[2] += a.z; o[3] += a.w; }
            if (p.y) *reinterpret_cast<float4*>(yr + c) = make_float4(o[0], o[1], o[2], o[3]);
            if (p.y16) *reinterpret_cast<uint2*>(p.y16 + row * p.C + c) = make_uint2(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]));
        }
        return;
    }
    float s = 0.f;
    if (vec) { for (int c = lane * 4; c < p.C; c += 256) { const float4 v = *reinterpret_cast<const float4*>(xr + c); s += (p.rms ? v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w : v.x + v.y + v.z + v.w); } }
    else     { for (int c = lane; c < p.C; c += 64) { const float v = xr[c]; s += p.rms ? v * v : v; } }
    s = wave_sum(s);
    float mean = 0.f, rstd;
    if (p.rms) {
        rstd = rsqrtf(s / (float)p.C + p.eps);
    } else {
        mean = s / (float)p.C;
        float q = 0.f;
        if (vec) { for (int c = lane * 4; c < p.C; c += 256) { const float4 v = *reinterpret_cast<const float4*>(xr + c); const float a = v.x - mean, b = v.y - mean, cc = v.z - mean, d = v.w - mean; q += a * a + b * b + cc * cc + d * d; } }
        else     { for (int c = lane; c < p.C; c += 64) { const float a = xr[c] - mean; q += a * a; } }
        q = wave_sum(q);
        rstd = rsqrtf(q / (float)p.C + p.eps);
    }
    const float rs = p.scale * (p.row_scale ? p.row_scale[row] : 1.f);
    const float* ca = p.col_add ? p.col_add + (row / p.rows_per_batch) * p.C : nullptr;
    for (int c = lane; c < p.C; c += 64) {
        float v = (xr[c] - mean) * rstd;
        if (gamma_p) v *= gamma_p[c];
        if (beta_p) v += beta_p[c];
        v = apply_act(p.act, v, 0.f) * rs;
        if (ca) v += ca[c];
        yr[c] = v;
    }
}

}  // namespace cv
