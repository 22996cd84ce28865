// GroupNorm (+ activation, + a per-batch column vector) over channel-last activations, and the batched column concat of the U-Net's skip connections: shared by the
// operator-level entry points of csrc/cv1_ops.hip and the CosyVoice-300M estimator inside the flow handle (csrc/flow.hip, cfg.estimator == 2).
#pragma once
#include "common.h"

namespace cv {

// ---- GroupNorm (torch.nn.GroupNorm(G, C) on [B, C, T]; here channel-last x [B][T][C]) ----------------------------------------------------
// Pass 1: partial sums per (batch, group, time slice) in double (a group of the regulator is the whole [80 x T] utterance: fp32 running sums would
// lose the digits torch's two-level Welford keeps).  part[((b * G + g) * S + s) * 2 + {0, 1}] = {sum, sum of squares}.
static __global__ __launch_bounds__(256) void group_stats_kernel(const float* x, double* part, int T, int C, int G, int S) {
    __shared__ double red[2][4];
    const int s = blockIdx.x, g = blockIdx.y, b = blockIdx.z, cg = C / G;
    const int t0 = (int)((long long)T * s / S), t1 = (int)((long long)T * (s + 1) / S);
    const float* xb = x + ((long long)b * T) * C + (long long)g * cg;
    const long long n = (long long)(t1 - t0) * cg;
    double a = 0.0, q = 0.0;
    for (long long i = threadIdx.x; i < n; i += 256) {
        const int t = t0 + (int)(i / cg), c = (int)(i % cg);
        const double v = (double)xb[(long long)t * C + c];
        a += v; q += v * v;
    }
    // fixed-order reduction: lanes of a wave through shuffles, then the four waves through LDS
    for (int off = 32; off >= 1; off >>= 1) { a += __shfl_xor(a, off); q += __shfl_xor(q, off); }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { red[0][w] = a; red[1][w] = q; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double* o = part + (((long long)b * G + g) * S + s) * 2;
        o[0] = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];
        o[1] = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
    }
}

// Pass 2: y = act((x - mean_g) * rstd_g * gamma[c] + beta[c]) + col_add[b][c]
//   (Block1D: Conv1d -> GroupNorm -> Mish, matcha decoder.py; the ResnetBlock1D's time projection `h += mlp(t_emb)[:, :, None]` rides along as col_add)
static __global__ __launch_bounds__(256) void group_apply_kernel(const float* x, float* y, const double* part, int T, int C, int G, int S,
                                                                  const float* gamma, const float* beta, float eps, int act,
                                                                  const float* col_add, long long col_add_batch, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C), cg = C / G, g = c / cg;
    const int b = (int)(i / ((long long)T * C));
    const double* p = part + ((long long)b * G + g) * S * 2;
    double a = 0.0, q = 0.0;
    for (int s = 0; s < S; ++s) { a += p[2 * s]; q += p[2 * s + 1]; }
    const double n = (double)T * cg, mean = a / n;
    double var = q / n - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    float v = (x[i] - (float)mean) * rstd;
    if (gamma) v *= gamma[c];
    if (beta) v += beta[c];
    v = apply_act(act, v, 0.f);
    if (col_add) v += col_add[(long long)b * col_add_batch + c];
    y[i] = v;
}

// Pass 1 for the shapes of the U-Net estimator (C % 4 == 0, 256 % (C / 4) == 0, a group = a multiple of 4 channels): one workgroup per (time slice, batch) reads whole
// rows as float4 (a row of 256 channels = 1 KB, fully coalesced; the kernel above walks a 128-byte group column and measured 13.8 us per launch at [2][1000][256])
// and reduces ALL groups of its slice: lane (row r, float4 j) sums its four channels over rows r, r + R, ..., the lanes of a group are then added in a fixed order
// through LDS.  Same `part` layout as group_stats_kernel.
static __global__ __launch_bounds__(256) void group_stats_rows_kernel(const float* x, double* part, int T, int C, int G, int S) {
    __shared__ double red[2][256];
    const int s = blockIdx.x, b = blockIdx.y, q = C >> 2, R = 256 / q, cg4 = (C / G) >> 2;
    const int j = threadIdx.x % q, r = threadIdx.x / q;
    const int t0 = (int)((long long)T * s / S), t1 = (int)((long long)T * (s + 1) / S);
    const float4* xb = reinterpret_cast<const float4*>(x + (long long)b * T * C);
    double a = 0.0, sq = 0.0;
    for (int t = t0 + r; t < t1; t += R) {
        const float4 v = xb[(long long)t * q + j];
        const double v0 = v.x, v1 = v.y, v2 = v.z, v3 = v.w;
        a += (v0 + v1) + (v2 + v3); sq += (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3);
    }
    red[0][threadIdx.x] = a; red[1][threadIdx.x] = sq;
    __syncthreads();
    if ((int)threadIdx.x < G) {
        const int g = threadIdx.x;
        double sa = 0.0, ss = 0.0;
        for (int rr = 0; rr < R; ++rr)
            for (int jj = 0; jj < cg4; ++jj) { sa += red[0][rr * q + g * cg4 + jj]; ss += red[1][rr * q + g * cg4 + jj]; }
        double* o = part + (((long long)b * G + g) * S + s) * 2;
        o[0] = sa; o[1] = ss;
    }
}

// Pass 2 for the same shapes: lane (row r, float4 j) owns four channels of ONE group - it reduces the group's partial sums once and then walks its rows with float4
// loads / stores (the kernel above redoes the double-precision reduction per element: 7.3 us per launch at [2][1011][256] against 2 MB in + 2 MB out).  Same expressions
// per element as group_apply_kernel: the same bits.
static __global__ __launch_bounds__(256) void group_apply_rows_kernel(const float* x, float* y, const double* part, int T, int C, int G, int S, int rows_per_wg,
                                                                       const float* gamma, const float* beta, float eps, int act,
                                                                       const float* col_add, long long col_add_batch) {
    __shared__ double ps[2 * 8 * 32];                        // the batch row's partial sums [G][S][2] (G <= 8, S <= 32: see group_norm()), then the totals behind them
    __shared__ double tot[2 * 8];
    const int b = blockIdx.y, q = C >> 2, R = 256 / q, cg = C / G;
    const int j = threadIdx.x % q, r = threadIdx.x / q, c = 4 * j, g = c / cg;
    // every lane needs ITS group's totals: S dependent loads per lane were what this launch took (11.5 us at S = 32).  The workgroup fetches the G * S pairs once, 2 G lanes
    // add them in slice order (the order of group_apply_kernel: the same bits), everybody reads the result.
    const double* p = part + (long long)b * G * S * 2;
    for (int i = threadIdx.x; i < G * S * 2; i += 256) ps[i] = p[i];
    __syncthreads();
    if ((int)threadIdx.x < 2 * G) {
        const int gg = threadIdx.x >> 1, which = threadIdx.x & 1;
        double acc = 0.0;
        for (int s = 0; s < S; ++s) acc += ps[(gg * S + s) * 2 + which];
        tot[threadIdx.x] = acc;
    }
    __syncthreads();
    const double a = tot[2 * g], sq = tot[2 * g + 1];
    const double n = (double)T * cg, mean = a / n;
    double var = sq / n - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps)), fm = (float)mean;
    const float4 gm = gamma ? *reinterpret_cast<const float4*>(gamma + c) : make_float4(1.f, 1.f, 1.f, 1.f);
    const float4 bt = beta ? *reinterpret_cast<const float4*>(beta + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 ca = col_add ? *reinterpret_cast<const float4*>(col_add + (long long)b * col_add_batch + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    const int t0 = blockIdx.x * rows_per_wg, t1 = min(T, t0 + rows_per_wg);
    const float4* xb = reinterpret_cast<const float4*>(x + (long long)b * T * C);
    float4* yb = reinterpret_cast<float4*>(y + (long long)b * T * C);
    for (int t = t0 + r; t < t1; t += R) {
        const float4 v = xb[(long long)t * q + j];
        float o[4] = {(v.x - fm) * rstd, (v.y - fm) * rstd, (v.z - fm) * rstd, (v.w - fm) * rstd};
        if (gamma) { o[0] *= gm.x; o[1] *= gm.y; o[2] *= gm.z; o[3] *= gm.w; }
        if (beta) { o[0] += bt.x; o[1] += bt.y; o[2] += bt.z; o[3] += bt.w; }
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = apply_act(act, o[k], 0.f);
        if (col_add) { o[0] += ca.x; o[1] += ca.y; o[2] += ca.z; o[3] += ca.w; }
        yb[(long long)t * q + j] = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// y = act(GroupNorm(x)) + col_add[b]; workspace: B * G * 64 doubles
static inline void group_norm(const float* x, float* y, int B, int T, int C, int G, const float* gamma, const float* beta, float eps, int act,
                              const float* col_add, long long col_add_batch, double* workspace, hipStream_t s) {
    const int q = C / 4;
    int S;
    const bool al16 = (((uintptr_t)x | (uintptr_t)y | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)col_add) & 15) == 0 && col_add_batch % 4 == 0;
    if (C % 4 == 0 && q <= 256 && 256 % q == 0 && (C / G) % 4 == 0 && G <= 8 && al16) {
        const int R = 256 / q;
        S = (T + 8 * R - 1) / (8 * R);                                    // ~8 rows per lane
        S = S < 1 ? 1 : (S > 32 ? 32 : S);
        hipLaunchKernelGGL(group_stats_rows_kernel, dim3(S, B), dim3(256), 0, s, x, workspace, T, C, G, S);
        const int rows = 4 * R;                                           // 4 rows per lane
        hipLaunchKernelGGL(group_apply_rows_kernel, dim3((unsigned)((T + rows - 1) / rows), B), dim3(256), 0, s, x, y, workspace, T, C, G, S, rows, gamma, beta, eps, act,
                           col_add, col_add_batch);
        return;
    } else {
        S = (int)(((long long)T * (C / G) + 8191) / 8192);               // ~8 K elements per workgroup of pass 1
        S = S < 1 ? 1 : (S > 32 ? 32 : S);
        hipLaunchKernelGGL(group_stats_kernel, dim3(S, G, B), dim3(256), 0, s, x, workspace, T, C, G, S);
    }
    const long long total = (long long)B * T * C;
    hipLaunchKernelGGL(group_apply_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, y, workspace, T, C, G, S, gamma, beta, eps, act, col_add, col_add_batch, total);
}

// out[b][t][0:ca] = a[b][t], out[b][t][ca:ca+cb] = bb[b][t] for t < T; a / bb have their own per-batch pitches (rows per batch may exceed T:
// the up-sampled stream of the U-Net is cut to the skip connection's length, flow/decoder.py:275)
static __global__ __launch_bounds__(256) void concat_cols_batched_kernel(const float* a, int ca, long long a_batch, const float* bb, int cb, long long b_batch,
                                                                          float* out, int T, int B) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const int w = ca + cb;
    if (i >= (long long)B * T * w) return;
    const int c = (int)(i % w), t = (int)((i / w) % T), b = (int)(i / ((long long)w * T));
    out[i] = c < ca ? a[(long long)b * a_batch + (long long)t * ca + c] : bb[(long long)b * b_batch + (long long)t * cb + (c - ca)];
}

}  // namespace cv
