// HiFT vocoder (cosyvoice/hifigan/generator.py) helper kernels: harmonic source, STFT-16 / iSTFT-16, layout fix-ups.
// The convolution stacks run on the implicit-GEMM kernel (gemm_conv.h) with fused Snake / leaky-ReLU prologues and fused
// bias / residual / average epilogues; what is left here is HBM-bound element work.
#pragma once
#include "common.h"

namespace cv {

constexpr float CV_PI_F = 3.14159265358979323846f;

// [C][T] channel-first (API) -> [T][C] channel-last
static __global__ __launch_bounds__(256) void to_channel_last_kernel(const float* in, float* out, int C, int T) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)C * T) return;
    const int t = (int)(i / C), c = (int)(i % C);
    out[i] = in[(long long)c * T + t];
}

// SineGen2._f02sine low-rate part (generator.py:233-258): one thread per harmonic walks the m frames sequentially
// (torch.cumsum order).  f0 is piecewise constant over a frame, so the 1/480 linear down-interpolation of `rad` returns the
// frame value exactly (and never samples t = 0, where rand_ini is added: the initial phase noise has no effect on this path).
//   P[i][h] = ((cumsum_i( ((f0_i*(h+1))/sr) mod 1 ) * 2) * pi) * scale
// The walk stays sequential per harmonic (the summation order is part of the result: the phase reaches thousands of radians); what is parallel is
// everything around it: f0 is fetched in coalesced chunks and the per-frame increment ((f0 * (h + 1)) / sr) mod 1 - the expensive part, fmodf - is
// computed by all 256 threads into LDS, so the serial loop is one add and one store per frame (65 us -> a few us for 500 frames).
static __global__ __launch_bounds__(256) void hift_phase_kernel(const float* f0, float* P, int m, int H, float sr, float scale) {
    constexpr int CH = 512;                                   // frames per chunk; H <= 16 harmonics
    __shared__ float rad[CH * 16];
    const int h = threadIdx.x;
    float c = 0.f;
    for (int base = 0; base < m; base += CH) {
        const int n = min(CH, m - base);
        for (int i = threadIdx.x; i < n * H; i += 256) {
            const int fr = i / H, hh = i - fr * H;
            const float fn = f0[base + fr] * (float)(hh + 1);
            rad[fr * 16 + hh] = fmodf(fn / sr, 1.0f);
        }
        __syncthreads();
        if (h < H)
            for (int i = 0; i < n; ++i) {
                c += rad[i * 16 + h];
                P[(long long)(base + i) * H + h] = ((c * 2.f) * CV_PI_F) * scale;
            }
        __syncthreads();
    }
}

__device__ __forceinline__ float gauss01(unsigned long long seed, unsigned long long idx) {
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (idx + 1ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
    const float u1 = ((float)(z >> 40) + 1.0f) * (1.0f / 16777217.0f);
    const float u2 = (float)((z >> 16) & 0xFFFFFFull) * (1.0f / 16777216.0f);
    return sqrtf(-2.f * logf(u1)) * cosf(2.f * CV_PI_F * u2);
}

__device__ __forceinline__ float unif01(unsigned long long seed, unsigned long long idx) {
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (idx + 1ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
    return (float)(z >> 40) * (1.0f / 16777216.0f);
}

// nearest-neighbour upsampling of channel-last rows: y[t][c] = x[t / u][c]   (torch.nn.Upsample(scale_factor=u, mode='nearest'))
static __global__ __launch_bounds__(256) void repeat_rows_kernel(const float* x, float* y, long long T, int C, int u) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= T * u * C) return;
    y[i] = x[(i / C / u) * C + i % C];
}

// SineGen2.forward + SourceModuleHnNSF (generator.py:289-317, 358-375): per output sample
//   phase = linear x`scale` up-interpolation (align_corners=False) of P;  sine = sin(phase)*amp
//   uv = f0 > thr;  noise_amp = uv*sigma + (1-uv)*amp/3;  wave_h = sine*uv + noise_amp*noise[t][h]
//   s[t] = tanh( sum_h w[h]*wave_h + b )
// noise == nullptr -> in-kernel counter RNG (the reference draws torch.randn_like on the device RNG, generator.py:312)
static __global__ __launch_bounds__(256) void hift_source_kernel(const float* f0, const float* P, const float* noise, unsigned long long seed,
                                                                  const float* lw, const float* lb, float* s, int m, int H, int scale,
                                                                  float amp, float sigma, float thr, int causal) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long L = (long long)m * scale;
    if (t >= L) return;
    const float rscale = (float)(1.0 / (double)scale);
    float src = rscale * ((float)t + 0.5f) - 0.5f;
    if (src < 0.f) src = 0.f;
    int i0 = (int)src, i1 = i0 + (i0 < m - 1 ? 1 : 0);
    float l1 = src - (float)i0, l0 = 1.f - l1;
    if (causal) { i0 = i1 = (int)(t / scale); l0 = 1.f; l1 = 0.f; }     // SineGen2(causal=True): nearest phase upsampling (generator.py:255-256)
    const float f = f0[t / scale];
    const float uv = f > thr ? 1.f : 0.f;
    const float namp = uv * sigma + (1.f - uv) * amp / 3.f;
    float acc = 0.f;
    for (int h = 0; h < H; ++h) {
        const float ph = l0 * P[(long long)i0 * H + h] + l1 * P[(long long)i1 * H + h];
        // causal: the reference adds noise_amp x a fixed UNIFORM buffer (torch.rand at construction, generator.py:223-226, 310-311)
        const float nz = noise ? noise[t * H + h] : (causal ? unif01(seed, (unsigned long long)(t * H + h)) : gauss01(seed, (unsigned long long)(t * H + h)));
        const float w = sinf(ph) * amp * uv + namp * nz;
        acc += w * lw[h];
    }
    s[t] = tanhf(acc + lb[0]);
}

// torch.stft(s, 16, 4, 16, hann_periodic, center=True, pad_mode='reflect', onesided) -> [F = L/4 + 1][18] = [re(9) | im(9)]
// (generator.py:491-497, 508-509), channel-last.
static __global__ __launch_bounds__(256) void hift_stft_kernel(const float* s, float* out, long long L, long long F) {
    __shared__ float tw_c[16 * 9], tw_s[16 * 9], win[16];
    const int tid = threadIdx.x;
    if (tid < 16) win[tid] = 0.5f - 0.5f * cosf(2.f * CV_PI_F * (float)tid / 16.f);
    if (tid < 144) { const int n = tid / 9, k = tid % 9; const float a = 2.f * CV_PI_F * (float)((k * n) & 15) / 16.f; tw_c[tid] = cosf(a); tw_s[tid] = sinf(a); }
    __syncthreads();
    const long long f = (long long)blockIdx.x * 256 + tid;
    if (f >= F) return;
    float x[16];
#pragma unroll
    for (int n = 0; n < 16; ++n) {
        long long j = 4 * f - 8 + n;
        if (j < 0) j = -j;
        if (j >= L) j = 2 * (L - 1) - j;
        x[n] = s[j] * win[n];
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        float re = 0.f, im = 0.f;
#pragma unroll
        for (int n = 0; n < 16; ++n) { re += x[n] * tw_c[n * 9 + k]; im -= x[n] * tw_s[n * 9 + k]; }
        out[f * 18 + k] = re; out[f * 18 + 9 + k] = im;
    }
}

// ReflectionPad1d((1, 0)) finished in place: rows 1.. were written by the transposed conv, row 0 <- row 2 (= old row 1)
static __global__ __launch_bounds__(256) void reflect_row0_kernel(float* x, int C) {
    for (int c = threadIdx.x; c < C; c += 256) x[c] = x[2 * C + c];
}

// conv_post output [F][18] -> complex spectrum: mag = min(exp(x[:9]), 100), ph = sin(x[9:]); (re, im) = mag*(cos ph, sin ph)
// (generator.py:499-505, 534-535).  In place.
static __global__ __launch_bounds__(256) void hift_spec_kernel(float* x, long long F) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= F * 9) return;
    const long long f = i / 9; const int k = (int)(i % 9);
    const float mag = fminf(expf(x[f * 18 + k]), 100.f);
    const float ph = sinf(x[f * 18 + 9 + k]);
    x[f * 18 + k] = mag * cosf(ph);
    x[f * 18 + 9 + k] = mag * sinf(ph);
}

// torch.istft(spec, 16, 4, 16, hann_periodic) (center=True): windowed irfft frames overlap-added, divided by the squared-window
// envelope, 8 samples trimmed per side; then clamp to +-limit (generator.py:499-505, 537-538).
static __global__ __launch_bounds__(256) void hift_istft_kernel(const float* spec, float* y, long long F, long long L, float limit) {
    __shared__ float tw_c[16 * 9], tw_s[16 * 9], win[16];
    const int tid = threadIdx.x;
    if (tid < 16) win[tid] = 0.5f - 0.5f * cosf(2.f * CV_PI_F * (float)tid / 16.f);
    if (tid < 144) { const int n = tid / 9, k = tid % 9; const float a = 2.f * CV_PI_F * (float)((k * n) & 15) / 16.f; tw_c[tid] = cosf(a); tw_s[tid] = sinf(a); }
    __syncthreads();
    const long long t = (long long)blockIdx.x * 256 + tid;
    if (t >= L) return;
    const long long tp = t + 8;
    float acc = 0.f, env = 0.f;
    for (long long f = tp / 4; f >= 0 && tp - 4 * f < 16; --f) {
        if (f >= F) continue;
        const int n = (int)(tp - 4 * f);
        const float* X = spec + f * 18;
        float v = X[0] + ((n & 1) ? -X[8] : X[8]);
#pragma unroll
        for (int k = 1; k < 8; ++k) v += 2.f * (X[k] * tw_c[n * 9 + k] - X[9 + k] * tw_s[n * 9 + k]);
        acc += (v * (1.f / 16.f)) * win[n];
        env += win[n] * win[n];
    }
    float o = env > 1e-11f ? acc / env : acc;
    y[t] = fminf(fmaxf(o, -limit), limit);
}

// y[t][c] = Snake(x[t][c], alpha[c]): the operand of a ResBlock's first convolution, activated once (round 3) instead of in the prologue of every
// tap and N-tile of that convolution (an 11-tap conv over 4 N-tiles evaluated sinf 44 times per element)
static __global__ __launch_bounds__(256) void snake_rows_kernel(const float* x, float* y, const float* alpha, long long n4, int C) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;          // float4 index over [rows][C / 4]
    if (i >= n4) return;
    const int c = (int)((i * 4) % C);
    const float4 v = *reinterpret_cast<const float4*>(x + i * 4), a = *reinterpret_cast<const float4*>(alpha + c);
    *reinterpret_cast<float4*>(y + i * 4) = make_float4(snake_f(v.x, a.x), snake_f(v.y, a.y), snake_f(v.z, a.z), snake_f(v.w, a.w));
}


// Conv1d in DOUBLE over channel-last rows - the float64 mode of the f0 predictor (the reference runs CausalConvRNNF0Predictor in float64, generator.py:716-717:
// its module is converted with .to(torch.float64), i.e. fp32 weights widened exactly, and so are they here).  out[t][n] = act(b[n] + sum_{j, c} in[t + j - pad][c] *
// W[n][j * Kp + c]), rows outside [0, in_rows) read as zero.  One workgroup = 16 rows x 64 columns, k in steps of 32 through LDS; a thread owns 4 rows of one
// column.  The network is small (5 x 512 channels): ~3 GFLOP per 500 frames on the fp64 vector pipe.  act: 0 none, 1 ELU, 2 abs; in_f32 / out_f32: element type.
static __global__ __launch_bounds__(256) void conv_f64_kernel(const void* in, int in_f32, int in_rows, int Cin, const float* W, int Kp, const float* bias, void* out, int out_f32,
                                                               int M, int N, int taps, int pad, int act) {
    __shared__ double As[16][33];
    __shared__ double Ws[64][33];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int t0 = blockIdx.x * 16, n0 = blockIdx.y * 64;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int j = 0; j < taps; ++j)
        for (int c0 = 0; c0 < Cin; c0 += 32) {
            for (int e = threadIdx.x; e < 16 * 32; e += 256) {
                const int r = e >> 5, c = c0 + (e & 31), row = t0 + r + j - pad;
                double v = 0.0;
                if (row >= 0 && row < in_rows && c < Cin && t0 + r < M)
                    v = in_f32 ? (double)reinterpret_cast<const float*>(in)[(long long)row * Cin + c] : reinterpret_cast<const double*>(in)[(long long)row * Cin + c];
                As[r][e & 31] = v;
            }
            for (int e = threadIdx.x; e < 64 * 32; e += 256) {
                const int n = n0 + (e >> 5), c = c0 + (e & 31);
                Ws[e >> 5][e & 31] = (n < N && c < Cin) ? (double)W[(long long)n * taps * Kp + (long long)j * Kp + c] : 0.0;
            }
            __syncthreads();
#pragma unroll 8
            for (int kk = 0; kk < 32; ++kk) {
                const double w = Ws[tx][kk];
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] += As[ty * 4 + r][kk] * w;
            }
            __syncthreads();
        }
    const int n = n0 + tx;
    if (n >= N) return;
    for (int r = 0; r < 4; ++r) {
        const int t = t0 + ty * 4 + r;
        if (t >= M) continue;
        double v = acc[r] + (bias ? (double)bias[n] : 0.0);
        if (act == 1) v = v > 0.0 ? v : expm1(v);
        else if (act == 2) v = fabs(v);
        if (out_f32) reinterpret_cast<float*>(out)[(long long)t * N + n] = (float)v;
        else reinterpret_cast<double*>(out)[(long long)t * N + n] = v;
    }
}

}  // namespace cv
