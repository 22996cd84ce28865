// Operators that only the first-generation CosyVoice-300M path needs (host side: cosyvoice_amd/cosyvoice1_hip.py):
// GroupNorm + Mish of the Matcha Block1D / InterpolateRegulator, the type-1 SineGen harmonic source of the 22.05 kHz HiFTGenerator, and the
// layout / update helpers of the non-causal ConditionalCFM solve.  Everything else of that path (linears, plain / strided / transposed convolutions,
// LayerNorm, relative-position attention) runs on the shared kernels behind cv_gemm_conv / cv_norm_rows / cv_attention.
// All of it is HBM-bound element work over channel-last [time][channel] activations: coalesced float4 rows, fixed-order reductions.
#include "api_common.h"
#include "common.h"
#include "flow_kernels.h"
#include "hift_kernels.h"
#include "group_norm.h"

namespace cv {

// ---- SineGen (type 1) + SourceModuleHnNSF of the 22.05 kHz HiFTGenerator (hifigan/generator.py:125-186, 318-375) ------------------------
// The reference integrates f0 * (h + 1) / sr over the SAMPLES (torch.cumsum, fp32) and takes the result mod 1.  f0 is constant over a frame
// (nearest up-sampling by `scale`), so the running sum at sample k of frame f is base[f][h] + (k + 1) * r[f][h] with r the fp32 per-sample
// increment: one thread per harmonic walks the frames in double (mod 1 per frame: the sum itself reaches thousands), the samples are then independent.
static __global__ __launch_bounds__(64) void sinegen1_walk_kernel(const float* f0, double* base, float* rinc, int m, int H1, float sr, int scale) {
    const int h = threadIdx.x;
    if (h >= H1) return;
    double c = 0.0;
    for (int f = 0; f < m; ++f) {
        const float r = (f0[f] * (float)(h + 1)) / sr;                     // the fp32 value the reference sums
        base[(long long)f * H1 + h] = c;
        rinc[(long long)f * H1 + h] = r;
        c += (double)r * (double)scale;
        c -= floor(c);
    }
}

// per sample: sine_h = amp * sin(2 pi frac(cumsum) + phase0[h]); uv = f0 > thr; wave_h = sine_h * uv + (uv * sigma + (1 - uv) * amp / 3) * noise[h][t];
// s[t] = tanh(sum_h w[h] * wave_h + b).  noise: dev [H1][L] N(0, 1) (the reference's torch.randn_like draw, parity hook) or null -> counter RNG keyed by seed.
static __global__ __launch_bounds__(256) void sinegen1_source_kernel(const float* f0, const double* base, const float* rinc, const float* phase0, const float* noise,
                                                                      unsigned long long seed, const float* lw, const float* lb, float* s, int m, int H1, int scale,
                                                                      float amp, float sigma, float thr) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long L = (long long)m * scale;
    if (t >= L) return;
    const int f = (int)(t / scale), k = (int)(t % scale);
    const float fv = f0[f];
    const float uv = fv > thr ? 1.f : 0.f;
    const float namp = uv * sigma + (1.f - uv) * amp / 3.f;
    float acc = 0.f;
    for (int h = 0; h < H1; ++h) {
        double c = base[(long long)f * H1 + h] + (double)(k + 1) * (double)rinc[(long long)f * H1 + h];
        c -= floor(c);
        const float theta = 2.f * CV_PI_F * (float)c;
        const float nz = noise ? noise[(long long)h * L + t] : gauss01(seed, (unsigned long long)(h * L + t));
        const float w = amp * sinf(theta + phase0[h]) * uv + namp * nz;
        acc += w * lw[h];
    }
    s[t] = tanhf(acc + lb[0]);
}

// F.interpolate(x, size=Tn, mode='linear') over TIME for channel-last rows: y[t][c] (row pitch ldy), x [T][C]   (length_regulator.py:52-70)
static __global__ __launch_bounds__(256) void interp_rows_kernel(const float* x, float* y, int C, int T, int Tn, int ldy) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)C * Tn) return;
    const int t = (int)(i / C), c = (int)(i % C);
    const float scale = (float)T / (float)Tn;
    float src = scale * ((float)t + 0.5f) - 0.5f;
    if (src < 0.f) src = 0.f;
    const int i0 = (int)src, i1 = i0 + (i0 < T - 1 ? 1 : 0);
    const float l1 = src - (float)i0, l0 = 1.f - l1;
    y[(long long)t * ldy + c] = l0 * x[(long long)i0 * C + c] + l1 * x[(long long)i1 * C + c];
}

static inline unsigned nblk256(long long n) { return (unsigned)((n + 255) / 256); }

}  // namespace cv

extern "C" {

int cv_group_norm(const float* x, float* y, int32_t B, int32_t T, int32_t C, int32_t G, const float* gamma, const float* beta, float eps, int32_t act,
                  const float* col_add, int64_t col_add_batch, double* workspace, void* stream) {
    return cv::guarded([&] {
        CV_CHECK(B > 0 && T > 0 && C > 0 && G > 0 && C % G == 0, "cv_group_norm: bad shape");
        CV_CHECK(workspace != nullptr, "cv_group_norm: workspace of B * G * 64 doubles");
        cv::group_norm(x, y, B, T, C, G, gamma, beta, eps, act, col_add, (long long)col_add_batch, workspace, cv::as_stream(stream));
    });
}

int cv_sinegen1_source(const float* f0, int32_t frames, int32_t scale, int32_t harmonics, float sr, const float* phase0, const float* noise, uint64_t seed,
                       const float* lin_w, const float* lin_b, float amp, float sigma, float thr, float* source_out, void* workspace, void* stream) {
    return cv::guarded([&] {
        CV_CHECK(frames > 0 && scale > 0 && harmonics >= 0 && harmonics < 64, "cv_sinegen1_source: bad shape");
        CV_CHECK(workspace != nullptr, "cv_sinegen1_source: workspace of frames * (harmonics + 1) * 12 bytes");
        hipStream_t s = cv::as_stream(stream);
        const int H1 = harmonics + 1;
        double* base = reinterpret_cast<double*>(workspace);
        float* rinc = reinterpret_cast<float*>(base + (size_t)frames * H1);
        hipLaunchKernelGGL(cv::sinegen1_walk_kernel, dim3(1), dim3(64), 0, s, f0, base, rinc, frames, H1, sr, scale);
        hipLaunchKernelGGL(cv::sinegen1_source_kernel, dim3(cv::nblk256((long long)frames * scale)), dim3(256), 0, s, f0, base, rinc, phase0, noise,
                           (unsigned long long)seed, lin_w, lin_b, source_out, frames, H1, scale, amp, sigma, thr);
    });
}

int cv_cfg_euler(float* x, const float* d, int64_t n, float dt, float rate, void* stream) {
    return cv::guarded([&] {
        if (n <= 0) return;
        hipLaunchKernelGGL(cv::cfg_euler_kernel, dim3(cv::nblk256(n)), dim3(256), 0, cv::as_stream(stream), x, d, (long long)n, dt, rate);
    });
}

int cv_pack_cfg_input(const float* x, const float* mu, const float* spks, const float* cond, float* h, int32_t T, int32_t mel, void* stream) {
    return cv::guarded([&] {
        CV_CHECK(T > 0 && mel > 0, "cv_pack_cfg_input: bad shape");
        hipLaunchKernelGGL(cv::pack_est_input_kernel, dim3(cv::nblk256(2LL * T * 4 * mel)), dim3(256), 0, cv::as_stream(stream), x, mu, spks, cond, h, T, mel, 1, 1);
    });
}

int cv_time_sinusoid(const float* t, float* out, int32_t n, int32_t dim, void* stream) {
    return cv::guarded([&] {
        CV_CHECK(n > 0 && dim >= 4 && dim % 2 == 0, "cv_time_sinusoid: bad shape");
        hipLaunchKernelGGL(cv::time_sinusoid_kernel, dim3(n), dim3(256), 0, cv::as_stream(stream), t, out, n, dim);
    });
}

int cv_concat_cols(const float* a, int32_t ca, int64_t a_batch, const float* b, int32_t cb, int64_t b_batch, float* out, int32_t T, int32_t B, void* stream) {
    return cv::guarded([&] {
        CV_CHECK(T > 0 && B > 0 && ca > 0 && cb > 0, "cv_concat_cols: bad shape");
        hipLaunchKernelGGL(cv::concat_cols_batched_kernel, dim3(cv::nblk256((long long)B * T * (ca + cb))), dim3(256), 0, cv::as_stream(stream), a, ca, (long long)a_batch,
                           b, cb, (long long)b_batch, out, T, B);
    });
}

int cv_interp_rows(const float* x, float* y, int32_t C, int32_t T, int32_t Tn, int32_t ldy, void* stream) {
    return cv::guarded([&] {
        CV_CHECK(C > 0 && T > 0 && Tn > 0 && ldy >= C, "cv_interp_rows: bad shape");
        hipLaunchKernelGGL(cv::interp_rows_kernel, dim3(cv::nblk256((long long)C * Tn)), dim3(256), 0, cv::as_stream(stream), x, y, C, T, Tn, ldy);
    });
}

int cv_transpose(const float* in, float* out, int32_t rows, int32_t cols, void* stream) {
    return cv::guarded([&] {
        CV_CHECK(rows > 0 && cols > 0, "cv_transpose: bad shape");
        // to_channel_last_kernel(in [C][T] -> out [T][C]) is a plain 2-D transpose of a [rows = C][cols = T] matrix
        hipLaunchKernelGGL(cv::to_channel_last_kernel, dim3(cv::nblk256((long long)rows * cols)), dim3(256), 0, cv::as_stream(stream), in, out, rows, cols);
    });
}

}  // extern "C"
