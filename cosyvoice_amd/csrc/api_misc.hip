// Glue operators of CosyVoice2Model.token2wav (cosyvoice/cli/model.py:292-326): Hamming cross-fade and speed change.
// The reference round-trips fade_in_out through the CPU (utils/common.py:170-178); here it stays on the device.
#include "api_common.h"
#include "common.h"

namespace cv {

// fade_in[..., :n] = fade_in[..., :n] * window[:n] + fade_out[..., -n:] * window[n:]   (n = len(window)/2), in place on fade_in
static __global__ __launch_bounds__(256) void fade_in_out_kernel(float* fade_in, const float* fade_out_tail, const float* window, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    fade_in[i] = fade_in[i] * window[i] + fade_out_tail[i] * window[n + i];
}

// F.interpolate(x[1,C,T], size=Tn, mode='linear') (align_corners=False), channel-first rows   (cli/model.py:322)
static __global__ __launch_bounds__(256) void interp_linear_kernel(const float* x, float* y, int C, int T, int Tn) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)C * Tn) return;
    const int c = (int)(i / Tn), t = (int)(i % Tn);
    const float scale = (float)T / (float)Tn;
    float src = scale * ((float)t + 0.5f) - 0.5f;
    if (src < 0.f) src = 0.f;
    const int i0 = (int)src, i1 = i0 + (i0 < T - 1 ? 1 : 0);
    const float l1 = src - (float)i0, l0 = 1.f - l1;
    y[i] = l0 * x[(long long)c * T + i0] + l1 * x[(long long)c * T + i1];
}

// ---- prompt-mel front end (cli/frontend.py:120-125 -> matcha.utils.audio.mel_spectrogram, center=False) -----------------------------
// y[0 .. L + 2 pad) = reflect padding of x[0 .. L) by `pad` samples on both sides (torch.nn.functional.pad(mode="reflect"))
static __global__ __launch_bounds__(256) void reflect_pad_kernel(const float* x, float* y, int L, int pad) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= L + 2 * pad) return;
    int j = i - pad;
    if (j < 0) j = -j;
    if (j >= L) j = 2 * (L - 1) - j;
    y[i] = x[j];
}
// spec [T][2 * bins] = (re | im) of the windowed DFT  ->  mag [T][ldm] = sqrt(re^2 + im^2 + eps), columns >= bins zeroed
static __global__ __launch_bounds__(256) void stft_magnitude_kernel(const float* spec, float* mag, int T, int bins, int ldm, float eps) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)T * ldm) return;
    const int t = (int)(i / ldm), f = (int)(i % ldm);
    float v = 0.f;
    if (f < bins) { const float re = spec[(long long)t * 2 * bins + f], im = spec[(long long)t * 2 * bins + bins + f]; v = sqrtf(re * re + im * im + eps); }
    mag[i] = v;
}


// ---- speech-tokenizer / speaker-embedding feature front ends (cli/frontend.py:95-118) ---------------------------------------------------
// spec [T][2 * bins] (re | im)  ->  pw [T][ldm] = re^2 + im^2, columns >= bins zeroed   (whisper: stft.abs() ** 2; kaldi: use_power)
static __global__ __launch_bounds__(256) void stft_power_kernel(const float* spec, float* pw, int T, int bins, int ldm) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)T * ldm) return;
    const int t = (int)(i / ldm), f = (int)(i % ldm);
    float v = 0.f;
    if (f < bins) { const float re = spec[(long long)t * 2 * bins + f], im = spec[(long long)t * 2 * bins + bins + f]; v = re * re + im * im; }
    pw[i] = v;
}

// whisper.audio.log_mel_spectrogram tail: lnmel [T][C] = ln(max(mel, 1e-10))  ->  out [C][T] = (max(log10, max(log10) - 8) + 4) / 4.
// The maximum runs over the WHOLE utterance, so this is one workgroup (<= 30 s x 128 mels = 384 k values, a few tens of microseconds).
static __global__ __launch_bounds__(1024) void whisper_lognorm_kernel(const float* lnmel, float* out, int T, int C) {
    __shared__ float red[16];
    const long long n = (long long)T * C;
    const float inv_ln10 = 0.43429448190325176f;
    float mx = -INFINITY;
    for (long long i = threadIdx.x; i < n; i += 1024) mx = fmaxf(mx, lnmel[i]);
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = red[0];
    for (int w = 1; w < 16; ++w) mx = fmaxf(mx, red[w]);
    const float floor10 = mx * inv_ln10 - 8.0f;
    for (long long i = threadIdx.x; i < n; i += 1024) {
        const int c = (int)(i / T), t = (int)(i % T);                          // consecutive threads write consecutive t of one mel row
        out[i] = (fmaxf(lnmel[(long long)t * C + c] * inv_ln10, floor10) + 4.0f) * 0.25f;
    }
}

// x [T][C] -= mean over T of every column (`feat - feat.mean(dim=0, keepdim=True)`, cli/frontend.py:113).  One workgroup per 32 columns:
// thread (ty, tx) walks rows ty, ty + 8, ... of column tx, so a wavefront reads two 128-byte row segments per step.
static __global__ __launch_bounds__(256) void sub_col_mean_kernel(float* x, int T, int C) {
    __shared__ float part[8][32];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5, c = blockIdx.x * 32 + tx;
    float s = 0.f;
    if (c < C) for (int t = ty; t < T; t += 8) s += x[(long long)t * C + c];
    part[ty][tx] = s;
    __syncthreads();
    float m = 0.f;
    for (int k = 0; k < 8; ++k) m += part[k][tx];
    m /= (float)T;
    if (c < C) for (int t = ty; t < T; t += 8) x[(long long)t * C + c] -= m;
}

}  // namespace cv

extern "C" {

int cv_reflect_pad(const float* x, float* y, int32_t L, int32_t pad, void* stream) {
    return cv::guarded([&] {
        CV_CHECK(x && y && L > pad && pad >= 0, "cv_reflect_pad: needs L > pad (reflection without the edge sample)");
        hipLaunchKernelGGL(cv::reflect_pad_kernel, dim3((unsigned)((L + 2 * pad + 255) / 256)), dim3(256), 0, cv::as_stream(stream), x, y, L, pad);
    });
}

int cv_stft_magnitude(const float* spec, float* mag, int32_t T, int32_t bins, int32_t ldm, float eps, void* stream) {
    return cv::guarded([&] {
        CV_CHECK(spec && mag && T > 0 && bins > 0 && ldm >= bins, "cv_stft_magnitude: bad arguments");
        hipLaunchKernelGGL(cv::stft_magnitude_kernel, dim3((unsigned)(((long long)T * ldm + 255) / 256)), dim3(256), 0, cv::as_stream(stream), spec, mag, T, bins, ldm, eps);
    });
}

int cv_stft_power(const float* spec, float* pw, int32_t T, int32_t bins, int32_t ldm, void* stream) {
    return cv::guarded([&] {
        CV_CHECK(spec && pw && T > 0 && bins > 0 && ldm >= bins, "cv_stft_power: bad arguments");
        hipLaunchKernelGGL(cv::stft_power_kernel, dim3((unsigned)(((long long)T * ldm + 255) / 256)), dim3(256), 0, cv::as_stream(stream), spec, pw, T, bins, ldm);
    });
}

int cv_whisper_lognorm(const float* lnmel, float* out, int32_t T, int32_t n_mels, void* stream) {
    return cv::guarded([&] {
        CV_CHECK(lnmel && out && lnmel != out && T > 0 && n_mels > 0, "cv_whisper_lognorm: bad arguments (out of place: the output is transposed)");
        hipLaunchKernelGGL(cv::whisper_lognorm_kernel, dim3(1), dim3(1024), 0, cv::as_stream(stream), lnmel, out, T, n_mels);
    });
}

int cv_sub_col_mean(float* x, int32_t T, int32_t C, void* stream) {
    return cv::guarded([&] {
        CV_CHECK(x && T > 0 && C > 0, "cv_sub_col_mean: bad arguments");
        hipLaunchKernelGGL(cv::sub_col_mean_kernel, dim3((unsigned)((C + 31) / 32)), dim3(256), 0, cv::as_stream(stream), x, T, C);
    });
}

int cv_fade_in_out(float* fade_in, const float* fade_out_tail, const float* window, int32_t overlap, void* stream) {
    return cv::guarded([&] {
        CV_CHECK(fade_in && fade_out_tail && window && overlap > 0, "cv_fade_in_out: bad arguments");
        hipLaunchKernelGGL(cv::fade_in_out_kernel, dim3((overlap + 255) / 256), dim3(256), 0, cv::as_stream(stream), fade_in, fade_out_tail, window, overlap);
    });
}

int cv_interp_linear(const float* x, float* y, int32_t C, int32_t T, int32_t Tn, void* stream) {
    return cv::guarded([&] {
        CV_CHECK(x && y && C > 0 && T > 0 && Tn > 0, "cv_interp_linear: bad arguments");
        hipLaunchKernelGGL(cv::interp_linear_kernel, dim3((unsigned)(((long long)C * Tn + 255) / 256)), dim3(256), 0, cv::as_stream(stream), x, y, C, T, Tn);
    });
}

}  // extern "C"
