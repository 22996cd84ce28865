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

}  // namespace cv

extern "C" {

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
