// Dev tool: what does ONE kernel of the batch-1 decode chain cost?  Chains of 120 dependent launches captured in a hipGraph,
// replayed 20x: an empty kernel, each decode GEMV shape (cold = a different weight matrix per launch as in 24 layers, warm = the
// same matrix every launch) and the decode attention at 381 keys.  Prints microseconds per kernel.
#include "../../cosyvoice_amd/csrc/llm_kernels.h"
#include <vector>
#include <cstdio>
#include <functional>
using namespace cv;

__global__ void empty_kernel(float* p) { if (p && threadIdx.x == 9999) p[0] = 1.f; }

static float time_chain(const char* name, int n, const std::function<void(int, hipStream_t)>& launch) {
    hipStream_t s; (void)hipStreamCreate(&s);
    hipGraph_t g; hipGraphExec_t ge;
    (void)hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < n; ++i) launch(i, s);
    (void)hipStreamEndCapture(s, &g);
    (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int i = 0; i < 3; ++i) (void)hipGraphLaunch(ge, s);
    (void)hipStreamSynchronize(s);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int reps = 20;
    (void)hipEventRecord(e0, s);
    for (int i = 0; i < reps; ++i) (void)hipGraphLaunch(ge, s);
    (void)hipEventRecord(e1, s); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const float us = ms * 1e3f / (reps * n);
    printf("%-46s %7.2f us/kernel\n", name, us); fflush(stdout);
    // eager for comparison
    (void)hipEventRecord(e0, s);
    for (int r = 0; r < 5; ++r) for (int i = 0; i < n; ++i) launch(i, s);
    (void)hipEventRecord(e1, s); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-46s %7.2f us/kernel (eager)\n", "", ms * 1e3f / (5 * n));
    (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g); (void)hipStreamDestroy(s);
    return us;
}

int main() {
    const int NL = 24, H = 896, QKV = 1152, I = 4864;
    auto dmalloc = [](size_t b) { void* p; (void)hipMalloc(&p, b); (void)hipMemset(p, 0, b); return p; };
    float* x = (float*)dmalloc(8192 * 4); float* y = (float*)dmalloc(16384 * 4); float* y2 = (float*)dmalloc(16384 * 4);
    float* gamma = (float*)dmalloc(H * 4);
    DecodeState* st = (DecodeState*)dmalloc(sizeof(DecodeState));
    DecodeState hs{}; hs.pos = 380; (void)hipMemcpy(st, &hs, sizeof(hs), hipMemcpyHostToDevice);
    std::vector<bf16_t*> wqkv(NL), wo(NL), wgu(NL), wd(NL);
    for (int i = 0; i < NL; ++i) {
        wqkv[i] = (bf16_t*)dmalloc((size_t)QKV * H * 2); wo[i] = (bf16_t*)dmalloc((size_t)H * H * 2);
        wgu[i] = (bf16_t*)dmalloc((size_t)2 * I * H * 2); wd[i] = (bf16_t*)dmalloc((size_t)H * I * 2);
    }
    const int max_len = 512;
    std::vector<float*> kc(NL), vc(NL);
    for (int i = 0; i < NL; ++i) { kc[i] = (float*)dmalloc((size_t)2 * max_len * 64 * 4); vc[i] = (float*)dmalloc((size_t)2 * max_len * 64 * 4); }
    float* rc = (float*)dmalloc(max_len * 32 * 4); float* rs = (float*)dmalloc(max_len * 32 * 4);
    float* part = (float*)dmalloc(14 * 16 * ATTN_PART * 4);
    const int N = 120;

    time_chain("empty <<<1,64>>>", N, [&](int, hipStream_t s) { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s, (float*)nullptr); });
    time_chain("empty <<<1024,64>>>", N, [&](int, hipStream_t s) { hipLaunchKernelGGL(empty_kernel, dim3(1024), dim3(64), 0, s, (float*)nullptr); });
    time_chain("empty <<<256,256>>>", N, [&](int, hipStream_t s) { hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, s, (float*)nullptr); });
    for (int cold = 1; cold >= 0; --cold) {
        char nm[96];
        snprintf(nm, 96, "gemv<7,1,1> qkv 1152x896 +rmsnorm  %s", cold ? "cold" : "warm");
        time_chain(nm, N, [&](int i, hipStream_t s) {
            GemvArgs a{wqkv[cold ? i % NL : 0], nullptr, (i & 1) ? y : x, (i & 1) ? x : y, QKV, H, gamma, 1e-6f, nullptr, 0, st};
            hipLaunchKernelGGL((gemv_kernel<7, 1, 1>), dim3(QKV / 4), dim3(64), 0, s, a); });
        snprintf(nm, 96, "gemv<7,1,1> o_proj 896x896 +res       %s", cold ? "cold" : "warm");
        time_chain(nm, N, [&](int i, hipStream_t s) {
            GemvArgs a{wo[cold ? i % NL : 0], nullptr, (i & 1) ? y : x, (i & 1) ? x : y, H, H, nullptr, 0.f, y2, 0, st};
            hipLaunchKernelGGL((gemv_kernel<7, 1, 1>), dim3(H / 4), dim3(64), 0, s, a); });
        snprintf(nm, 96, "gemv<7,2,1> gate_up 9728x896 +rms+silu %s", cold ? "cold" : "warm");
        time_chain(nm, N, [&](int i, hipStream_t s) {
            GemvArgs a{wgu[cold ? i % NL : 0], nullptr, (i & 1) ? y : x, (i & 1) ? x : y, 2 * I, H, gamma, 1e-6f, nullptr, 1, st};
            hipLaunchKernelGGL((gemv_kernel<7, 2, 1>), dim3(I / 4), dim3(64), 0, s, a); });
        snprintf(nm, 96, "gemv<10,1,4> down 896x4864 +res        %s", cold ? "cold" : "warm");
        time_chain(nm, N, [&](int i, hipStream_t s) {
            GemvArgs a{wd[cold ? i % NL : 0], nullptr, (i & 1) ? y : x, (i & 1) ? x : y, H, I, nullptr, 0.f, y2, 0, st};
            hipLaunchKernelGGL((gemv_kernel<10, 1, 4>), dim3(H / 4), dim3(256), 0, s, a); });
        for (int nsp : {4, 8, 16}) {
            snprintf(nm, 96, "attn_decode 381 keys, %d slice(s)          %s", nsp, cold ? "cold" : "warm");
            time_chain(nm, N, [&](int i, hipStream_t s) {
                const int l = cold ? i % NL : 0;
                AttnDecodeArgs a{x, kc[l], vc[l], rc, rs, 14, 2, max_len, st, part, nsp};
                hipLaunchKernelGGL(attn_decode_kernel, dim3(14 * nsp), dim3(64), 0, s, a); });
            snprintf(nm, 96, "gemv<2,1,4,NSP=%d> o_proj over partials     %s", nsp, cold ? "cold" : "warm");
            time_chain(nm, N, [&](int i, hipStream_t s) {
                GemvArgs a{wo[cold ? i % NL : 0], nullptr, nullptr, (i & 1) ? x : y, H, H, nullptr, 0.f, y2, 0, st}; a.part = part;
                if (nsp == 4) hipLaunchKernelGGL((gemv_kernel<2, 1, 4, 4>), dim3(H / 4), dim3(256), 0, s, a);
                else if (nsp == 8) hipLaunchKernelGGL((gemv_kernel<2, 1, 4, 8>), dim3(H / 4), dim3(256), 0, s, a);
                else hipLaunchKernelGGL((gemv_kernel<2, 1, 4, 16>), dim3(H / 4), dim3(256), 0, s, a); });
        }
    }
    // a whole layer the way llm.hip enqueues it
    for (int nsp : {4, 8, 16}) {
        char nm[96]; snprintf(nm, 96, "full layer x24 (5 kernels), %d slice(s): per LAYER", nsp);
        hipStream_t s; (void)s;
        float us = time_chain(nm, NL, [&](int i, hipStream_t s) {
            GemvArgs a0{wqkv[i], nullptr, x, y, QKV, H, gamma, 1e-6f, nullptr, 0, st};
            hipLaunchKernelGGL((gemv_kernel<7, 1, 1>), dim3(QKV / 4), dim3(64), 0, s, a0);
            AttnDecodeArgs ad{y, kc[i], vc[i], rc, rs, 14, 2, max_len, st, part, nsp};
            hipLaunchKernelGGL(attn_decode_kernel, dim3(14 * nsp), dim3(64), 0, s, ad);
            GemvArgs a1{wo[i], nullptr, nullptr, x, H, H, nullptr, 0.f, x, 0, st}; a1.part = part;
            if (nsp == 4) hipLaunchKernelGGL((gemv_kernel<2, 1, 4, 4>), dim3(H / 4), dim3(256), 0, s, a1);
            else if (nsp == 8) hipLaunchKernelGGL((gemv_kernel<2, 1, 4, 8>), dim3(H / 4), dim3(256), 0, s, a1);
            else hipLaunchKernelGGL((gemv_kernel<2, 1, 4, 16>), dim3(H / 4), dim3(256), 0, s, a1);
            GemvArgs a2{wgu[i], nullptr, x, y, 2 * I, H, gamma, 1e-6f, nullptr, 1, st};
            hipLaunchKernelGGL((gemv_kernel<7, 2, 1>), dim3(I / 4), dim3(64), 0, s, a2);
            GemvArgs a3{wd[i], nullptr, y, x, H, I, nullptr, 0.f, x, 0, st};
            hipLaunchKernelGGL((gemv_kernel<10, 1, 4>), dim3(H / 4), dim3(256), 0, s, a3); });
        (void)us;
    }
    return 0;
}
