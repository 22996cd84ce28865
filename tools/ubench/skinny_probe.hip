// Dev tool (round 3): the batched-decode skinny GEMM (llm_batch_kernels.h) on the Qwen2-0.5B layer shapes at nb = 8 / 16 sequences, as dependent chains of
// 24 launches with 24 different weight sets inside a hipGraph (cold weights as in a decode step), next to the batch-1 GEMV of the same matrix, plus clock64()
// phase stamps of thread 0 of every workgroup: entry -> first k-tile multiplied -> all tiles multiplied -> barrier -> end.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I cosyvoice_amd/csrc -I include tools/ubench/skinny_probe.hip -o tools/ubench/skinny_probe
#include "../../cosyvoice_amd/csrc/llm_batch_kernels.h"
#include <vector>
#include <cstdio>
#include <cstring>
#include <functional>
#include <algorithm>
using namespace cv;

static float time_graph(int n_units, const std::function<void(hipStream_t)>& enqueue, int reps = 10) {
    hipStream_t s; (void)hipStreamCreate(&s);
    hipGraph_t g; hipGraphExec_t ge;
    (void)hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    enqueue(s);
    (void)hipStreamEndCapture(s, &g);
    (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int i = 0; i < 2; ++i) (void)hipGraphLaunch(ge, s);
    (void)hipStreamSynchronize(s);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int trial = 0; trial < 3; ++trial) {
        (void)hipEventRecord(e0, s);
        for (int i = 0; i < reps; ++i) (void)hipGraphLaunch(ge, s);
        (void)hipEventRecord(e1, s); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        best = std::min(best, ms * 1e3f / (reps * n_units));
    }
    (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g); (void)hipStreamDestroy(s);
    return best;
}
static void* dmalloc(size_t b) { void* p; if (hipMalloc(&p, b) != hipSuccess) { printf("hipMalloc failed\n"); exit(1); } (void)hipMemset(p, 0, b); return p; }
static unsigned long long rs = 88172645463325252ull;
static inline unsigned long long rnd() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return rs; }
static inline float rndf() { return (float)((rnd() >> 40) & 0xffff) / 32768.f - 1.f; }
static inline unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (unsigned short)(u >> 16); }

__global__ void empty_kernel(int) {}

template <typename K>
static void stamps(const char* what, K kern, dim3 grid, SkinnyArgs a, long long* dbg) {
    const int nwg = (int)grid.x;
    (void)hipMemset(dbg, 0, (size_t)nwg * 64);
    SkinnyArgs b = a; b.dbg = nullptr;
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, grid, dim3(256), 0, nullptr, b);
    a.dbg = dbg;
    hipLaunchKernelGGL(kern, grid, dim3(256), 0, nullptr, a);
    (void)hipDeviceSynchronize();
    std::vector<long long> h((size_t)nwg * 8);
    (void)hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost);
    double ph[4] = {0, 0, 0, 0}; double tot = 0, mx = 0;
    for (int w = 0; w < nwg; ++w) {
        for (int q = 0; q < 4; ++q) ph[q] += (double)(h[(size_t)w * 8 + q + 1] - h[(size_t)w * 8 + q]) / nwg;
        const double t = (double)(h[(size_t)w * 8 + 4] - h[(size_t)w * 8]); tot += t / nwg; mx = std::max(mx, t);
    }
    (void)tot; (void)mx;
    printf("    stamps %-30s %4d WG: entry -> first k-tile multiplied %6.0f | remaining tiles %6.0f | barrier %5.0f | combine + store %5.0f | total %6.0f clocks\n",
           what, nwg, ph[0], ph[1], ph[2], ph[3], ph[0] + ph[1] + ph[2] + ph[3]);
}

int main() {
    constexpr int H = 896, I = 4864, Q = 1152, NL = 24;
    std::vector<float> hx((size_t)16 * I);
    for (auto& v : hx) v = rndf();
    float* x = (float*)dmalloc(hx.size() * 4); (void)hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    std::vector<float> hg(H); for (auto& v : hg) v = 1.f + 0.1f * rndf();
    float* gam = (float*)dmalloc(H * 4); (void)hipMemcpy(gam, hg.data(), H * 4, hipMemcpyHostToDevice);
    auto make_w = [&](size_t numel) {
        std::vector<unsigned short> h(numel);
        for (auto& v : h) v = f2bf(rndf() * 0.05f);
        void* d = dmalloc(numel * 2); (void)hipMemcpy(d, h.data(), numel * 2, hipMemcpyHostToDevice); return (bf16_t*)d;
    };
    std::vector<bf16_t*> wgu(NL), wqkv(NL), wo(NL), wdown(NL);
    for (int l = 0; l < NL; ++l) { wgu[l] = make_w((size_t)2 * I * H); wqkv[l] = make_w((size_t)Q * H); wo[l] = make_w((size_t)H * H); wdown[l] = make_w((size_t)H * I); }
    auto pack = [&](const bf16_t* W, int N, int K) {
        const long long pieces = (long long)((N + 15) / 16) * (K / 32) * 64;
        bf16_t* d = (bf16_t*)dmalloc((size_t)pieces * 16);
        hipLaunchKernelGGL(pack_frag_kernel, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, nullptr, W, d, N, K);
        return d;
    };
    std::vector<bf16_t*> pgu(NL), pqkv(NL), po(NL), pdown(NL);
    for (int l = 0; l < NL; ++l) { pgu[l] = pack(wgu[l], 2 * I, H); pqkv[l] = pack(wqkv[l], Q, H); po[l] = pack(wo[l], H, H); pdown[l] = pack(wdown[l], H, I); }
    (void)hipDeviceSynchronize();
    float* y = (float*)dmalloc((size_t)16 * 2 * I * 4 * 8);
    float* y2 = (float*)dmalloc((size_t)16 * 2 * I * 4 * 8);
    auto same = [&](size_t n) {
        std::vector<float> a(n), b(n);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(a.data(), y, n * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(b.data(), y2, n * 4, hipMemcpyDeviceToHost);
        size_t bad = 0; for (size_t i = 0; i < n; ++i) bad += memcmp(&a[i], &b[i], 4) != 0;
        return bad;
    };
    float* res = (float*)dmalloc((size_t)16 * H * 4);
    float* dpart = (float*)dmalloc((size_t)8 * 16 * H * 4);
    long long* dbg = (long long*)dmalloc((size_t)4096 * 64);

    printf("empty <<<256,256>>> chain                                  %6.2f us per launch\n", time_graph(NL, [&](hipStream_t s) { for (int l = 0; l < NL; ++l) hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, s, l); }));
    // batch-1 reference points (llm_kernels.h)
    {
        float* xs = (float*)dmalloc(H * 4); float* ys = (float*)dmalloc((size_t)2 * I * 4);
        const float us = time_graph(NL, [&](hipStream_t s) {
            for (int l = 0; l < NL; ++l) { GemvArgs a{wgu[l], nullptr, x, ys, 2 * I, H, gam, 1e-6f, nullptr, 1, nullptr}; hipLaunchKernelGGL((gemv_norm_kernel<7, 2, 5>), dim3((I + 19) / 20), dim3(320), 0, s, a); } });
        printf("batch 1: gemv_norm_kernel<7,2,5> gate/up (17.4 MB)          %6.2f us per launch\n", us);
        (void)xs;
    }
    for (int nb : {8, 16}) {
        printf("==== %d sequences\n", nb);
        auto gu = [&](int l, const float* g) { return SkinnyArgs{wgu[l], nullptr, x, H, y, I, 2 * I, H, g, 1e-6f, nullptr, 0, 1, nb, 1, nullptr}; };
        auto qkv = [&](int l, const float* g) { return SkinnyArgs{wqkv[l], nullptr, x, H, y, Q, Q, H, g, 1e-6f, nullptr, 0, 0, nb, 1, nullptr}; };
        auto op = [&](int l) { return SkinnyArgs{wo[l], nullptr, x, H, y, H, H, H, nullptr, 0.f, res, H, 0, nb, 1, nullptr}; };
        auto down = [&](int l) { return SkinnyArgs{wdown[l], nullptr, x, I, dpart, H, H, I, nullptr, 0.f, nullptr, 0, 2, nb, 8, nullptr}; };
        const dim3 g_gu((2 * I / 16 + 1) / 2), g_qkv(Q / 16), g_o(H / 16), g_down(((H / 16 + 1) / 2) * 8);
        float us;
        us = time_graph(NL, [&](hipStream_t s) { for (int l = 0; l < NL; ++l) hipLaunchKernelGGL((skinny_mfma_kernel<2, 7, true>), g_gu, dim3(256), 0, s, gu(l, gam)); });
        printf("gate/up  skinny<2,7> + RMSNorm (17.4 MB, %3u WG)              %6.2f us per launch\n", g_gu.x, us);
        stamps("gate/up + norm", skinny_mfma_kernel<2, 7, true>, g_gu, gu(3, gam), dbg);
        us = time_graph(NL, [&](hipStream_t s) { for (int l = 0; l < NL; ++l) hipLaunchKernelGGL((skinny_mfma_kernel<2, 7, true>), g_gu, dim3(256), 0, s, gu(l, nullptr)); });
        printf("gate/up  skinny<2,7> WITHOUT the norm (no gamma loads)       %6.2f us per launch\n", us);
        stamps("gate/up no norm", skinny_mfma_kernel<2, 7, true>, g_gu, gu(3, nullptr), dbg);
        us = time_graph(NL, [&](hipStream_t s) { for (int l = 0; l < NL; ++l) hipLaunchKernelGGL((skinny_mfma_kernel<2, 7, true>), g_gu, dim3(256), 0, s, gu(0, gam)); });
        printf("gate/up  skinny<2,7> + RMSNorm, ONE weight set (L2 / MALL warm) %6.2f us per launch\n", us);
        us = time_graph(NL, [&](hipStream_t s) { for (int l = 0; l < NL; ++l) hipLaunchKernelGGL((skinny_mfma_kernel<1, 7, true>), dim3(2 * I / 16), dim3(256), 0, s, gu(l, gam)); });
        printf("gate/up  skinny<1,7> + RMSNorm (one row tile per WG, 608 WG) %6.2f us per launch\n", us);
        us = time_graph(NL, [&](hipStream_t s) { for (int l = 0; l < NL; ++l) hipLaunchKernelGGL((skinny_mfma_kernel<4, 7, true>), dim3((2 * I / 16 + 3) / 4), dim3(256), 0, s, gu(l, gam)); });
        printf("gate/up  skinny<4,7> + RMSNorm (four row tiles per WG, 152 WG) %6.2f us per launch\n", us);
        us = time_graph(NL, [&](hipStream_t s) { for (int l = 0; l < NL; ++l) hipLaunchKernelGGL((skinny_mfma_kernel<1, 7, true>), g_qkv, dim3(256), 0, s, qkv(l, gam)); });
        printf("qkv      skinny<1,7> + RMSNorm (2.1 MB, %3u WG)              %6.2f us per launch\n", g_qkv.x, us);
        stamps("qkv + norm", skinny_mfma_kernel<1, 7, true>, g_qkv, qkv(3, gam), dbg);
        us = time_graph(NL, [&](hipStream_t s) { for (int l = 0; l < NL; ++l) hipLaunchKernelGGL((skinny_mfma_kernel<1, 7, true>), g_o, dim3(256), 0, s, op(l)); });
        printf("o_proj   skinny<1,7> + residual (1.6 MB, %3u WG)             %6.2f us per launch\n", g_o.x, us);
        stamps("o_proj", skinny_mfma_kernel<1, 7, true>, g_o, op(3), dbg);
        us = time_graph(NL, [&](hipStream_t s) { for (int l = 0; l < NL; ++l) hipLaunchKernelGGL((skinny_mfma_kernel<2, 5, true>), g_down, dim3(256), 0, s, down(l)); });
        printf("down     skinny<2,5> 8 K ranges (8.7 MB, %3u WG)             %6.2f us per launch\n", g_down.x, us);
        stamps("down split-K", skinny_mfma_kernel<2, 5, true>, g_down, down(3), dbg);
        // ---- round 3: fragment-ordered weights + wave-private activation staging (skinny_pk_kernel)
        auto with = [&](SkinnyArgs a, const bf16_t* wp, float* out) { a.W = wp; a.y = out; return a; };
        (void)hipMemset(y, 0, (size_t)16 * 2 * I * 4); (void)hipMemset(y2, 0, (size_t)16 * 2 * I * 4);
        hipLaunchKernelGGL((skinny_mfma_kernel<2, 7, true>), g_gu, dim3(256), 0, nullptr, gu(3, gam)); hipLaunchKernelGGL((skinny_pk_kernel<2, 7>), g_gu, dim3(256), 0, nullptr, with(gu(3, gam), pgu[3], y2));
        const size_t bad_gu = same((size_t)nb * I);
        us = time_graph(NL, [&](hipStream_t s) { for (int l = 0; l < NL; ++l) hipLaunchKernelGGL((skinny_pk_kernel<2, 7>), g_gu, dim3(256), 0, s, with(gu(l, gam), pgu[l], y2)); });
        printf("gate/up  skinny_pk<2,7> + RMSNorm                            %6.2f us per launch   (%zu values differ from skinny_mfma)\n", us, bad_gu);
        stamps("pk gate/up + norm", skinny_pk_kernel<2, 7>, g_gu, with(gu(3, gam), pgu[3], y2), dbg);
        us = time_graph(NL, [&](hipStream_t s) { for (int l = 0; l < NL; ++l) hipLaunchKernelGGL((skinny_pk_kernel<4, 7>), dim3((2 * I / 16 + 3) / 4), dim3(256), 0, s, with(gu(l, gam), pgu[l], y2)); });
        printf("gate/up  skinny_pk<4,7> + RMSNorm (152 WG)                   %6.2f us per launch\n", us);
        us = time_graph(NL, [&](hipStream_t s) { for (int l = 0; l < NL; ++l) hipLaunchKernelGGL((skinny_pk_kernel<1, 7>), dim3(2 * I / 16), dim3(256), 0, s, with(gu(l, gam), pgu[l], y2)); });
        printf("gate/up  skinny_pk<1,7> + RMSNorm (608 WG)                   %6.2f us per launch\n", us);
        (void)hipMemset(y, 0, (size_t)16 * Q * 4); (void)hipMemset(y2, 0, (size_t)16 * Q * 4);
        hipLaunchKernelGGL((skinny_mfma_kernel<1, 7, true>), g_qkv, dim3(256), 0, nullptr, qkv(3, gam)); hipLaunchKernelGGL((skinny_pk_kernel<1, 7>), g_qkv, dim3(256), 0, nullptr, with(qkv(3, gam), pqkv[3], y2));
        const size_t bad_qkv = same((size_t)nb * Q);
        us = time_graph(NL, [&](hipStream_t s) { for (int l = 0; l < NL; ++l) hipLaunchKernelGGL((skinny_pk_kernel<1, 7>), g_qkv, dim3(256), 0, s, with(qkv(l, gam), pqkv[l], y2)); });
        printf("qkv      skinny_pk<1,7> + RMSNorm                            %6.2f us per launch   (%zu values differ)\n", us, bad_qkv);
        stamps("pk qkv + norm", skinny_pk_kernel<1, 7>, g_qkv, with(qkv(3, gam), pqkv[3], y2), dbg);
        (void)hipMemset(y, 0, (size_t)16 * H * 4); (void)hipMemset(y2, 0, (size_t)16 * H * 4);
        hipLaunchKernelGGL((skinny_mfma_kernel<1, 7, true>), g_o, dim3(256), 0, nullptr, op(3)); hipLaunchKernelGGL((skinny_pk_kernel<1, 7>), g_o, dim3(256), 0, nullptr, with(op(3), po[3], y2));
        const size_t bad_o = same((size_t)nb * H);
        us = time_graph(NL, [&](hipStream_t s) { for (int l = 0; l < NL; ++l) hipLaunchKernelGGL((skinny_pk_kernel<1, 7>), g_o, dim3(256), 0, s, with(op(l), po[l], y2)); });
        printf("o_proj   skinny_pk<1,7> + residual                           %6.2f us per launch   (%zu values differ)\n", us, bad_o);
        {
            auto dn = [&](int l, float* out) { SkinnyArgs a = down(l); a.W = pdown[l]; a.y = out; return a; };
            float* dp2 = y2;
            (void)hipMemset(dpart, 0, (size_t)8 * 16 * H * 4); (void)hipMemset(dp2, 0, (size_t)8 * 16 * H * 4);
            hipLaunchKernelGGL((skinny_mfma_kernel<2, 5, true>), g_down, dim3(256), 0, nullptr, down(3)); hipLaunchKernelGGL((skinny_pk_kernel<2, 5>), g_down, dim3(256), 0, nullptr, dn(3, dp2));
            (void)hipDeviceSynchronize();
            std::vector<float> a((size_t)8 * nb * H), b(a.size());
            (void)hipMemcpy(a.data(), dpart, a.size() * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(b.data(), dp2, b.size() * 4, hipMemcpyDeviceToHost);
            size_t bad = 0; for (size_t i = 0; i < a.size(); ++i) bad += memcmp(&a[i], &b[i], 4) != 0;
            us = time_graph(NL, [&](hipStream_t s) { for (int l = 0; l < NL; ++l) hipLaunchKernelGGL((skinny_pk_kernel<2, 5>), g_down, dim3(256), 0, s, dn(l, dp2)); });
            printf("down     skinny_pk<2,5> 8 K ranges                           %6.2f us per launch   (%zu values differ)\n", us, bad);
            stamps("pk down split-K", skinny_pk_kernel<2, 5>, g_down, dn(3, dp2), dbg);
            // 4 K ranges x 1 row tile: 56 x 4 = 224 workgroups as well, half the partial sums
            auto dn4 = [&](int l) { SkinnyArgs a = down(l); a.W = pdown[l]; a.y = dp2; a.ksplit = 4; return a; };
            us = time_graph(NL, [&](hipStream_t s) { for (int l = 0; l < NL; ++l) hipLaunchKernelGGL((skinny_pk_kernel<1, 8, 5>), dim3(56 * 4), dim3(320), 0, s, dn4(l)); });
            printf("down     skinny_pk<1,8,5 waves> 4 K ranges x 1 row tile      %6.2f us per launch\n", us);
        }
        us = time_graph(NL, [&](hipStream_t s) { for (int l = 0; l < NL; ++l) hipLaunchKernelGGL(sum_partials_kernel, dim3((unsigned)((nb * H / 4 + 255) / 256)), dim3(256), 0, s, dpart, 8, nb, H, res, (long long)H, res, (long long)H); });
        printf("sum_partials_kernel                                          %6.2f us per launch\n", us);
        fflush(stdout);
    }
    return 0;
}
