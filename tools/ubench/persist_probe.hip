// Dev tool (round 3): the gate/up -> SiLU*up -> down(+residual) pair of a batch-1 decode layer as ONE launch with an
// in-launch granule hand-off (cosyvoice_amd/csrc/llm_persist.h) against the two-launch chain the product runs today
// (gemv_norm_kernel<7,2,5> + gemv_kernel<10,1,4>), each as a hipGraph of 24 layers (own weights per layer, as in the decode step), 20 replays
// between one event pair.  Checks the one-launch result against the chain's and against a double-precision host reference, prints the
// per-workgroup phase stamps (entry -> act published -> gather done -> exit) and the give-up flag.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I cosyvoice_amd/csrc -I include tools/ubench/persist_probe.hip -o tools/ubench/persist_probe
#include "../../cosyvoice_amd/csrc/experiments/llm_persist.h"
#include <vector>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <functional>
#include <algorithm>
using namespace cv;

__global__ void bump_kernel(int* e) { if (threadIdx.x == 0) *e += 1; }
__global__ void empty_kernel(float* p) { if (p && threadIdx.x == 9999) p[0] = 1.f; }

static float time_graph(const char* name, int n_units, const std::function<void(hipStream_t)>& enqueue, int reps = 20) {
    hipStream_t s; (void)hipStreamCreate(&s);
    hipGraph_t g; hipGraphExec_t ge;
    (void)hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    enqueue(s);
    (void)hipStreamEndCapture(s, &g);
    (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int i = 0; i < 3; ++i) (void)hipGraphLaunch(ge, s);
    (void)hipStreamSynchronize(s);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f, sum = 0.f;
    for (int trial = 0; trial < 3; ++trial) {
        (void)hipEventRecord(e0, s);
        for (int i = 0; i < reps; ++i) (void)hipGraphLaunch(ge, s);
        (void)hipEventRecord(e1, s); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const float us = ms * 1e3f / (reps * n_units);
        best = std::min(best, us); sum += us;
    }
    printf("%-64s %7.2f us per layer (best of 3; mean %.2f)\n", name, best, sum / 3); fflush(stdout);
    (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g); (void)hipStreamDestroy(s);
    return best;
}

static float bf(bf16_t v) { unsigned u = (unsigned)v << 16; float f; memcpy(&f, &u, 4); return f; }
static bf16_t to_bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (bf16_t)(u >> 16); }

int main() {
    const int NL = 24, H = 896, I = 4864, G = 256;
    auto dmalloc = [](size_t b) { void* p; if (hipMalloc(&p, b) != hipSuccess) { printf("hipMalloc failed\n"); exit(1); } (void)hipMemset(p, 0, b); return p; };
    // seeded weights (the same for every layer index pattern: layer l uses its own buffers, filled with different values)
    std::vector<std::vector<bf16_t>> hgu(NL), hd(NL);
    std::vector<bf16_t*> wgu(NL), wd(NL);
    std::vector<u64_t*> gran(NL);
    unsigned long long z = 88172645463325252ull;
    auto rnd = [&]() { z ^= z << 13; z ^= z >> 7; z ^= z << 17; return (float)((z >> 11) * (1.0 / 9007199254740992.0)) * 2.f - 1.f; };
    for (int l = 0; l < NL; ++l) {
        hgu[l].resize((size_t)2 * I * H); hd[l].resize((size_t)H * I);
        for (auto& v : hgu[l]) v = to_bf(rnd() * 0.06f);
        for (auto& v : hd[l]) v = to_bf(rnd() * 0.025f);
        wgu[l] = (bf16_t*)dmalloc(hgu[l].size() * 2); wd[l] = (bf16_t*)dmalloc(hd[l].size() * 2);
        (void)hipMemcpy(wgu[l], hgu[l].data(), hgu[l].size() * 2, hipMemcpyHostToDevice);
        (void)hipMemcpy(wd[l], hd[l].data(), hd[l].size() * 2, hipMemcpyHostToDevice);
        gran[l] = (u64_t*)dmalloc((size_t)I * 8);
    }
    std::vector<float> hgamma(H), hx0(H);
    for (auto& v : hgamma) v = 1.f + 0.1f * rnd();
    for (auto& v : hx0) v = rnd();
    float* gamma = (float*)dmalloc(H * 4); (void)hipMemcpy(gamma, hgamma.data(), H * 4, hipMemcpyHostToDevice);
    float* h_chain = (float*)dmalloc(H * 4); float* h_pers = (float*)dmalloc(H * 4); float* act = (float*)dmalloc(I * 4);
    int* epoch = (int*)dmalloc(4); int* fail = (int*)dmalloc(4);
    long long* stamps = (long long*)dmalloc((size_t)G * 8 * 8);
    DecodeState* st = (DecodeState*)dmalloc(sizeof(DecodeState));
    const float eps = 1e-6f;

    auto chain_layer = [&](int l, float* h, hipStream_t s) {
        GemvArgs a2{wgu[l], nullptr, h, act, 2 * I, H, gamma, eps, nullptr, 1, st};
        hipLaunchKernelGGL((gemv_norm_kernel<7, 2, 5>), dim3((I + 19) / 20), dim3(320), 0, s, a2);
        GemvArgs a3{wd[l], nullptr, act, h, H, I, nullptr, 0.f, h, 0, st};
        hipLaunchKernelGGL((gemv_kernel<10, 1, 4>), dim3(H / 4), dim3(256), 0, s, a3);
    };
    auto pers_args = [&](int l, float* h, int mode, long long* stp) {
        MlpPairArgs a{}; a.Wgu = wgu[l]; a.Wd = wd[l]; a.gamma = gamma; a.eps = eps; a.h = h; a.gran = gran[l]; a.epoch = epoch; a.fail = fail; a.H = H; a.I = I;
        a.stamps = stp; a.mode = mode; return a;
    };
    auto pers_layer = [&](int gw, int l, float* h, int mode, long long* stp, hipStream_t s) {
        const MlpPairArgs a = pers_args(l, h, mode, stp);
        if (gw == 105) hipLaunchKernelGGL((mlp_pair_kernel<7, 5, 4, 5, true>), dim3(G), dim3(320), 0, s, a);       // one workgroup per CU enforced by an LDS reservation
        else if (gw == 101) hipLaunchKernelGGL((mlp_pair_kernel<7, 5, 4, 1, true>), dim3(G), dim3(320), 0, s, a);
        else if (gw == 1) hipLaunchKernelGGL((mlp_pair_kernel<7, 5, 4, 1>), dim3(G), dim3(320), 0, s, a);
        else if (gw == 2) hipLaunchKernelGGL((mlp_pair_kernel<7, 5, 4, 2>), dim3(G), dim3(320), 0, s, a);
        else hipLaunchKernelGGL((mlp_pair_kernel<7, 5, 4, 5>), dim3(G), dim3(320), 0, s, a);
    };

    // ---- correctness: 24 layers applied once by both paths from the same h
    (void)hipMemcpy(h_chain, hx0.data(), H * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(h_pers, hx0.data(), H * 4, hipMemcpyHostToDevice);
    for (int l = 0; l < NL; ++l) chain_layer(l, h_chain, nullptr);
    hipLaunchKernelGGL(bump_kernel, dim3(1), dim3(64), 0, nullptr, epoch);
    for (int l = 0; l < NL; ++l) pers_layer(5, l, h_pers, 0, nullptr, nullptr);
    (void)hipDeviceSynchronize();
    std::vector<float> a(H), bb(H);
    (void)hipMemcpy(a.data(), h_chain, H * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(bb.data(), h_pers, H * 4, hipMemcpyDeviceToHost);
    int hf = 0; (void)hipMemcpy(&hf, fail, 4, hipMemcpyDeviceToHost);
    // host reference in double
    std::vector<double> hx(hx0.begin(), hx0.end());
    for (int l = 0; l < NL; ++l) {
        double ss = 0; for (int k = 0; k < H; ++k) ss += (double)(float)hx[k] * (float)hx[k];
        const float rstd = 1.0f / std::sqrt((float)(ss / H) + eps);
        std::vector<float> xn(H); for (int k = 0; k < H; ++k) xn[k] = (float)hx[k] * rstd * hgamma[k];
        std::vector<float> ac(I);
        for (int j = 0; j < I; ++j) {
            double g = 0, u = 0;
            const bf16_t* wg = &hgu[l][(size_t)(2 * j) * H]; const bf16_t* wu = wg + H;
            for (int k = 0; k < H; ++k) { g += (double)bf(wg[k]) * xn[k]; u += (double)bf(wu[k]) * xn[k]; }
            ac[j] = (float)((g / (1.0 + std::exp(-g))) * u);
        }
        for (int n = 0; n < H; ++n) {
            double d = 0; const bf16_t* wr = &hd[l][(size_t)n * I];
            for (int k = 0; k < I; ++k) d += (double)bf(wr[k]) * ac[k];
            hx[n] = (float)(hx[n] + d);
        }
    }
    double e_chain = 0, e_pers = 0, e_cp = 0, nrm = 0;
    for (int k = 0; k < H; ++k) { e_chain = std::max(e_chain, std::fabs(a[k] - hx[k])); e_pers = std::max(e_pers, std::fabs(bb[k] - hx[k])); e_cp = std::max(e_cp, (double)std::fabs(a[k] - bb[k])); nrm = std::max(nrm, std::fabs(hx[k])); }
    printf("correctness after 24 layers (max |h| %.3f): chain vs host %.3e, one-launch vs host %.3e, one-launch vs chain %.3e, give-up flag %d\n", nrm, e_chain, e_pers, e_cp, hf);
    const bool ok = hf == 0 && e_pers < 2e-4 * std::max(1.0, nrm) && e_chain < 2e-4 * std::max(1.0, nrm);
    printf("RESULT correctness: %s\n", ok ? "PASS" : "FAIL");

    // ---- repeated replays must stay correct (stale granules of the previous epoch must never be taken): 50 replays, compare with 50 chain replays
    {
        (void)hipMemcpy(h_chain, hx0.data(), H * 4, hipMemcpyHostToDevice);
        (void)hipMemcpy(h_pers, hx0.data(), H * 4, hipMemcpyHostToDevice);
        // contract the map so 50 x 24 applications stay finite: re-normalise h between replays on both sides identically (copy chain -> pers each replay is not a test);
        // instead run few layers per replay
        for (int rep = 0; rep < 50; ++rep) {
            hipLaunchKernelGGL(bump_kernel, dim3(1), dim3(64), 0, nullptr, epoch);
            const int l = rep % NL;
            chain_layer(l, h_chain, nullptr);
            pers_layer(1 + (rep % 3 == 2 ? 4 : rep % 3), l, h_pers, 0, nullptr, nullptr);
        }
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(a.data(), h_chain, H * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(bb.data(), h_pers, H * 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(&hf, fail, 4, hipMemcpyDeviceToHost);
        double d = 0, n2 = 0; for (int k = 0; k < H; ++k) { d = std::max(d, (double)std::fabs(a[k] - bb[k])); n2 = std::max(n2, (double)std::fabs(a[k])); }
        printf("50 single-layer replays, rotating gather-wave counts: one-launch vs chain max diff %.3e (max |h| %.3f), give-up flag %d -> %s\n", d, n2, hf,
               (hf == 0 && d < 1e-3 * std::max(1.0, n2)) ? "PASS" : "FAIL");
    }

    // ---- timing
    time_graph("empty <<<256,320>>> x24", NL, [&](hipStream_t s) { for (int l = 0; l < NL; ++l) hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(320), 0, s, (float*)nullptr); });
    (void)hipMemcpy(h_chain, hx0.data(), H * 4, hipMemcpyHostToDevice);
    const float t_chain = time_graph("two launches: gemv_norm<7,2,5> gate/up + gemv<10,1,4> down", NL, [&](hipStream_t s) { for (int l = 0; l < NL; ++l) chain_layer(l, h_chain, s); });
    time_graph("  gate/up alone", NL, [&](hipStream_t s) { for (int l = 0; l < NL; ++l) {
        GemvArgs a2{wgu[l], nullptr, h_chain, act, 2 * I, H, gamma, eps, nullptr, 1, st};
        hipLaunchKernelGGL((gemv_norm_kernel<7, 2, 5>), dim3((I + 19) / 20), dim3(320), 0, s, a2); } });
    time_graph("  gate/up alone, x requested BEHIND the weights (round-2 order)", NL, [&](hipStream_t s) { for (int l = 0; l < NL; ++l) {
        GemvArgs a2{wgu[l], nullptr, h_chain, act, 2 * I, H, gamma, eps, nullptr, 1, st};
        hipLaunchKernelGGL((gemv_norm_kernel<7, 2, 5, false>), dim3((I + 19) / 20), dim3(320), 0, s, a2); } });
    // the qkv-shaped launch (1152 rows, one per group) in both orders: the first 1152 rows of the gate/up matrix stand in for wqkv
    time_graph("  qkv-shaped gemv_norm<7,1> 1152x896, x first", NL, [&](hipStream_t s) { for (int l = 0; l < NL; ++l) {
        GemvArgs a0{wgu[l], nullptr, h_chain, act, 1152, H, gamma, eps, nullptr, 0, st};
        hipLaunchKernelGGL((gemv_norm_kernel<7, 1>), dim3(72), dim3(256), 0, s, a0); } });
    time_graph("  qkv-shaped gemv_norm<7,1> 1152x896, x behind the weights (round-2 order)", NL, [&](hipStream_t s) { for (int l = 0; l < NL; ++l) {
        GemvArgs a0{wgu[l], nullptr, h_chain, act, 1152, H, gamma, eps, nullptr, 0, st};
        hipLaunchKernelGGL((gemv_norm_kernel<7, 1, 4, false>), dim3(72), dim3(256), 0, s, a0); } });
    time_graph("  down alone", NL, [&](hipStream_t s) { for (int l = 0; l < NL; ++l) {
        GemvArgs a3{wd[l], nullptr, act, h_chain, H, I, nullptr, 0.f, h_chain, 0, st};
        hipLaunchKernelGGL((gemv_kernel<10, 1, 4>), dim3(H / 4), dim3(256), 0, s, a3); } });
    float t_best = 1e30f; int gw_best = 0;
    for (int gw : {1, 2, 5, 101, 105}) {
        char nm[128]; snprintf(nm, 128, "ONE launch, granule hand-off, %d gather wave(s)%s", gw % 100, gw >= 100 ? ", 1 workgroup per CU enforced" : "");
        (void)hipMemcpy(h_pers, hx0.data(), H * 4, hipMemcpyHostToDevice);
        const float t = time_graph(nm, NL, [&](hipStream_t s) {
            hipLaunchKernelGGL(bump_kernel, dim3(1), dim3(64), 0, s, epoch);
            for (int l = 0; l < NL; ++l) pers_layer(gw, l, h_pers, 0, nullptr, s); });
        if (t < t_best) { t_best = t; gw_best = gw; }
    }
    time_graph("ONE launch, hand-off NOT waited for (timing decomposition only)", NL, [&](hipStream_t s) {
        hipLaunchKernelGGL(bump_kernel, dim3(1), dim3(64), 0, s, epoch);
        for (int l = 0; l < NL; ++l) pers_layer(5, l, h_pers, 1, nullptr, s); });
    (void)hipMemcpy(&hf, fail, 4, hipMemcpyDeviceToHost);
    printf("give-up flag after the timing runs: %d\n", hf);
    printf("RESULT pair: two launches %.2f us, one launch %.2f us (%d gather waves): saves %.2f us per layer (go if >= 1.5)\n", t_chain, t_best, gw_best, t_chain - t_best);

    // ---- phase stamps of one replay (100 MHz wall clock: 10 ns units), per workgroup
    for (int gw : {1, 5, 105}) {
        (void)hipMemset(stamps, 0, (size_t)G * 64);
        hipLaunchKernelGGL(bump_kernel, dim3(1), dim3(64), 0, nullptr, epoch);
        for (int l = 0; l < 3; ++l) pers_layer(gw, l, h_pers, 0, l == 2 ? stamps : nullptr, nullptr);
        (void)hipDeviceSynchronize();
        std::vector<long long> st_h((size_t)G * 8);
        (void)hipMemcpy(st_h.data(), stamps, st_h.size() * 8, hipMemcpyDeviceToHost);
        long long t0 = st_h[0]; for (int b = 0; b < G; ++b) t0 = std::min(t0, st_h[b * 8]);
        double mx[4] = {0, 0, 0, 0}, mean[4] = {0, 0, 0, 0};
        for (int b = 0; b < G; ++b) {
            const double entry = (st_h[b * 8] - t0) * 0.01;
            const double v[4] = {entry, entry + st_h[b * 8 + 1] * 0.01, entry + st_h[b * 8 + 2] * 0.01, entry + st_h[b * 8 + 3] * 0.01};
            for (int k = 0; k < 4; ++k) { mx[k] = std::max(mx[k], v[k]); mean[k] += v[k] / G; }
        }
        printf("stamps (%d gather waves; us since the first workgroup's entry): entry mean %.2f max %.2f | act published mean %.2f max %.2f | gather done mean %.2f max %.2f | exit mean %.2f max %.2f\n",
               gw, mean[0], mx[0], mean[1], mx[1], mean[2], mx[2], mean[3], mx[3]);
    }
    return 0;
}
