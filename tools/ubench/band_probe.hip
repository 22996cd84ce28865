// Dev tool (round 5): where does a launch of flow_band_kernel (flow_band.h) spend its time?
//   * launch time at M = 10 784 (169 bands), 5 392 and 2 696 rows with 56 different weight streams (cold, as in an Euler step) and one re-used (L2-warm);
//   * the same with parts removed (template MODE): 1 = the weight stream requested once only, 2 = the stream alone (no MFMA, no fragment reads),
//     3 = MFMAs on register operands (no LDS fragment reads);
//   * clock64() stamps of thread 0 at the phase boundaries, averaged over the workgroups.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I cosyvoice_amd/csrc -I include tools/ubench/band_probe.hip -o tools/ubench/band_probe
#include "../../cosyvoice_amd/csrc/flow_band.h"
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

template <int MODE>
static void launch(const FlowBandArgs& a, hipStream_t s) {
    hipLaunchKernelGGL((flow_band_kernel<256, 512, 1024, true, 8, MODE>), dim3((a.M + 63) / 64), dim3(512), 0, s, a);
}

int main() {
    constexpr int C = 256, INNER = 512, FF = 1024, NB = 56, MMAX = 10784;
    using S = FlowBandShape<C, INNER, FF, 8>;
    auto dmalloc = [](size_t b) { void* p; if (hipMalloc(&p, b) != hipSuccess) { printf("hipMalloc failed\n"); exit(1); } (void)hipMemset(p, 0, b); return p; };
    const size_t stream_bytes = (size_t)8 * S::TOTAL * 64 * 16;
    std::vector<u32x4_t*> ws(NB);
    std::vector<unsigned short> hw(std::max(stream_bytes, (size_t)MMAX * INNER * 2) / 2);
    unsigned long long z = 88172645463325252ull;
    for (auto& v : hw) { z ^= z << 13; z ^= z >> 7; z ^= z << 17; v = (unsigned short)(0x3a00u + (z & 0x1ff)) ^ (unsigned short)((z >> 20) & 0x8000u); }      // small bf16 values of both signs
    for (int i = 0; i < NB; ++i) { ws[i] = (u32x4_t*)dmalloc(stream_bytes); (void)hipMemcpy(ws[i], hw.data(), stream_bytes, hipMemcpyHostToDevice); }
    bf16_t* att = (bf16_t*)dmalloc((size_t)MMAX * INNER * 2); (void)hipMemcpy(att, hw.data(), (size_t)MMAX * INNER * 2, hipMemcpyHostToDevice);
    float* x = (float*)dmalloc((size_t)MMAX * C * 4);
    float* prm = (float*)dmalloc((size_t)(6 * C + FF) * 4);
    std::vector<float> hp(6 * C + FF, 0.01f); (void)hipMemcpy(prm, hp.data(), hp.size() * 4, hipMemcpyHostToDevice);
    bf16_t* xn = (bf16_t*)dmalloc((size_t)MMAX * C * 2);
    long long* dbg = (long long*)dmalloc((size_t)(MMAX / 64 + 1) * 16 * 8);
    auto args = [&](int blk, int M, long long* d) {
        FlowBandArgs a{}; a.att = att; a.ld_att = INNER; a.x = x; a.ldx = C; a.wstream = ws[blk]; a.prm = prm; a.eps = 1e-5f; a.M = M; a.xn = xn; a.ld_xn = C; a.dbg = d; return a;
    };
    printf("flow_band_kernel<256,512,1024,next LN, 8 waves>: %d fragments of 1 KB per wave, %.2f MB of weights per workgroup, %.1f MFLOP per band\n", S::TOTAL, stream_bytes / 1e6,
           2.0 * 64 * (C * INNER + 2.0 * C * FF) / 1e6);
    const char* mode_names[4] = {"the kernel", "weight stream requested once", "stream alone (no MFMA, no LDS fragment reads)", "MFMAs on register operands (no LDS fragment reads)"};
    for (int mode = 0; mode < 4; ++mode) {
        for (int cold = 1; cold >= 0; --cold) {
            printf("%-52s %s:", mode_names[mode], cold ? "cold" : "warm");
            for (int M : {64, 2696, 5392, 10784, 16384 > MMAX ? MMAX : 16384}) {
                const float us = time_graph(NB, [&](hipStream_t s) {
                    for (int b = 0; b < NB; ++b) {
                        const FlowBandArgs a = args(cold ? b : 0, M, nullptr);
                        if (mode == 0) launch<0>(a, s); else if (mode == 1) launch<1>(a, s); else if (mode == 2) launch<2>(a, s); else launch<3>(a, s);
                    } });
                printf("  M=%d (%d WG): %.1f us", M, (M + 63) / 64, us);
            }
            printf("\n"); fflush(stdout);
        }
    }
    for (int M : {64, 10784}) {
        for (int rep = 0; rep < 3; ++rep) { const FlowBandArgs a = args(7 + rep, M, dbg); launch<0>(a, nullptr); }
        (void)hipDeviceSynchronize();
        const int nwg = (M + 63) / 64;
        std::vector<long long> h((size_t)nwg * 16);
        (void)hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost);
        const char* names[9] = {"staging of the band's operands, barrier", "A out-projection (32 fragments per wave)", "B LayerNorm", "C/D chunk 0 (FF1 + GELU, FF2: 32 fragments)", "C/D chunk 1",
                                "C/D chunk 2", "C/D chunk 3", "FF2 epilogue, next LayerNorm, bf16 rows out", "write-out of x"};
        printf("phase durations of thread 0, M = %d (%d workgroups), shader clocks (mean over workgroups; max of the total):\n", M, nwg);
        double tot_mean = 0; long long tot_max = 0;
        for (int k = 0; k < 9; ++k) {
            double m = 0; for (int w = 0; w < nwg; ++w) m += (double)(h[(size_t)w * 16 + k + 1] - h[(size_t)w * 16 + k]) / nwg;
            printf("  %-52s %9.0f\n", names[k], m); tot_mean += m;
        }
        for (int w = 0; w < nwg; ++w) tot_max = std::max(tot_max, h[(size_t)w * 16 + 9] - h[(size_t)w * 16]);
        printf("  %-52s %9.0f (max %lld)\n", "total", tot_mean, tot_max);
    }
    return 0;
}
