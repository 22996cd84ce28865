// Dev tool (round 3): what does ONE launch of flow_tail_kernel cost, and where inside a workgroup does the time go?
//   * launch time against the number of 16-row bands (1 .. 337 workgroups) - a per-workgroup dependent chain shows as a time that does NOT grow with the
//     band count, a shared-resource limit (L2 / fabric) as one that does - with 56 different weight streams (the estimator's 56 blocks: 112 MB, cold L2 per
//     launch as in the real Euler step) and with ONE stream re-used (L2-warm);
//   * clock64() stamps of thread 0 at the phase boundaries (FlowTailArgs::dbg), averaged over the workgroups.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I cosyvoice_amd/csrc -I include tools/ubench/tail_probe.hip -o tools/ubench/tail_probe
#include "../../cosyvoice_amd/csrc/experiments/flow_tail.h"
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

int main() {
    constexpr int C = 256, INNER = 512, FF = 1024, NB = 56, MMAX = 5392;
    using S = FlowTailShape<C, INNER, FF, true>;
    auto dmalloc = [](size_t b) { void* p; if (hipMalloc(&p, b) != hipSuccess) { printf("hipMalloc failed\n"); exit(1); } (void)hipMemset(p, 0, b); return p; };
    const size_t stream_bytes = (size_t)4 * S::TOTAL * 64 * 16;
    std::vector<u32x4_t*> ws(NB);
    std::vector<unsigned short> hw(stream_bytes / 2);
    unsigned long long z = 88172645463325252ull;
    for (auto& v : hw) { z ^= z << 13; z ^= z >> 7; z ^= z << 17; v = (unsigned short)(0x3c00u + (z & 0x1ff)) ^ (unsigned short)((z >> 20) & 0x8000u); }      // small bf16 values of both signs
    for (int i = 0; i < NB; ++i) { ws[i] = (u32x4_t*)dmalloc(stream_bytes); (void)hipMemcpy(ws[i], hw.data(), stream_bytes, hipMemcpyHostToDevice); }
    bf16_t* att = (bf16_t*)dmalloc((size_t)MMAX * INNER * 2); (void)hipMemcpy(att, hw.data(), std::min(stream_bytes, (size_t)MMAX * INNER * 2), hipMemcpyHostToDevice);
    float* x = (float*)dmalloc((size_t)MMAX * C * 4);
    float* prm = (float*)dmalloc((size_t)(6 * C + FF) * 4);
    std::vector<float> hp(6 * C + FF, 0.5f); (void)hipMemcpy(prm, hp.data(), hp.size() * 4, hipMemcpyHostToDevice);
    bf16_t* qk = (bf16_t*)dmalloc((size_t)MMAX * 2 * INNER * 2);
    const int ldt = (MMAX + 63) / 64 * 64;
    bf16_t* vt = (bf16_t*)dmalloc((size_t)INNER * ldt * 2 * 2);
    long long* dbg = (long long*)dmalloc((size_t)(MMAX / 16 + 1) * 16 * 8);
    auto args = [&](int blk, int M, long long* d) {
        FlowTailArgs a{}; a.att = att; a.ld_att = INNER; a.x = x; a.ldx = C; a.wstream = ws[blk]; a.prm = prm; a.eps = 1e-5f; a.M = M;
        a.qk = qk; a.ld_qk = 2 * INNER; a.vt = vt; a.vt_batch = (long long)INNER * ldt; a.ldt = ldt; a.rows_per_batch = M; a.dbg = d; return a;
    };
    printf("flow_tail_kernel<256,512,1024,next QKV>: %d fragments of 1 KB per wave, %.2f MB of weights per workgroup\n", S::TOTAL, stream_bytes / 1e6);
    for (int ring : {8, 16}) {
        for (int cold = 1; cold >= 0; --cold) {
            printf("ring %2d, %s:", ring, cold ? "56 different weight streams (cold)" : "one weight stream re-used (L2-warm)");
            for (int M : {16, 128, 256, 674, 1348, 2696, 5392}) {
                const float us = time_graph(NB, [&](hipStream_t s) {
                    for (int b = 0; b < NB; ++b) {
                        const FlowTailArgs a = args(cold ? b : 0, M, nullptr);
                        if (ring == 8) hipLaunchKernelGGL((flow_tail_kernel<C, INNER, FF, true, 8>), dim3((M + 15) / 16), dim3(256), 0, s, a);
                        else hipLaunchKernelGGL((flow_tail_kernel<C, INNER, FF, true, 16>), dim3((M + 15) / 16), dim3(256), 0, s, a);
                    } });
                printf("  M=%d (%d WG): %.1f us", M, (M + 15) / 16, us);
            }
            printf("\n"); fflush(stdout);
        }
    }
    // phase stamps at M = 1348 (85 workgroups), ring 8, cold stream
    for (int M : {16, 1348}) {
        for (int rep = 0; rep < 3; ++rep) {
            const FlowTailArgs a = args(7 + rep, M, dbg);
            hipLaunchKernelGGL((flow_tail_kernel<C, INNER, FF, true, 8>), dim3((M + 15) / 16), dim3(256), 0, nullptr, a);
        }
        (void)hipDeviceSynchronize();
        const int nwg = (M + 15) / 16;
        std::vector<long long> h((size_t)nwg * 16);
        (void)hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost);
        const char* names[8] = {"operand + ring requests, LDS staging, barrier", "A out-projection (64 fragments)", "B LayerNorm", "C FF1 + GELU (128 fragments)", "D FF2 (128 fragments)",
                                "E LayerNorm", "F Q, K, V^T (192 fragments) + V^T scatter", "write-out of x and Q | K"};
        printf("phase durations of thread 0, M = %d (%d workgroups), shader clocks (mean over workgroups; max of the total):\n", M, nwg);
        double tot_mean = 0; long long tot_max = 0;
        for (int k = 0; k < 8; ++k) {
            double m = 0; for (int w = 0; w < nwg; ++w) m += (double)(h[(size_t)w * 16 + k + 1] - h[(size_t)w * 16 + k]) / nwg;
            printf("  %-52s %9.0f\n", names[k], m); tot_mean += m;
        }
        for (int w = 0; w < nwg; ++w) tot_max = std::max(tot_max, h[(size_t)w * 16 + 8] - h[(size_t)w * 16]);
        printf("  %-52s %9.0f (max %lld)\n", "total", tot_mean, tot_max);
    }
    return 0;
}
