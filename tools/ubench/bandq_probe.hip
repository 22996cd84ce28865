// Dev tool (round 5): what does the QKV phase of flow_band_kernel<.., HAS_QKV> (flow_band.h) cost?
//   * launch time at M = 10 784 / 5 392 rows with 56 different weight streams, for 64- / 48- / 32-row bands: the band alone (next LayerNorm rows out), the band with the next
//     block's QKV GEMM, and the latter without its stores (MODE 4: the products kept by a register checksum);
//   * clock64() stamps of thread 0 at the phase boundaries of the HAS_QKV form.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I cosyvoice_amd/csrc -I include tools/ubench/bandq_probe.hip -o tools/ubench/bandq_probe
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

template <int BM, int FORM>      // FORM 0: band (next LayerNorm rows out); 1: band + QKV; 2: band + QKV without its stores; 3: band + QKV, pipelined FF chunks (<= 48 rows)
static void launch(const FlowBandArgs& a, hipStream_t s) {
    const dim3 g((a.M + BM - 1) / BM), b(512);
    if constexpr (FORM == 0) hipLaunchKernelGGL((flow_band_kernel<256, 512, 1024, true, 8, 0, BM, false>), g, b, 0, s, a);
    else if constexpr (FORM == 1) hipLaunchKernelGGL((flow_band_kernel<256, 512, 1024, true, 8, 0, BM, true>), g, b, 0, s, a);
    else if constexpr (FORM == 2) hipLaunchKernelGGL((flow_band_kernel<256, 512, 1024, true, 8, 4, BM, true>), g, b, 0, s, a);
    else if constexpr (BM <= 48) hipLaunchKernelGGL((flow_band_kernel<256, 512, 1024, true, 8, 0, BM, true, true>), g, b, 0, s, a);
    else hipLaunchKernelGGL((flow_band_kernel<256, 512, 1024, true, 8, 0, BM, true>), g, b, 0, s, a);       // (64 rows: no pipelined form)
}

int main() {
    constexpr int C = 256, INNER = 512, FF = 1024, NB = 56, MMAX = 10784, T = 674;
    using S = FlowBandShape<C, INNER, FF, 8>;
    auto dmalloc = [](size_t b) { void* p; if (hipMalloc(&p, b) != hipSuccess) { printf("hipMalloc failed\n"); exit(1); } (void)hipMemset(p, 0, b); return p; };
    const size_t stream_bytes = (size_t)8 * S::TOTALQ * 64 * 16;
    std::vector<u32x4_t*> ws(NB);
    std::vector<unsigned short> hw(std::max(stream_bytes, (size_t)MMAX * INNER * 2) / 2);
    unsigned long long z = 88172645463325252ull;
    for (auto& v : hw) { z ^= z << 13; z ^= z >> 7; z ^= z << 17; v = (unsigned short)(0x3a00u + (z & 0x1ff)) ^ (unsigned short)((z >> 20) & 0x8000u); }
    for (int i = 0; i < NB; ++i) { ws[i] = (u32x4_t*)dmalloc(stream_bytes); (void)hipMemcpy(ws[i], hw.data(), stream_bytes, hipMemcpyHostToDevice); }
    bf16_t* att = (bf16_t*)dmalloc((size_t)MMAX * INNER * 2); (void)hipMemcpy(att, hw.data(), (size_t)MMAX * INNER * 2, hipMemcpyHostToDevice);
    float* x = (float*)dmalloc((size_t)MMAX * C * 4);
    float* prm = (float*)dmalloc((size_t)(6 * C + FF) * 4);
    std::vector<float> hp(6 * C + FF, 0.01f); (void)hipMemcpy(prm, hp.data(), hp.size() * 4, hipMemcpyHostToDevice);
    bf16_t* xn = (bf16_t*)dmalloc((size_t)MMAX * C * 2);
    bf16_t* qk = (bf16_t*)dmalloc((size_t)MMAX * 2 * INNER * 2);
    const int ldt = (T + 63) / 64 * 64;
    bf16_t* vt = (bf16_t*)dmalloc((size_t)(MMAX / T) * INNER * ldt * 2);
    long long* dbg = (long long*)dmalloc((size_t)(MMAX / 32 + 1) * 16 * 8);
    auto args = [&](int blk, int M, long long* d) {
        FlowBandArgs a{}; a.att = att; a.ld_att = INNER; a.x = x; a.ldx = C; a.wstream = ws[blk]; a.prm = prm; a.eps = 1e-5f; a.M = M; a.xn = xn; a.ld_xn = C; a.dbg = d;
        a.qk = qk; a.ld_qk = 2 * INNER; a.vt = vt; a.vt_batch = (long long)INNER * ldt; a.ldt = ldt; a.rows_per_batch = T; return a;
    };
    // NOTE: the FORM 0 kernel walks the same buffers with the stride of the shorter stream (S::TOTAL): synthetic values, only the time matters
    printf("flow_band_kernel<256,512,1024, 8 waves>: %d + %d fragments of 1 KB per wave\n", S::TOTAL, S::TOTALQ - S::TOTAL);
    const char* form_names[4] = {"band (next LayerNorm rows out)", "band + next QKV GEMM", "band + next QKV GEMM, stores removed", "band + next QKV GEMM, pipelined FF chunks"};
    for (int M : {10784, 5392}) {
        for (int form = 0; form < 4; ++form) {
            printf("M = %5d  %-40s", M, form_names[form]);
            for (int bm : {64, 48, 32}) {
                const float us = time_graph(NB, [&](hipStream_t s) {
                    for (int b = 0; b < NB; ++b) {
                        const FlowBandArgs a = args(b, M, nullptr);
#define L(BM_) { if (form == 0) launch<BM_, 0>(a, s); else if (form == 1) launch<BM_, 1>(a, s); else if (form == 2) launch<BM_, 2>(a, s); else launch<BM_, 3>(a, s); }
                        if (bm == 64) L(64) else if (bm == 48) L(48) else L(32)
#undef L
                    } });
                printf("  %d rows (%d WG): %.1f us", bm, (M + bm - 1) / bm, us);
            }
            printf("\n"); fflush(stdout);
        }
    }
    for (int bm : {64, 48, -48}) {                                            // -48: 48 rows, pipelined FF chunks
        const bool pipe = bm < 0; if (pipe) bm = -bm;
        const int M = 10784, nwg = (M + bm - 1) / bm;
        for (int rep = 0; rep < 3; ++rep) { const FlowBandArgs a = args(7 + rep, M, dbg); if (bm == 64) launch<64, 1>(a, nullptr); else if (pipe) launch<48, 3>(a, nullptr); else launch<48, 1>(a, nullptr); }
        (void)hipDeviceSynchronize();
        std::vector<long long> h((size_t)nwg * 16);
        (void)hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost);
        const char* names[10] = {"staging of the band's operands, barrier", "A out-projection", "B LayerNorm", "C/D chunk 0", "C/D chunk 1", "C/D chunk 2", "C/D chunk 3",
                                 "FF2 epilogue, next LayerNorm", "F next QKV GEMM (6 passes) + stores", "write-out of x"};
        printf("phase durations of thread 0, %d-row bands%s, M = %d (%d workgroups), shader clocks (mean over workgroups; max of the total):\n", bm, pipe ? ", pipelined FF chunks (stages 0 - 3, then the last FF2 + epilogue)" : "", M, nwg);
        double tot_mean = 0; long long tot_max = 0;
        for (int k = 0; k < 10; ++k) {
            double m = 0; for (int w = 0; w < nwg; ++w) m += (double)(h[(size_t)w * 16 + k + 1] - h[(size_t)w * 16 + k]) / nwg;
            printf("  %-52s %9.0f\n", names[k], m); tot_mean += m;
        }
        for (int w = 0; w < nwg; ++w) tot_max = std::max(tot_max, h[(size_t)w * 16 + 10] - h[(size_t)w * 16]);
        printf("  %-52s %9.0f (max %lld)\n", "total", tot_mean, tot_max);
    }
    return 0;
}
