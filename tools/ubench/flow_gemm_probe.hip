// Dev tool (round 3): the two GEMM shapes of the flow estimator's transformer blocks at the U10 size (M = 1348 rows; M = 5392 for 4 utterances per pass),
// round-2 kernels (flow_fused.h) against the round-3 forms (flow_gemm2.h), as dependent chains of 56 launches with 56 different weight sets inside a
// hipGraph (the estimator's 56 blocks), plus clock64() phase stamps of thread 0 of every workgroup and the distribution of workgroup start times.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I cosyvoice_amd/csrc -I include tools/ubench/flow_gemm_probe.hip -o tools/ubench/flow_gemm_probe
#include "../../cosyvoice_amd/csrc/experiments/flow_gemm2.h"
#include <vector>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <functional>
#include <algorithm>
#include <string>
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
static unsigned long long rng_state = 88172645463325252ull;
static inline unsigned long long rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }
static inline float rndf() { return (float)((rnd() >> 40) & 0xffff) / 32768.f - 1.f; }            // [-1, 1)
static inline unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (unsigned short)(u >> 16); }

typedef void (*kern_t)(FlowGemmArgs);
struct Variant { const char* name; kern_t k; int bm, bn, threads; };

static void stamps_report(const char* what, kern_t k, dim3 grid, int threads, FlowGemmArgs a, long long* dbg, int nwg) {
    (void)hipMemset(dbg, 0, (size_t)nwg * 8 * 8);
    FlowGemmArgs b = a; b.dbg = nullptr;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, grid, dim3(threads), 0, nullptr, b);
    a.dbg = dbg;
    hipLaunchKernelGGL(k, grid, dim3(threads), 0, nullptr, a);
    (void)hipDeviceSynchronize();
    std::vector<long long> h((size_t)nwg * 8);
    (void)hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost);
    long long t0 = h[0], tend = 0;
    for (int w = 0; w < nwg; ++w) { t0 = std::min(t0, h[(size_t)w * 8]); tend = std::max(tend, h[(size_t)w * 8 + 4]); }
    std::vector<long long> st(nwg), en(nwg);
    double ph[4] = {0, 0, 0, 0};
    for (int w = 0; w < nwg; ++w) {
        st[w] = h[(size_t)w * 8] - t0; en[w] = h[(size_t)w * 8 + 4] - t0;
        for (int q = 0; q < 4; ++q) ph[q] += (double)(h[(size_t)w * 8 + q + 1] - h[(size_t)w * 8 + q]) / nwg;
    }
    std::sort(st.begin(), st.end()); std::sort(en.begin(), en.end());
    printf("  stamps %-34s %4d WG | phases (mean clocks): loads+prologue %6.0f  barrier %5.0f  mfma %5.0f  epilogue %6.0f | WG start p50 %6lld p90 %6lld max %6lld | WG end p50 %6lld max %6lld\n",
           what, nwg, ph[0], ph[1], ph[2], ph[3], st[nwg / 2], st[nwg * 9 / 10], st[nwg - 1], en[nwg / 2], tend - t0);
}

int main() {
    constexpr int C = 256, INNER = 512, FF = 1024, NB = 56, MMAX = 5392;
    // operands
    std::vector<float> hx((size_t)MMAX * C);
    for (auto& v : hx) v = rndf() * 2.f;
    float* x = (float*)dmalloc(hx.size() * 4); (void)hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    float* x2 = (float*)dmalloc(hx.size() * 4);
    float* xref = (float*)dmalloc(hx.size() * 4);
    std::vector<float> hg(C), hb(C), hbias(1536);
    for (auto& v : hg) v = 1.f + 0.1f * rndf();
    for (auto& v : hb) v = 0.1f * rndf();
    for (auto& v : hbias) v = 0.1f * rndf();
    float* gam = (float*)dmalloc(C * 4); float* bet = (float*)dmalloc(C * 4); float* bias = (float*)dmalloc(1536 * 4);
    (void)hipMemcpy(gam, hg.data(), C * 4, hipMemcpyHostToDevice); (void)hipMemcpy(bet, hb.data(), C * 4, hipMemcpyHostToDevice); (void)hipMemcpy(bias, hbias.data(), 1536 * 4, hipMemcpyHostToDevice);
    auto make_w = [&](size_t numel) {
        std::vector<unsigned short> h(numel);
        for (auto& v : h) v = f2bf(rndf() * 0.06f);
        void* d = dmalloc(numel * 2); (void)hipMemcpy(d, h.data(), numel * 2, hipMemcpyHostToDevice); return (bf16_t*)d;
    };
    std::vector<bf16_t*> wqkv(NB), wff1(NB), wout(NB), wff2(NB);
    for (int i = 0; i < NB; ++i) { wqkv[i] = make_w((size_t)1536 * C); wff1[i] = make_w((size_t)FF * C); wout[i] = make_w((size_t)C * INNER); wff2[i] = make_w((size_t)C * FF); }
    bf16_t* abf = make_w((size_t)MMAX * FF);                        // bf16 activations (attention output / FF hidden)
    bf16_t* qk = (bf16_t*)dmalloc((size_t)MMAX * 1024 * 2); bf16_t* qk2 = (bf16_t*)dmalloc((size_t)MMAX * 1024 * 2);
    const int ldt = (MMAX + 63) / 64 * 64;
    bf16_t* vt = (bf16_t*)dmalloc((size_t)INNER * ldt * 2); bf16_t* vt2 = (bf16_t*)dmalloc((size_t)INNER * ldt * 2);
    bf16_t* hid = (bf16_t*)dmalloc((size_t)MMAX * FF * 2); bf16_t* hid2 = (bf16_t*)dmalloc((size_t)MMAX * FF * 2);
    long long* dbg = (long long*)dmalloc((size_t)16384 * 8 * 8);

    auto ln_args = [&](int blk, int M, bool qkv, bf16_t* o, bf16_t* oT) {
        FlowGemmArgs a{}; a.A = x; a.lda = C; a.gamma = gam; a.beta = bet; a.eps = 1e-5f; a.W = qkv ? wqkv[blk] : wff1[blk]; a.Kp = C; a.bias = bias;
        a.M = M; a.N = qkv ? 1536 : FF; a.K = C; a.act = qkv ? ACT_NONE : ACT_GELU_ERF; a.out = o; a.ldo = qkv ? 1024 : FF; a.n_row = qkv ? 1024 : FF;
        a.outT = qkv ? oT : nullptr; a.t_batch = (long long)INNER * ldt; a.ldt = ldt; a.rows_per_batch = M; return a;
    };
    auto res_args = [&](int blk, int M, bool ff2, float* c) {
        FlowGemmArgs a{}; a.A = abf; a.lda = ff2 ? FF : INNER; a.W = ff2 ? wff2[blk] : wout[blk]; a.Kp = ff2 ? FF : INNER; a.bias = bias; a.M = M; a.N = C; a.K = ff2 ? FF : INNER;
        a.C = c; a.ldc = C; a.res = x; a.n_row = C; return a;
    };
    auto grid_of = [](int M, int N, int bm, int bn) { return dim3((unsigned)(((M + bm - 1) / bm) * ((N + bn - 1) / bn))); };

    const std::vector<Variant> lnv = {
        {"round 2: flow_gemm<32,64,1,0> (W + A via LDS)", flow_gemm_kernel<32, 64, 1, 0>, 32, 64, 256},
        {"flow_ln_gemm<32,4>  32 x 64,  4 waves", flow_ln_gemm_kernel<32, 4>, 32, 64, 256},
        {"flow_ln_gemm<16,4>  16 x 64,  4 waves", flow_ln_gemm_kernel<16, 4>, 16, 64, 256},
        {"flow_ln_gemm<64,4>  64 x 64,  4 waves", flow_ln_gemm_kernel<64, 4>, 64, 64, 256},
        {"flow_ln_gemm<32,8>  32 x 128, 8 waves", flow_ln_gemm_kernel<32, 8>, 32, 128, 512},
        {"flow_ln_gemm<16,8>  16 x 128, 8 waves", flow_ln_gemm_kernel<16, 8>, 16, 128, 512},
        {"flow_ln_gemm<32,2>  32 x 32,  2 waves", flow_ln_gemm_kernel<32, 2>, 32, 32, 128},
        {"flow_ln_gemm<16,2>  16 x 32,  2 waves", flow_ln_gemm_kernel<16, 2>, 16, 32, 128},
    };
    const std::vector<Variant> outv = {
        {"round 2: flow_gemm<32,64,0,1> (256-k chunks via LDS)", flow_gemm_kernel<32, 64, 0, 1>, 32, 64, 256},
        {"flow_res_gemm<32,4,2>  32 x 64,  8 waves", flow_res_gemm_kernel<32, 4, 2>, 32, 64, 512},
        {"flow_res_gemm<16,4,2>  16 x 64,  8 waves", flow_res_gemm_kernel<16, 4, 2>, 16, 64, 512},
        {"flow_res_gemm<32,2,2>  32 x 32,  4 waves", flow_res_gemm_kernel<32, 2, 2>, 32, 32, 256},
        {"flow_res_gemm<16,2,2>  16 x 32,  4 waves", flow_res_gemm_kernel<16, 2, 2>, 16, 32, 256},
        {"flow_res_gemm<16,8,2>  16 x 128, 16 waves", flow_res_gemm_kernel<16, 8, 2>, 16, 128, 1024},
    };
    const std::vector<Variant> ff2v = {
        {"round 2: flow_gemm<32,64,0,1> (256-k chunks via LDS)", flow_gemm_kernel<32, 64, 0, 1>, 32, 64, 256},
        {"flow_res_gemm<32,4,4>  32 x 64, 16 waves", flow_res_gemm_kernel<32, 4, 4>, 32, 64, 1024},
        {"flow_res_gemm<16,4,4>  16 x 64, 16 waves", flow_res_gemm_kernel<16, 4, 4>, 16, 64, 1024},
        {"flow_res_gemm<32,2,4>  32 x 32,  8 waves", flow_res_gemm_kernel<32, 2, 4>, 32, 32, 512},
        {"flow_res_gemm<16,2,4>  16 x 32,  8 waves", flow_res_gemm_kernel<16, 2, 4>, 16, 32, 512},
        {"flow_res_gemm<16,1,4>  16 x 16,  4 waves", flow_res_gemm_kernel<16, 1, 4>, 16, 16, 256},
    };

    // ---- empty-kernel floor of the chain
    for (int M : {1348, 5392}) {
        printf("==== M = %d rows\n", M);
        for (int which = 0; which < 2; ++which) {
            const bool qkv = which == 0;
            printf("-- LayerNorm -> %s  (K = 256, N = %d)\n", qkv ? "Q | K | V^T" : "FF1 + GELU", qkv ? 1536 : FF);
            // reference output of the round-2 kernel for the correctness check
            { const FlowGemmArgs a = ln_args(3, M, qkv, qkv ? qk : hid, vt); hipLaunchKernelGGL(lnv[0].k, grid_of(M, a.N, 32, 64), dim3(256), 0, nullptr, a); }
            (void)hipDeviceSynchronize();
            for (const Variant& v : lnv) {
                const int N = qkv ? 1536 : FF;
                const dim3 g = grid_of(M, N, v.bm, v.bn);
                (void)hipMemset(qkv ? (void*)qk2 : (void*)hid2, 0, (size_t)M * (qkv ? 1024 : FF) * 2); (void)hipMemset(vt2, 0, (size_t)INNER * ldt * 2);
                { const FlowGemmArgs a = ln_args(3, M, qkv, qkv ? qk2 : hid2, vt2); hipLaunchKernelGGL(v.k, g, dim3(v.threads), 0, nullptr, a); }
                (void)hipDeviceSynchronize();
                std::vector<unsigned short> r((size_t)M * (qkv ? 1024 : FF)), t(r.size());
                (void)hipMemcpy(r.data(), qkv ? qk : hid, r.size() * 2, hipMemcpyDeviceToHost); (void)hipMemcpy(t.data(), qkv ? qk2 : hid2, t.size() * 2, hipMemcpyDeviceToHost);
                size_t bad = 0; for (size_t i = 0; i < r.size(); ++i) bad += r[i] != t[i];
                if (qkv) {
                    std::vector<unsigned short> rv((size_t)INNER * ldt), tv(rv.size());
                    (void)hipMemcpy(rv.data(), vt, rv.size() * 2, hipMemcpyDeviceToHost); (void)hipMemcpy(tv.data(), vt2, tv.size() * 2, hipMemcpyDeviceToHost);
                    for (size_t i = 0; i < rv.size(); ++i) bad += rv[i] != tv[i];
                }
                const float us = time_graph(NB, [&](hipStream_t s) {
                    for (int b = 0; b < NB; ++b) { const FlowGemmArgs a = ln_args(b, M, qkv, qkv ? qk2 : hid2, vt2); hipLaunchKernelGGL(v.k, g, dim3(v.threads), 0, s, a); } });
                printf("%-58s %5u WG  %7.2f us per launch   %zu elements differ from round 2\n", v.name, g.x, us, bad); fflush(stdout);
                if (M == 1348) stamps_report(v.name, v.k, g, v.threads, ln_args(5, M, qkv, qkv ? qk2 : hid2, vt2), dbg, (int)g.x);
            }
        }
        for (int which = 0; which < 2; ++which) {
            const bool ff2 = which == 1;
            const auto& vs = ff2 ? ff2v : outv;
            printf("-- %s + bias + residual  (K = %d, N = 256)\n", ff2 ? "FF2" : "out-projection", ff2 ? FF : INNER);
            { const FlowGemmArgs a = res_args(3, M, ff2, xref); hipLaunchKernelGGL(vs[0].k, grid_of(M, C, 32, 64), dim3(256), 0, nullptr, a); }
            (void)hipDeviceSynchronize();
            for (const Variant& v : vs) {
                const dim3 g = grid_of(M, C, v.bm, v.bn);
                (void)hipMemset(x2, 0, (size_t)M * C * 4);
                { const FlowGemmArgs a = res_args(3, M, ff2, x2); hipLaunchKernelGGL(v.k, g, dim3(v.threads), 0, nullptr, a); }
                (void)hipDeviceSynchronize();
                std::vector<float> r((size_t)M * C), t(r.size());
                (void)hipMemcpy(r.data(), xref, r.size() * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(t.data(), x2, t.size() * 4, hipMemcpyDeviceToHost);
                double md = 0, mx = 0; for (size_t i = 0; i < r.size(); ++i) { md = std::max(md, (double)std::fabs(r[i] - t[i])); mx = std::max(mx, (double)std::fabs(r[i])); }
                const float us = time_graph(NB, [&](hipStream_t s) {
                    for (int b = 0; b < NB; ++b) { const FlowGemmArgs a = res_args(b, M, ff2, x2); hipLaunchKernelGGL(v.k, g, dim3(v.threads), 0, s, a); } });
                printf("%-58s %5u WG  %7.2f us per launch   max |this - round 2| = %.3e (max |value| %.2f)\n", v.name, g.x, us, md, mx); fflush(stdout);
                if (M == 1348) stamps_report(v.name, v.k, g, v.threads, res_args(5, M, ff2, x2), dbg, (int)g.x);
            }
        }
    }
    return 0;
}
