// Dev tool (round 6): the flow estimator's bf16 flash attention ALONE - attn_flow_kernel (flow_fused.h, rounds 2-5) next to attn_flow32_kernel (flow_attn32.h) on the
// shapes of the bench (B = 2 x utterances, H = 8, T = 674, head_dim 64; chunk mask = the streaming passes), random data in the layouts the QKV epilogues write
// (Q | K row-major [B T][1024], V^T [B][512][ldt] key-permuted by vt_col).  Every variant is checked against a plain fp32 attention of the same bf16 inputs (one
// thread per query, no rounding of P) and timed as 20 back-to-back launches between one event pair, best of 5.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I cosyvoice_amd/csrc -I include tools/ubench/attn_probe.hip -o tools/ubench/attn_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <string>
#include <algorithm>
#include <map>
#include "flow_attn32.h"
using namespace cv;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (unsigned short)(u >> 16); }
static float bf2f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }
static int h_vt_col(int t) { return (t & ~31) + ((t >> 2) & 3) * 8 + ((t >> 4) & 1) * 4 + (t & 3); }

__global__ void ref_kernel(const bf16_t* qk, int ld, int inner, const bf16_t* vt, long long vt_batch, int ldt, float* o, int B, int H, int T, float scale, int chunk, const int* klen) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * H * T) return;
    const int q = idx % T, h = (idx / T) % H, b = idx / (T * H);
    const int Tk = klen ? min(T, klen[b]) : T;
    const int klim = chunk > 0 ? min(Tk, (q / chunk + 1) * chunk) : Tk;
    const bf16_t* qp = qk + ((long long)b * T + q) * ld + h * 64;
    float qv[64];
    for (int d = 0; d < 64; ++d) qv[d] = bf16_to_f32(qp[d]);
    float m = -INFINITY;
    for (int k = 0; k < klim; ++k) {
        const bf16_t* kp = qk + ((long long)b * T + k) * ld + inner + h * 64;
        float s = 0.f;
        for (int d = 0; d < 64; ++d) s += qv[d] * bf16_to_f32(kp[d]);
        m = fmaxf(m, s * scale);
    }
    float l = 0.f, acc[64];
    for (int d = 0; d < 64; ++d) acc[d] = 0.f;
    for (int k = 0; k < klim; ++k) {
        const bf16_t* kp = qk + ((long long)b * T + k) * ld + inner + h * 64;
        float s = 0.f;
        for (int d = 0; d < 64; ++d) s += qv[d] * bf16_to_f32(kp[d]);
        const float e = expf(s * scale - m);
        l += e;
        const int col = (k & ~31) + ((k >> 2) & 3) * 8 + ((k >> 4) & 1) * 4 + (k & 3);
        for (int d = 0; d < 64; ++d) acc[d] += e * bf16_to_f32(vt[(long long)b * vt_batch + (long long)(h * 64 + d) * ldt + col]);
    }
    for (int d = 0; d < 64; ++d) o[((long long)b * T + q) * inner + h * 64 + d] = klim > 0 ? acc[d] / l : 0.f;
}

struct Case { int B, H, T, chunk; bool ragged; float qscale; };

int main(int argc, char** argv) {
    const int only = argc > 1 ? atoi(argv[1]) : -1;          // attn_probe <case> <variant substring>: one case, the variants whose name contains the string (rocprofv3 --pmc passes)
    const char* vsel = argc > 2 ? argv[2] : nullptr;
    std::vector<Case> all_cases = {{16, 8, 674, 0, false, 1.f}, {2, 8, 674, 0, false, 1.f}, {16, 8, 674, 50, false, 1.f}, {4, 8, 1174, 0, false, 1.f}, {16, 8, 250, 50, false, 1.f}, {6, 8, 301, 0, true, 4.f}, {2, 16, 674, 0, false, 1.f}};
    std::vector<Case> cases;
    for (int i = 0; i < (int)all_cases.size(); ++i) if (only < 0 || i == only) cases.push_back(all_cases[i]);
    const int reps = 20;
    for (const Case& c : cases) {
        const int B = c.B, H = c.H, T = c.T, inner = H * 64, ld = 2 * inner, ldt = (T + 63) / 64 * 64;
        const long long vt_batch = (long long)inner * ldt;
        std::vector<unsigned short> hqk((size_t)B * T * ld), hvt((size_t)B * vt_batch);
        srand(1234 + T);
        auto rnd = [&]() { return (float)rand() / RAND_MAX * 2.f - 1.f; };
        // scores with a realistic spread: q, k ~ U(-1, 1) * qscale * 1.5 -> s * scale has std ~ qscale^2 * 0.75 ... a few keys dominate at qscale = 4
        for (auto& v : hqk) v = f2bf(rnd() * 1.5f * c.qscale);
        for (auto& v : hvt) v = f2bf(rnd() * 2.f);          // pad columns included: finite
        std::vector<int> hklen(B, T);
        if (c.ragged) for (int b = 0; b < B; ++b) hklen[b] = T - 37 * (b % 4) - (b == 1 ? 200 : 0);
        bf16_t *dqk, *dvt, *dout; float* dref; int* dklen;
        CK(hipMalloc(&dqk, hqk.size() * 2)); CK(hipMalloc(&dvt, hvt.size() * 2)); CK(hipMalloc(&dout, (size_t)B * T * inner * 2)); CK(hipMalloc(&dref, (size_t)B * T * inner * 4)); CK(hipMalloc(&dklen, B * 4));
        CK(hipMemcpy(dqk, hqk.data(), hqk.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dvt, hvt.data(), hvt.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dklen, hklen.data(), B * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(ref_kernel, dim3((B * H * T + 63) / 64), dim3(64), 0, 0, dqk, ld, inner, dvt, vt_batch, ldt, dref, B, H, T, 0.125f, c.chunk, c.ragged ? dklen : nullptr);
        CK(hipDeviceSynchronize());
        std::vector<float> href((size_t)B * T * inner);
        CK(hipMemcpy(href.data(), dref, href.size() * 4, hipMemcpyDeviceToHost));
        AttnFlowArgs a{};
        a.q = dqk; a.k = dqk + inner; a.ld = ld; a.vt = dvt; a.vt_batch = vt_batch; a.ldt = ldt; a.o = dout; a.ldo = inner; a.B = B; a.H = H; a.T = T; a.scale = 0.125f;
        a.mask_mode = c.chunk > 0 ? MASK_CHUNK : MASK_NONE; a.chunk = c.chunk; a.klen = c.ragged ? dklen : nullptr;
        printf("== B %d H %d T %d chunk %d ragged %d qscale %.0f: %.2f GFLOP (unmasked count)\n", B, H, T, c.chunk, (int)c.ragged, c.qscale, 4.0 * B * H * (double)T * T * 64 / 1e9);
        auto run = [&](const char* name, auto launch) {
            if (vsel && !strstr(name, vsel)) return;
            CK(hipMemset(dout, 0xff, (size_t)B * T * inner * 2));
            launch(); CK(hipDeviceSynchronize());
            std::vector<unsigned short> hout((size_t)B * T * inner);
            CK(hipMemcpy(hout.data(), dout, hout.size() * 2, hipMemcpyDeviceToHost));
            double maxerr = 0, sumerr = 0; long long nbad = 0, n = 0;
            for (int b = 0; b < B; ++b)
                for (int t = 0; t < (c.ragged ? hklen[b] : T); ++t)          // padded rows are not part of any result
                    for (int j = 0; j < inner; ++j) {
                        const size_t i = ((size_t)b * T + t) * inner + j;
                        const double e = fabs((double)bf2f(hout[i]) - href[i]);
                        if (!(e <= 1e30)) ++nbad;
                        else { maxerr = fmax(maxerr, e); sumerr += e; }
                        ++n;
                    }
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            float best = 1e30f;
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipEventRecord(e0, 0));
                for (int i = 0; i < reps; ++i) launch();
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = fminf(best, ms);
            }
            const double us = best * 1e3 / reps, gf = 4.0 * B * H * (double)T * T * 64 / 1e9;
            printf("  %-34s %8.2f us  %7.1f TFLOP/s   max |err| %.4f  mean %.5f  nan/inf %lld\n", name, us, gf / us * 1e3, maxerr, sumerr / n, nbad);
        };
        const unsigned g64 = (unsigned)(((T + 63) / 64) * H * B), g128 = (unsigned)(((T + 127) / 128) * H * B), g256 = (unsigned)(((T + 255) / 256) * H * B);
        run("attn_flow<4,2,2,1> (r5 default)", [&] { hipLaunchKernelGGL((attn_flow_kernel<4, 2, 2, 1>), dim3(g64), dim3(512), 0, 0, a); });
        run("attn_flow<4,2,2,2>", [&] { hipLaunchKernelGGL((attn_flow_kernel<4, 2, 2, 2>), dim3(g128), dim3(512), 0, 0, a); });
        run("attn_flow32<2,3>  64 q / wg", [&] { hipLaunchKernelGGL((attn_flow32_kernel<2, 3>), dim3(g64), dim3(128), 0, 0, a); });
        run("attn_flow32<4,3> 128 q / wg", [&] { hipLaunchKernelGGL((attn_flow32_kernel<4, 3>), dim3(g128), dim3(256), 0, 0, a); });
        run("attn_flow32<4,2> 128 q / wg", [&] { hipLaunchKernelGGL((attn_flow32_kernel<4, 2>), dim3(g128), dim3(256), 0, 0, a); });
        run("attn_flow32<8,4> 256 q / wg", [&] { hipLaunchKernelGGL((attn_flow32_kernel<8, 4>), dim3(g256), dim3(512), 0, 0, a); });
        run("attn_flow32<8,2> 256 q / wg", [&] { hipLaunchKernelGGL((attn_flow32_kernel<8, 2>), dim3(g256), dim3(512), 0, 0, a); });
        if (only >= 0 || (c.B == 16 && c.T == 674 && c.chunk == 0)) {
            run("abl 1 no softmax  <4,3>", [&] { hipLaunchKernelGGL((attn_flow32_kernel<4, 3, 1>), dim3(g128), dim3(256), 0, 0, a); });
            run("abl 2 no loop DMA <4,3>", [&] { hipLaunchKernelGGL((attn_flow32_kernel<4, 3, 2>), dim3(g128), dim3(256), 0, 0, a); });
            run("abl 6 no DMA, no barrier <4,3>", [&] { hipLaunchKernelGGL((attn_flow32_kernel<4, 3, 6>), dim3(g128), dim3(256), 0, 0, a); });
            run("abl 8 no LDS reads <4,3>", [&] { hipLaunchKernelGGL((attn_flow32_kernel<4, 3, 8>), dim3(g128), dim3(256), 0, 0, a); });
            run("abl 14 compute only <4,3>", [&] { hipLaunchKernelGGL((attn_flow32_kernel<4, 3, 14>), dim3(g128), dim3(256), 0, 0, a); });
            run("abl 15 MFMA only <4,3>", [&] { hipLaunchKernelGGL((attn_flow32_kernel<4, 3, 15>), dim3(g128), dim3(256), 0, 0, a); });
            run("abl 30 softmax only <4,3>", [&] { hipLaunchKernelGGL((attn_flow32_kernel<4, 3, 30>), dim3(g128), dim3(256), 0, 0, a); });
            run("abl 16 no MFMA <4,3>", [&] { hipLaunchKernelGGL((attn_flow32_kernel<4, 3, 16>), dim3(g128), dim3(256), 0, 0, a); });
        }
        if (only >= 0 || (c.B == 16 && c.T == 674 && c.chunk == 0) || (c.B == 2 && c.H == 8)) {
            // phase stamps of attn_flow32<4,3> (wall_clock64, 10 ns ticks): one launch alone, then the second of two back-to-back launches
            long long* ddbg; CK(hipMalloc(&ddbg, (size_t)g128 * 8 * 8));
            for (int pass = 0; pass < 2; ++pass) {
                CK(hipMemset(ddbg, 0, (size_t)g128 * 8 * 8)); CK(hipDeviceSynchronize());
                AttnFlowArgs a2 = a; a2.dbg = ddbg;
                if (pass) hipLaunchKernelGGL((attn_flow32_kernel<4, 3>), dim3(g128), dim3(256), 0, 0, a);
                hipLaunchKernelGGL((attn_flow32_kernel<4, 3>), dim3(g128), dim3(256), 0, 0, a2);
                CK(hipDeviceSynchronize());
                std::vector<long long> hd((size_t)g128 * 8); CK(hipMemcpy(hd.data(), ddbg, hd.size() * 8, hipMemcpyDeviceToHost));
                long long t0 = hd[0], tend = 0; double ph[4] = {0, 0, 0, 0}, startspread = 0; long long lateststart = 0;
                for (unsigned w = 0; w < g128; ++w) { t0 = std::min(t0, hd[w * 8]); tend = std::max(tend, hd[w * 8 + 4]); lateststart = std::max(lateststart, hd[w * 8]); }
                for (unsigned w = 0; w < g128; ++w) for (int k = 0; k < 4; ++k) ph[k] += (double)(hd[w * 8 + k + 1] - hd[w * 8 + k]) / g128;
                {
                    std::vector<double> ends, durs, loops;
                    for (unsigned w = 0; w < g128; ++w) { ends.push_back((hd[w * 8 + 4] - t0) * 0.01); durs.push_back((hd[w * 8 + 4] - hd[w * 8]) * 0.01); loops.push_back((hd[w * 8 + 3] - hd[w * 8 + 2]) * 0.01); }
                    std::sort(ends.begin(), ends.end()); std::sort(durs.begin(), durs.end()); std::sort(loops.begin(), loops.end());
                    auto q = [&](std::vector<double>& v, double f) { return v[(size_t)(f * (v.size() - 1))]; };
                    printf("    end time since first start, us: min %.2f  10%% %.2f  50%% %.2f  90%% %.2f  99%% %.2f  max %.2f;  lifetime: min %.2f 50%% %.2f 90%% %.2f max %.2f;  key loop: min %.2f 10%% %.2f 50%% %.2f 90%% %.2f max %.2f\n",
                           q(ends, 0), q(ends, .1), q(ends, .5), q(ends, .9), q(ends, .99), q(ends, 1), q(durs, 0), q(durs, .5), q(durs, .9), q(durs, 1), q(loops, 0), q(loops, .1), q(loops, .5), q(loops, .9), q(loops, 1));
                }
                if (pass) {
                    // where the time goes by placement: key loop by query block, by XCC, and by how many workgroups shared the CU
                    std::vector<double> byq(16, 0), nq(16, 0), byx(8, 0), nx(8, 0);
                    std::map<long long, std::vector<double>> bycu;
                    for (unsigned w = 0; w < g128; ++w) {
                        const double lp = (hd[w * 8 + 3] - hd[w * 8 + 2]) * 0.01; const int qbi = (int)hd[w * 8 + 7], xcc = (int)(hd[w * 8 + 6] & 7); const long long hw = hd[w * 8 + 5];
                        const int cu = (int)((hw >> 8) & 15), sh = (int)((hw >> 12) & 1), se = (int)((hw >> 13) & 7);
                        if (qbi < 16) { byq[qbi] += lp; nq[qbi] += 1; } byx[xcc] += lp; nx[xcc] += 1;
                        bycu[((long long)xcc << 16) | (se << 8) | (sh << 4) | cu].push_back(lp);
                    }
                    printf("    key loop by query block:"); for (int i = 0; i < 16; ++i) if (nq[i] > 0) printf(" %d: %.2f", i, byq[i] / nq[i]); printf("\n    key loop by XCC:");
                    for (int i = 0; i < 8; ++i) if (nx[i] > 0) printf(" %d: %.2f (%d)", i, byx[i] / nx[i], (int)nx[i]); printf("\n    CUs by resident workgroups:");
                    std::map<size_t, std::pair<int, double>> bycount; for (auto& kv : bycu) { double mx = 0; for (double v : kv.second) mx = std::max(mx, v); auto& e = bycount[kv.second.size()]; e.first++; e.second += mx; }
                    for (auto& kv : bycount) printf(" %zu wg: %d CUs, mean of the CU's slowest key loop %.2f us;", kv.first, kv.second.first, kv.second.second / kv.second.first); printf("  (%zu CUs used)\n", bycu.size());
                }
                printf("  stamps attn_flow32<4,3> %s: first start -> last end %.2f us; starts spread over %.2f us; mean per workgroup: setup %.2f, first tile lands %.2f, key loop %.2f, store %.2f us\n",
                       pass ? "behind another launch" : "alone", (tend - t0) * 0.01, (lateststart - t0) * 0.01, ph[0] * 0.01, ph[1] * 0.01, ph[2] * 0.01, ph[3] * 0.01);
            }
            CK(hipFree(ddbg));
        }
        CK(hipFree(dqk)); CK(hipFree(dvt)); CK(hipFree(dout)); CK(hipFree(dref)); CK(hipFree(dklen));
    }
    return 0;
}
