// Dev tool: per-phase timeline of gemm_conv_kernel (bf16 MFMA variant) on one flow-estimator shape (M=1348, N=256, K=1024, bf16 W).
#include "../../cosyvoice_amd/csrc/gemm_conv.h"
#include <vector>
#include <cstdio>
using namespace cv;
int main() {
    const int M = 1348, N = 256, K = 1024;
    float *A, *C; unsigned short* W; long long* dbg;
    (void)hipMalloc(&A, (size_t)M * K * 4); (void)hipMalloc(&C, (size_t)M * N * 4); (void)hipMalloc(&W, (size_t)N * K * 2);
    const int nblk = 43 * 8;
    (void)hipMalloc(&dbg, (size_t)nblk * 64 * 8); (void)hipMemset(dbg, 0, (size_t)nblk * 64 * 8);
    (void)hipMemset(A, 0, (size_t)M * K * 4); (void)hipMemset(W, 0, (size_t)N * K * 2);
    GemmConvArgs a{};
    a.A = A; a.a_len = (long long)M * K; a.lda = K; a.taps = 1; a.K = K; a.a_vec = 1; a.W = W; a.Kp = K; a.C = C; a.c_len = (long long)M * N; a.ldc = N; a.c_vec = 1;
    a.M = M; a.N = N; a.out_scale = 1.f; a.a_bf16 = 1;
    for (int rep = 0; rep < 3; ++rep) {
        a.dbg = rep == 2 ? dbg : nullptr;
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((gemm_conv_kernel<32, 32, 256, true, true, 2, true>), dim3(43, 8, 1), dim3(256), 0, 0, a);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); printf("rep %d: %.1f us\n", rep, ms * 1e3);
    }
    std::vector<long long> h((size_t)nblk * 64);
    (void)hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost);
    long long t0 = h[0];
    for (int b = 0; b < nblk; ++b) if (h[(size_t)b * 64] && h[(size_t)b * 64] < t0) t0 = h[(size_t)b * 64];
    for (int b : {0, 1, 100, 200, 343}) {
        printf("block %3d start=%6lld :", b, h[(size_t)b * 64] - t0);
        for (int i = 1; i < 40 && h[(size_t)b * 64 + i]; ++i) printf(" %lld", h[(size_t)b * 64 + i] - h[(size_t)b * 64 + i - 1]);
        printf("\n");
    }
    long long last = 0; for (int b = 0; b < nblk; ++b) for (int i = 0; i < 64; ++i) if (h[(size_t)b * 64 + i] > last) last = h[(size_t)b * 64 + i];
    printf("first start -> last stamp: %lld cycles\n", last - t0);
    return 0;
}
