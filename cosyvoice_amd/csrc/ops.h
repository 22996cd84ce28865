// Host-callable launchers of the shared operators (used by the stage code in llm/flow/hift .hip files
// and by the operator-level C ABI in api_ops.hip).
#pragma once
#include "api_common.h"
#include "gemm_conv.h"
#include "norm.h"
#include "attention.h"

namespace cv {

// fills a_vec / c_vec from pointer+stride alignment, then launches
void gemm_conv(GemmConvArgs a, bool w_bf16, int batch, hipStream_t s);
void norm_rows(const NormArgs& a, hipStream_t s);
void attention(const AttnArgs& a, hipStream_t s);

// Convenience: plain Linear  C[M,N] = act(A[M,K] W^T + bias) (+res)
struct LinearW { const void* w = nullptr; const float* b = nullptr; int N = 0, K = 0, Kp = 0; bool bf16 = true; };
void linear(const float* A, int M, const LinearW& w, float* C, int act, const float* res, hipStream_t s,
            int lda = -1, int ldc = -1, bool accumulate = false, float out_scale = 1.f);

}  // namespace cv
