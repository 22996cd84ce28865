/* cosyvoice_amd — C ABI of the MI355X-native CosyVoice2 synthesis hot path.
 *
 * The reference (FunAudioLLM/CosyVoice) has no FFI: its plug points are Python attribute swaps on the
 * objects held by CosyVoice2Model.  This header is what a binding for those plug points
 * binds; each entry point names the reference interface it replaces.  Conventions:
 *   - every pointer marked "dev" is device memory owned by the caller (torch tensors on the Python side);
 *     nothing is retained past the call except tensors registered with cv_*_set_tensor (weights), which the
 *     caller must keep alive until cv_*_destroy;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all work is enqueued on it and the
 *     call returns without synchronising unless stated;
 *   - return value: 0 on success, non-zero on error; cv_last_error() gives the message (thread-local);
 *   - entry points are re-entrant across handles; one handle must not be used from two threads at once.
 */
#ifndef COSYVOICE_AMD_H
#define COSYVOICE_AMD_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct cv_llm cv_llm;
typedef struct cv_flow cv_flow;
typedef struct cv_hift cv_hift;

enum { CV_F32 = 0, CV_BF16 = 1, CV_I32 = 2 };
enum { CV_ACT_NONE = 0, CV_ACT_SILU = 1, CV_ACT_GELU_ERF = 2, CV_ACT_ELU = 3, CV_ACT_LEAKY = 4, CV_ACT_TANH = 5,
       CV_ACT_MISH = 6, CV_ACT_ABS = 7, CV_ACT_SNAKE = 8 };
enum { CV_MASK_NONE = 0, CV_MASK_CAUSAL = 1, CV_MASK_CHUNK = 2 };

const char* cv_last_error(void);
const char* cv_version(void);
/* 1 when the library was built by the CPU emulator used in tests (never shipped), 0 for the gfx950 build. */
int cv_is_emulated(void);

/* ------------------------------------------------------------------------------------------------------
 * Operator level (used by the stage entry points below; exported so each kernel can be parity-tested
 * in isolation against the oracle).
 * ---------------------------------------------------------------------------------------------------- */

/* Implicit GEMM: see cosyvoice_amd/csrc/gemm_conv.h for the exact index algebra.
 * Replaces torch.nn.Linear / Conv1d / ConvTranspose1d calls on the hot path, e.g.
 * cosyvoice/flow/decoder.py:36-62 (CausalConv1d), cosyvoice/hifigan/generator.py:46-122 (ResBlock convs). */
typedef struct cv_gemm_conv_args {
    const float* A; int64_t a_batch; int64_t a_len; int32_t lda; int32_t a_off0; int32_t tap_step; int32_t taps; int32_t K;
    int32_t pro; float pro_p; const float* pro_alpha;
    const void* W; int32_t w_dtype; int32_t Kp;
    const float* bias;
    float* C; int64_t c_batch; int64_t c_len; int32_t ldc; int64_t c_off;
    int32_t M; int32_t N; int32_t batch;
    int32_t act; float act_p;
    const float* res; int64_t res_batch;
    float out_scale;
    const float* row_scale; int64_t row_scale_batch;
    int32_t accumulate;
} cv_gemm_conv_args;
int cv_gemm_conv(const cv_gemm_conv_args* args, void* stream);

/* Row LayerNorm / RMSNorm over the last (channel) axis of a [rows, C] fp32 matrix.
 * y = act( (x-mean)*rstd*gamma + beta ) * scale * row_scale[row] + col_add[batch_of_row][c]
 * Replaces nn.LayerNorm (cosyvoice/transformer/encoder_layer.py:145-146, flow/decoder.py:71) and
 * Qwen2RMSNorm (transformers, pinned 4.51.3). gamma/beta/col_add may be NULL; rms != 0 selects RMSNorm. */
int cv_norm_rows(const float* x, float* y, int64_t rows, int32_t C, const float* gamma, const float* beta, float eps,
                 int32_t rms, int32_t act, float scale, const float* row_scale, const float* col_add,
                 int64_t rows_per_batch, void* stream);

/* Flash-style attention, head_dim = 64, fp32.
 * q/k/v are addressed as base + b*batch_stride + t*row_stride + h*head_stride (+ kv head = h / kv_group).
 * mask_mode: NONE (keys < Tk), CAUSAL (key <= q + Tk - Tq), CHUNK (key < (q/chunk+1)*chunk)
 *   (cosyvoice/utils/mask.py:127-158 subsequent_chunk_mask).
 * rel_bd: optional [B,H,Tq,2*Tq-1] matrix_bd of RelPositionMultiHeadedAttention BEFORE rel_shift; the kernel
 *   reads rel_bd[q][Tq-1-q+key] (cosyvoice/transformer/attention.py:222-244,318-326).
 * Replaces F.scaled_dot_product_attention (diffusers Attention inside matcha BasicTransformerBlock),
 * the conformer matmul-softmax (attention.py:108-124) and HF Qwen2 sdpa in prefill. */
typedef struct cv_attn_args {
    const float* q; int64_t q_batch; int32_t q_row; int32_t q_head;
    const float* k; int64_t k_batch; int32_t k_row; int32_t k_head;
    const float* v; int64_t v_batch; int32_t v_row; int32_t v_head;
    float* o; int64_t o_batch; int32_t o_row; int32_t o_head;
    int32_t B; int32_t H; int32_t kv_group; int32_t Tq; int32_t Tk;
    float scale; int32_t mask_mode; int32_t chunk;
    const float* rel_bd; int64_t bd_batch; int64_t bd_head; int32_t bd_row;
} cv_attn_args;
int cv_attention(const cv_attn_args* args, void* stream);

#ifdef __cplusplus
}
#endif
#endif
