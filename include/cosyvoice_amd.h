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

enum { CV_F32 = 0, CV_BF16 = 1, CV_I32 = 2, CV_U8 = 3 };      /* CV_U8: OCP fp8 e4m3 bit patterns (the opt-in fp8 weights of the batched LLM decode) */
enum { CV_ACT_NONE = 0, CV_ACT_SILU = 1, CV_ACT_GELU_ERF = 2, CV_ACT_ELU = 3, CV_ACT_LEAKY = 4, CV_ACT_TANH = 5,
       CV_ACT_MISH = 6, CV_ACT_ABS = 7, CV_ACT_SNAKE = 8, CV_ACT_LOGCLAMP = 9 /* log(max(x, act_p)) */, CV_ACT_GELU_TANH = 10, CV_ACT_RELU = 11 };
enum { CV_MASK_NONE = 0, CV_MASK_CAUSAL = 1, CV_MASK_CHUNK = 2 };

const char* cv_last_error(void);
const char* cv_version(void);
/* 1 when the library was built by the CPU emulator used in tests (never shipped), 0 for the gfx950 build. */
int cv_is_emulated(void);
/* 1 when the library was built with -DCV_BUILD_EXPERIMENTS (cosyvoice_amd/build.py with CV_BUILD_EXPERIMENTS=1; always in the test-suite's emulator build): the measured
 * no-go variants (csrc/experiments/, the attention kernels of rounds 2-5, the decode-step fusions of round 3 ...) are then instantiated and their option switches
 * accepted; the default build leaves them out and a switch that selects one fails with an error. */
int cv_has_experiments(void);

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
    const void* W; int32_t w_dtype; int32_t Kp; int64_t ldw; int64_t w_batch;
    const float* bias;
    float* C; int64_t c_batch; int64_t c_len; int32_t ldc; int64_t c_off;
    int32_t M; int32_t N; int32_t batch;
    int32_t act; float act_p;
    const float* res; int64_t res_batch;
    float out_scale;
    const float* row_scale; int64_t row_scale_batch;
    int32_t accumulate;
    int32_t a_bf16;      /* 1: round the (prologue'd) activations to bf16 and multiply on the bf16 MFMA with fp32 accumulate (needs bf16 W,
                            16-byte aligned A layout; otherwise the fp32-accurate path runs); 0: fp32 accuracy - the fp32 MFMA chain for fp32 weights, the exact three-term bf16
                            split of the activations on the bf16 matrix pipe for bf16 weights */
    const void* W3;      /* optional, fp32 W only: the same weights as three bf16 planes, rows [3 N][taps * Kp], row 3 n + p = plane p of row n with
                            w = w1 + w2 + w3 exactly (cosyvoice_amd/weights.py::split3_planes).  When set (and the A layout is 16-byte aligned) the products run on
                            the bf16 matrix pipe with BOTH operands split, six exact plane products per k - fp32 accuracy at 2.5x the fp32 MFMA rate (HiFT). NULL: fp32 chain */
} cv_gemm_conv_args;
/* One-row calls (M == 1, batch 1, one tap, fp32 W, no prologue / scales) run as a GEMV over the matrix as registered in `W`: they ignore W3, and their sums differ
 * from the tile kernels' in summation order only (fp32 rounding) - the decode rows of a model agree with its prefill rows to that, not bit for bit. */
int cv_gemm_conv(const cv_gemm_conv_args* args, void* stream);
/* Process-wide switches of the operator layer (no reference counterpart: A/B and test knobs).  "gemv_f32" (default 1, initialised once from CV_GEMV_F32): a
 * cv_gemm_conv with ONE output row over fp32 weights runs on the GEMV kernel (1) or on the GEMM tile with one useful row (0); the two differ in summation order only. */
int cv_ops_set_option(const char* name, int32_t value);

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
    int32_t bf16;        /* 1: q, k, v and the probabilities rounded to bf16, both products on the bf16 MFMA (fp32 softmax and accumulate);
                            workgroup shape chosen from the problem size (2 / 3 force the 64- / 32-query variant) */
    const int32_t* klen; /* optional dev [B]: batch row b attends keys < min(Tk, klen[b]) (a padded batch of unequal lengths, the reference's
                            `mask` of flow/flow.py:236-281); null = every row attends Tk keys.  A row's result is bit-identical to the row alone. */
} cv_attn_args;
int cv_attention(const cv_attn_args* args, void* stream);


/* ------------------------------------------------------------------------------------------------------
 * Stage B2 — speech-token language model.  Replaces Qwen2LM.inference / inference_wrapper and
 * Qwen2Encoder.forward_one_step (cosyvoice/llm/llm.py:242-254, 458-549).  Weights are registered by name
 * (packed by cosyvoice_amd/weights.py: "layers.N.{ln1,wqkv,bqkv,wo,ln2,wgu,wdown}", "norm", "head.{w,b}",
 * "embed.speech"); bf16 matrices are [out][in] row-major, wgu interleaves (gate_j, up_j) rows.
 * ---------------------------------------------------------------------------------------------------- */
typedef struct cv_llm_config {
    int32_t hidden, layers, heads, kv_heads, inter, speech_vocab /* speech_token_size + 3 */, max_len /* KV capacity */;
    float rms_eps, rope_theta;
} cv_llm_config;

/* Sampling + loop bounds of Qwen2LM.inference_wrapper (llm/llm.py:535-549) and ras_sampling (utils/common.py:138-167). */
typedef struct cv_sampling {
    int32_t mode;            /* 0 greedy (argmax), 1 repetition-aware sampling */
    int32_t eos, n_stop;     /* stop ids = [eos, eos + n_stop); eos is suppressed while step < min_len */
    int32_t min_len, max_len;
    float top_p; int32_t top_k; int32_t win_size; float tau_r;
    uint64_t seed;
    int32_t use_uniforms;    /* 1: draw from the uniforms set by cv_llm_set_uniforms (parity tests) */
} cv_sampling;

int cv_llm_create(cv_llm** out, const cv_llm_config* cfg);
int cv_llm_set_tensor(cv_llm* m, const char* name, const void* dev_ptr, int32_t dtype, int64_t numel);
int cv_llm_finalize(cv_llm* m);
void cv_llm_destroy(cv_llm* m);
/* Options (all optional; defaults are the fastest measured forms): "use_graph" (decode hipGraph, 1), "attn_splits" (4 | 8 | 16 key slices per
 * head, 8), "batch_fp8" (0), "batch_packed" (batched decode on fragment-ordered weight copies, 1).  Measured alternatives kept for A/B runs, all default 0:
 * "fused_qkv_attn", "fused_attn_oproj" (+ "oproj_rblocks" 4 | 8, "oproj_waves" 8 | 16), "prefetch" (1 | 2, + "prefetch_shift"), "prefill_rows", "head_rows" (1 | 2),
 * "head_waves" (4 | 7). */
int cv_llm_set_option(cv_llm* m, const char* name, int32_t value);
/* lm_input: dev fp32 [L0, hidden] = [sos | text emb | task_id | prompt speech emb] (llm.py:494). Resets the KV cache. */
int cv_llm_prefill(cv_llm* m, const float* lm_input, int32_t L0, void* stream);
/* Qwen2LM.inference_bistream (llm/llm.py:551-661): forward `n_rows` more input rows on top of the cached positions - the reference's
 * `forward_one_step(lm_input, cache=cache)` with a multi-row lm_input - keeping the running request (emitted tokens, step counter) and
 * clearing a `done` left by a fill token; the next decode samples from the hidden state of the last appended row. */
int cv_llm_prefill_append(cv_llm* m, const float* rows, int32_t n_rows, void* stream);
/* appends a token the HOST decided on (a forced or sampled fill token) to the decoded-token history the repetition-aware sampler looks
 * at (`out_tokens` of the reference loop), without running the backbone */
int cv_llm_push_token(cv_llm* m, int32_t token, void* stream);
/* the special id that ended the last cv_llm_decode (eos, fill, ...), -1 if it did not end on one */
int cv_llm_last_stop_token(cv_llm* m);
int cv_llm_set_uniforms(cv_llm* m, const float* host_uniforms, int32_t n, void* stream);
/* Runs up to n_steps iterations of the decode loop on the device, then synchronises and returns the tokens emitted by
 * this call (host ints, like the reference's `yield top_ids`).  *finished != 0 when a stop id was sampled or max_len hit. */
int cv_llm_decode(cv_llm* m, int32_t n_steps, const cv_sampling* sp, int32_t* out_tokens, int32_t* n_out, int32_t* finished, void* stream);
/* one eager decode step with a HIP-event pair around every launch: per category (0 qkv, 1 attention, 2 o_proj, 3 gate_up, 4 down,
 * 5 head, 6 sample) launch count and summed duration in ms — used by bench.py for the live roofline figure */
int cv_llm_profile_step(cv_llm* m, const cv_sampling* sp, int32_t* counts8, float* ms8, void* stream);
/* one kernel class (category as above, 0..5) as a dependent chain of its real launches (one per layer) in a hipGraph, `reps` replays
 * between ONE event pair: total duration and number of launches — the per-launch duration bench.py prices against the HBM roofline */
int cv_llm_profile_chain(cv_llm* m, int32_t category, int32_t reps, float* total_ms, int32_t* launches, void* stream);
/* Lock-step batched decode (BASELINE.json configs[2]/[3]; the reference batches through vLLM, cli/model.py:281-290): up to 32 sequences
 * (16 on the fp8 path) advance one token per step and every weight matrix is streamed once per step for all of them (skinny GEMMs on the matrix pipe at fp32 accuracy: exact three-term bf16 split of the activations).
 * cv_llm_batch_begin sizes the slots, cv_llm_batch_prefill runs the normal prefill for one request and parks its KV prefix / state /
 * sampling parameters in `slot`; cv_llm_batch_prefill_many fills n slots with ONE prefill pass over the row-stacked prompts (rows: dev
 * [sum L0s][hidden], slot j's rows after slot j-1's; sps: n sampling structs) - the GEMMs see M = sum of the prompt lengths;
 * cv_llm_batch_decode runs n_steps steps and returns, per slot, the tokens emitted by this call (out_tokens[slot * n_steps + k]),
 * their count and whether the slot has finished.  A sequence decoded in a batch gives the same tokens as decoded alone. */
int cv_llm_batch_begin(cv_llm* m, int32_t nb, void* stream);
int cv_llm_batch_prefill(cv_llm* m, int32_t slot, const float* lm_input, int32_t L0, const cv_sampling* sp, void* stream);
int cv_llm_batch_prefill_many(cv_llm* m, int32_t n, const int32_t* slots, const float* rows, const int32_t* L0s, const cv_sampling* sps, void* stream);
int cv_llm_batch_decode(cv_llm* m, int32_t n_steps, int32_t* out_tokens, int32_t* n_out, int32_t* finished, void* stream);
/* Opt-in fp8 batched decode (BASELINE.json configs[4] "fp8 MFMA LLM path"; cv_llm_set_option(m, "batch_fp8", 1) after registering "<matrix>.f8"
 * (CV_U8: OCP e4m3 bit patterns, same row-major layout as the bf16 matrix) and "<matrix>.f8s" (fp32 scale per row) for wqkv / wo / wgu / wdown of
 * every layer and head.w).  cv_skinny_fp8 is the test hook of its one kernel: y[b][n] = epi(sum_k W8[n][k] * q(x[b][k] * gamma[k])) with the
 * activations quantised per sequence inside the kernel; mode 0: + bias + res, 1: silu(gate) * up over interleaved rows, 2: split-K partials
 * [ksplit][nb][N]; rt = row tiles (16 rows) per workgroup. */
int cv_skinny_fp8(const void* w8, const float* wscale, const float* bias, const float* x, int64_t ldx, float* y, int64_t ldy, int32_t N, int32_t K,
                  const float* gamma, float eps, const float* res, int64_t ldres, int32_t mode, int32_t nb, int32_t ksplit, int32_t rt, void* stream);
int cv_llm_last_logits(cv_llm* m, float* host_out, void* stream);
int cv_llm_batch_logits(cv_llm* m, int32_t slot, float* host_out, void* stream);      /* test hook: one slot's logits after the last batched step */
int cv_llm_last_hidden(cv_llm* m, float* host_out, void* stream);
/* out[r][:] = table[ids[r]][:] * scale  (nn.Embedding lookups that build lm_input / flow token embeddings) */
int cv_gather_rows(const void* table, int32_t dtype, int64_t table_rows, int32_t dim, const int32_t* ids_dev, int32_t n, float* out, float scale, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Stages B3/B4/B5 — token -> mel flow-matching decoder.  Weight names: cosyvoice_amd/weights.py:pack_flow.
 * ---------------------------------------------------------------------------------------------------- */
typedef struct cv_flow_config {
    int32_t vocab, dim, enc_heads, ffn, enc_blocks, up_blocks, spk_dim, mel, est_ch, est_heads, est_blocks, est_mid,
            pre_lookahead, chunk /* static_chunk_size in tokens; the estimator uses 2*chunk frames */;
    float cfg_rate;
    int32_t estimator;   /* 0: CausalConditionalDecoder U-Net (CosyVoice2, flow/decoder.py:294-494); 1: DiT (Fun-CosyVoice3: CausalMaskedDiffWithDiT.inference
                          * flow/flow.py:369-414, DiT.forward flow/DiT/dit.py:145-176).  For 1: dim = input_size = 80, ffn = PreLookaheadLayer channels,
                          * est_ch = DiT width, est_blocks = depth, est_mid = ff_mult, enc_blocks = up_blocks = 0.
                          * 2: ConditionalDecoder, the U-Net of CosyVoice-300M (flow/decoder.py:88-291) ALONE - the handle then serves cv_flow_estimator and
                          * cv_flow_solve only (that model's encoder / length regulator / flow cache are the caller's, cosyvoice_amd/cosyvoice1_hip.py); mel, est_ch
                          * (one width for every level), est_heads, est_blocks, est_mid and cfg_rate are read, the number of down / up stages is taken from the tensors. */
} cv_flow_config;
int cv_flow_create(cv_flow** out, const cv_flow_config* cfg);
int cv_flow_set_tensor(cv_flow* m, const char* name, const void* dev_ptr, int32_t dtype, int64_t numel);
int cv_flow_finalize(cv_flow* m);
/* Options: "use_graph" (Euler-solve hipGraph cache: up to 32 shapes per handle, least recently used evicted; default on), "bf16_mfma" (precision mode), "fused" (bf16
 * mode: fused transformer blocks, 1), "flow_tile" / "flow_ntile" / "attn_waves" / "attn_kt" / "attn_ks" (tile choices, 0 = by size).  Measured alternatives, default off:
 * "fused_tail" (+ "tail_ring" 8 | 16), "est_streams" (1 | 2); "graph_cap" (1 .. 256 cached shapes).  Round 5: "big_rows" (2000: passes of at least this many estimator rows run
 * on the large-M kernel set, csrc/flow_big.h), "fused_band" (1: in such passes everything between a block's attention and the next QKV GEMM is one launch per row band,
 * csrc/flow_band.h), "band_qkv" (1: that launch also runs the NEXT block's QKV GEMM - a block of a large pass is two launches, attention and band), "band_bm" (0: rows
 * per band by the row count of the pass - 32 / 48 / 64, csrc/flow.hip::band_rows_for; 32 | 48 | 64 forces), "band_pipe" (2: the feed-forward chunks of a 48- / 32-row band
 * as a software pipeline - MFMA slices between the GELU pieces; 1: 48-row bands only, 0: never), "eager_streams" (1; 2: the batch rows of a pass that runs
 * eager as two launch chains on two streams - faster in isolation, slower next to the model's token2wav lanes, profiles/r5_band_qkv.txt).  Every combination is
 * bit-identical per utterance.  Round 6: "ln_qkv" (1), "attn32_waves" (0 = 4 | 2 | 4: 128- or 64-query workgroups of the flash attention), "res_tile" (1: the
 * small-pass residual GEMMs on 32 x 32 tiles, 0: 32 x 64). */
int cv_flow_set_option(cv_flow* m, const char* name, int32_t value);
/* "graph_captures" (Euler-solve graphs captured so far), "graphs_cached" (held now; option "graph_cap", default 32) - test / monitoring hook, no reference counterpart */
int cv_flow_get_stat(cv_flow* m, const char* name, int64_t* value);
void cv_flow_destroy(cv_flow* m);
/* measurement hook (bench.py `roofline_mfma`; no reference counterpart): the QKV GEMM, the flash attention and the 64-row band launch (csrc/flow_band.h) of one transformer
 * block of a LARGE pass - nz estimator batch rows (2 per utterance) of T frames - each timed as `reps` back-to-back launches between one HIP-event pair on `stream`;
 * us3[0..2] = microseconds per launch.  bf16 mode, U-Net estimator only. */
int cv_flow_profile_block(cv_flow* m, int32_t nz, int32_t T, int32_t reps, float* us3, void* stream);
/* B4: flow.encoder(token_emb[1,n,dim], token_len, context=[1,3,dim] or empty, streaming) -> h[1,2n,dim]
 * (cosyvoice/flow/flow.py:258-261, transformer/upsample_encoder.py:244-307).  tok_emb / context / h_out: dev fp32, row-major. */
int cv_flow_encoder(cv_flow* m, const float* tok_emb, int32_t n_tok, const float* context, int32_t streaming, float* h_out, void* stream);
/* B3: flow.decoder.estimator(x[2,80,T], mask[2,1,T], mu[2,80,T], t[2], spks[2,80], cond[2,80,T], streaming) -> [2,80,T]
 * (flow/flow_matching.py:126-128 nn.Module branch, flow/decoder.py:405-494); all dev fp32 contiguous, reference layouts.
 * `out` may alias `x` (x is read by the first launch only): that is what the reference's other branch hands over - raw addresses with the output on x
 * (flow_matching.py:129-153, the TensorRT-shaped form; host side: cosyvoice_amd.flow.EstimatorEngine).
 * mask must be all ones (batch-1 inference, flow.py:270); it is applied to the output like the reference's `output * mask`. */
int cv_flow_estimator(cv_flow* m, const float* x, const float* mask, const float* mu, const float* t, const float* spks, const float* cond,
                      int32_t T, int32_t streaming, float* out, void* stream);
/* The same with a PADDED mask (the reference's estimator takes one, flow/decoder.py:405-494: every block multiplies by it, the attention bias is built
 * from it): mask row b = key_len[b] ones followed by zeros (HOST int32 [2], 1..T).  A row's valid positions come out as they would for that row alone
 * at its own length - the convolutions are causal, norms / linears per position, and the attention takes key_len[b] as the row's key count - and the
 * padded positions are zero (`output * mask`).  key_len == NULL: cv_flow_estimator. */
int cv_flow_estimator_masked(cv_flow* m, const float* x, const float* mask, const int32_t* key_len, const float* mu, const float* t, const float* spks,
                             const float* cond, int32_t T, int32_t streaming, float* out, void* stream);
/* ConditionalCFM.solve_euler for ONE utterance (flow/flow_matching.py:71-124; cosine t schedule, classifier-free guidance at cfg.cfg_rate, the estimator the handle
 * was built with): x [T][mel] dev fp32 CHANNEL-LAST - in: the initial noise z (with whatever the caller's flow cache wrote over its head), out: the mel; mu, cond
 * [T][mel] channel-last, spks [mel].  The n_timesteps estimator evaluations are one hipGraph from the second call of a shape on (option "use_graph"). */
int cv_flow_solve(cv_flow* m, float* x, const float* mu, const float* spks, const float* cond, int32_t T, int32_t n_timesteps, void* stream);
/* B5: flow.inference(token ++ prompt_token, prompt_feat, embedding, streaming, finalize) -> mel[1,80,mel_len2]
 * (flow/flow.py:235-281 + flow_matching.py:71-124,203-227).  token_ids: dev int32 [n_tok] = prompt tokens then new tokens;
 * prompt_feat: dev [mel_len1,80]; embedding: dev [spk_dim]; noise_cl: dev [>=T,80] = CausalConditionalCFM.rand_noise
 * transposed to channel-last; mel_out: dev [80, T - mel_len1] (channel-first like the reference). */
int cv_flow_inference(cv_flow* m, const int32_t* token_ids, int32_t n_tok, const float* prompt_feat, int32_t mel_len1, const float* embedding,
                      const float* noise_cl, int32_t streaming, int32_t finalize, int32_t n_timesteps, float* mel_out, int32_t* mel_len2_out, void* stream);
/* The same for n_utt (1..8) utterances of EQUAL shape (token count, prompt frames) in one pass: token_ids dev [n_utt][n_tok], prompt_feat dev
 * [n_utt][mel_len1][80], embedding dev [n_utt][spk_dim], mel_out dev [n_utt][80][T - mel_len1].  The encoder runs per utterance, the CFM Euler solve
 * once over all of them (estimator batch rows = 2 x n_utt); each utterance's result is what cv_flow_inference gives for it alone
 * (the reference's own contract for batched flow, flow/flow.py:246).  CausalConditionalDecoder estimator only. */
/* The same for utterances of DIFFERENT lengths (the reference's padded batch with `mask`, flow/flow.py:236-281): token_ids dev, the utterances'
 * ids one after the other; n_tok / mel_len1 HOST arrays [n_utt]; prompt_feat dev, the prompts' [mel_len1[u]][80] blocks one after the other;
 * mel_out dev, the results' [80][mel_len2[u]] blocks one after the other; mel_len2_out HOST [n_utt].  Each result is bit-identical to the
 * utterance alone: convolutions are causal, attention takes a key count per batch row (cv_attn_args.klen). */
int cv_flow_inference_ragged(cv_flow* m, int32_t n_utt, const int32_t* token_ids, const int32_t* n_tok, const float* prompt_feat, const int32_t* mel_len1,
                             const float* embedding, const float* noise_cl, int32_t streaming, int32_t finalize, int32_t n_timesteps, float* mel_out,
                             int32_t* mel_len2_out, void* stream);
int cv_flow_inference_batch(cv_flow* m, int32_t n_utt, const int32_t* token_ids, int32_t n_tok, const float* prompt_feat, int32_t mel_len1, const float* embedding,
                            const float* noise_cl, int32_t streaming, int32_t finalize, int32_t n_timesteps, float* mel_out, int32_t* mel_len2_out, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Stage B6 — HiFT vocoder.  Weight names: cosyvoice_amd/weights.py:pack_hift (weight-norm folded, fp32).
 * ---------------------------------------------------------------------------------------------------- */
typedef struct cv_hift_config {
    int32_t mel, base, harmonics, sr;
    int32_t n_ups, ups[4], up_k[4];
    int32_t n_res, res_k[4], src_k[4];
    int32_t n_dil, dil[4];
    int32_t n_fft, hop, f0_ch;
    float nsf_alpha, nsf_sigma, voiced_thr, lrelu, audio_limit;
    int32_t causal;        /* 1: CausalHiFTGenerator / CausalConvRNNF0Predictor (Fun-CosyVoice3; hifigan/generator.py:572-726, f0_predictor.py:62-103) */
    int32_t look_right;    /* conv_pre_look_right (causal only; cosyvoice3.yaml: 4) */
} cv_hift_config;
int cv_hift_create(cv_hift** out, const cv_hift_config* cfg);
int cv_hift_set_tensor(cv_hift* m, const char* name, const void* dev_ptr, int32_t dtype, int64_t numel);
int cv_hift_finalize(cv_hift* m);
/* Options: "f0_float64" (0 | 1, default 0): the f0 predictor - five convolutions + classifier - with every sum in double and f0 rounded to fp32 at the end, the mode
 * the reference runs CausalConvRNNF0Predictor in (hifigan/generator.py:716-717: module and input converted to float64 on every call).  Works for both generators.
 * "terms" (6 | 3, default 6): plane products per k of the decoder's convolutions (both generators) - 6 = the fp32-exact class, 3 = 16 significand bits per factor with
 * fp32 accumulation: the reduced-precision mode CosyVoice3Model(fp16=True) selects where the reference runs its vocoder under autocast (cli/model.py:426-447); the f0
 * predictor and the source are not affected. */
int cv_hift_set_option(cv_hift* m, const char* name, int32_t value);
void cv_hift_destroy(cv_hift* m);
/* ConvRNNF0Predictor.forward (hifigan/f0_predictor.py:56-59): speech_feat dev [80, frames] -> f0 dev [frames] */
int cv_hift_f0(cv_hift* m, const float* speech_feat, int32_t frames, float* f0_out, void* stream);
/* HiFTGenerator.decode(x, s) (hifigan/generator.py:507-539): speech_feat dev [80, frames], source dev [480*frames] -> speech dev [480*frames] */
int cv_hift_decode(cv_hift* m, const float* speech_feat, int32_t frames, const float* source, float* speech_out, void* stream);
/* B6: HiFTGenerator.inference(speech_feat[1,80,m], cache_source[1,1,c]) -> (speech[1,480m], source[1,1,480m]) (generator.py:557-569).
 * noise: optional dev [480m, 9] N(0,1) variates for SineGen2 (parity tests); NULL -> in-kernel counter RNG keyed by `seed`. */
int cv_hift_inference(cv_hift* m, const float* speech_feat, int32_t frames, const float* cache_source, int32_t cache_len,
                      const float* noise, uint64_t seed, float* speech_out, float* source_out, void* stream);
/* The same for n_utt (1..16) utterances of EQUAL length in ONE launch sequence - the batched vocoding of the reference's high-throughput runtime
 * (runtime/triton_trtllm/token2wav.py:20-22 batches token2wav): speech_feat dev [n_utt][80][frames], noise dev [n_utt][480 frames][9] or NULL, seeds HOST
 * [n_utt] (one counter-RNG key per utterance), speech_out / source_out dev [n_utt][480 frames].  Every utterance's result is bit-identical to
 * cv_hift_inference of it alone with its key (convolutions pad per utterance, the phase walk and the iSTFT run per utterance).  No cache_source: one-shot requests. */
int cv_hift_inference_batch(cv_hift* m, int32_t n_utt, const float* speech_feat, int32_t frames, const float* noise, const uint64_t* seeds, float* speech_out,
                            float* source_out, void* stream);
/* Fun-CosyVoice3 (handle created with causal = 1).  `finalize` = 0: a streaming chunk whose trailing frames are look-ahead context only.
 * cv_hift_causal_f0: CausalConvRNNF0Predictor.forward(x, finalize) -> f0 [frames] or [frames - 3] (f0_predictor.py:94-103; fp32 here, float64 in
 * the reference generator.py:716-717).  cv_hift_causal_decode: CausalHiFTGenerator.decode(x, s, finalize) (generator.py:684-711): speech_feat dev
 * [80, frames], source dev [480 frames] -> 480 frames samples, or 480 (frames - look_right - 1) when not final.  cv_hift_causal_inference:
 * CausalHiFTGenerator.inference(speech_feat, finalize) (generator.py:713-726) -> speech (480 frames | 480 (frames - 8)) and source (480 frames |
 * 480 (frames - 3)); noise: optional dev [>= n_source, 9] uniform [0, 1) variates standing in for the model's fixed SineGen2 buffer, NULL -> counter RNG. */
int cv_hift_causal_f0(cv_hift* m, const float* speech_feat, int32_t frames, int32_t finalize, float* f0_out, int32_t* n_out, void* stream);
int cv_hift_causal_decode(cv_hift* m, const float* speech_feat, int32_t frames, const float* source, int32_t finalize, float* speech_out, int64_t* n_out,
                          void* stream);
int cv_hift_causal_inference(cv_hift* m, const float* speech_feat, int32_t frames, int32_t finalize, const float* noise, uint64_t seed,
                             float* speech_out, int64_t* n_speech, float* source_out, int64_t* n_source, void* stream);


/* ------------------------------------------------------------------------------------------------------
 * Glue of CosyVoice2Model.token2wav (boundary B1, cosyvoice/cli/model.py:292-326)
 * ---------------------------------------------------------------------------------------------------- */
/* fade_in_out (cosyvoice/utils/common.py:170-178) in place on the device: fade_in[:n] = fade_in[:n]*window[:n] + fade_out_tail[:n]*window[n:2n];
 * fade_out_tail points at the LAST n samples of the cached speech; window: dev fp32 [2n] (np.hamming). */
int cv_fade_in_out(float* fade_in, const float* fade_out_tail, const float* window, int32_t overlap, void* stream);
/* F.interpolate(x[1,C,T], size=Tn, mode='linear') for the `speed` argument (cli/model.py:322), channel-first. */
int cv_interp_linear(const float* x, float* y, int32_t C, int32_t T, int32_t Tn, void* stream);

/* Prompt-mel front end: replaces `matcha.utils.audio.mel_spectrogram(y, n_fft, num_mels, sampling_rate, hop_size, win_size,
 * fmin, fmax, center=False)` as `CosyVoiceFrontEnd._extract_speech_feat` calls it (cosyvoice/cli/frontend.py:120-125, cosyvoice2.yaml:150-158).
 * cv_reflect_pad: y[L + 2 pad] = F.pad(x[L], (pad, pad), mode="reflect").  cv_stft_magnitude: spec [T][2 bins] (re | im, produced by cv_gemm_conv
 * over the padded signal with lda = hop and the windowed DFT basis as W) -> mag [T][ldm] = sqrt(re^2 + im^2 + eps), pad columns zeroed.  The mel
 * projection + log(clamp(., 1e-5)) is one more cv_gemm_conv with act = CV_ACT_LOGCLAMP.  Host side: cosyvoice_amd/frontend.py. */
int cv_reflect_pad(const float* x, float* y, int32_t L, int32_t pad, void* stream);
int cv_stft_magnitude(const float* spec, float* mag, int32_t T, int32_t bins, int32_t ldm, float eps, void* stream);

/* Feature front ends of the two ONNX extractors (cosyvoice/cli/frontend.py:95-118).  The networks themselves
 * (speech_tokenizer_v2.onnx, campplus.onnx) stay onnxruntime sessions of the caller; what runs here is what the reference computes on the CPU in
 * front of them: `whisper.log_mel_spectrogram(speech, n_mels=128)` (frontend.py:98) and `kaldi.fbank(speech, num_mel_bins=80, dither=0,
 * sample_frequency=16000)` minus its mean over frames (frontend.py:109-113).  Both are cv_gemm_conv over the (padded) signal against a DFT basis
 * (for the fbank: with DC removal, pre-emphasis and the Povey window folded into it), cv_stft_power, and a second cv_gemm_conv against the mel bank
 * with act = CV_ACT_LOGCLAMP; then
 * cv_whisper_lognorm: lnmel [T][n_mels] = ln(max(mel, 1e-10)) -> out [n_mels][T] = (max(log10 mel, max over everything - 8) + 4) / 4;
 * cv_sub_col_mean:    x [T][C] -= its mean over T (in place).  Host side: cosyvoice_amd/frontend.py (WhisperLogMel, KaldiFbank). */
int cv_stft_power(const float* spec, float* pw, int32_t T, int32_t bins, int32_t ldm, void* stream);
int cv_whisper_lognorm(const float* lnmel, float* out, int32_t T, int32_t n_mels, void* stream);
int cv_sub_col_mean(float* x, int32_t T, int32_t C, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * SURVEY.md section 8 row f4 - CosyVoice-300M (first generation) on the kernels.  Its three stages are sequenced by the host side
 * (cosyvoice_amd/cosyvoice1_hip.py: TransformerLM.inference llm/llm.py:162-223, MaskedDiffWithXvec.inference flow/flow.py:102-146,
 * HiFTGenerator.inference at 22.05 kHz hifigan/generator.py:557-569) over the operator-level entry points above (cv_gemm_conv, cv_norm_rows,
 * cv_attention, cv_gather_rows, cv_hift_f0, cv_hift_decode) plus the operators below.  Activations: dev fp32, channel-last [batch][time][channel].
 * ---------------------------------------------------------------------------------------------------- */
/* torch.nn.GroupNorm(G, C) over a [B, C, T] tensor held channel-last, with the activation and a per-(batch, channel) addend fused:
 * y = act((x - mean_bg) * rstd_bg * gamma[c] + beta[c]) + col_add[b * col_add_batch + c]   (Matcha Block1D: Conv1d -> GroupNorm(8) -> Mish, and the
 * ResnetBlock1D's `+ mlp(t_emb)` - flow/decoder.py:88-205 via matcha.models.components.decoder; InterpolateRegulator's GroupNorm(1), length_regulator.py:36-44).
 * Statistics are accumulated in double.  workspace: dev, B * G * 64 doubles.  gamma / beta / col_add may be NULL. */
int cv_group_norm(const float* x, float* y, int32_t B, int32_t T, int32_t C, int32_t G, const float* gamma, const float* beta, float eps, int32_t act,
                  const float* col_add, int64_t col_add_batch, double* workspace, void* stream);
/* SourceModuleHnNSF over SineGen (the 22.05 kHz generator's harmonic source, hifigan/generator.py:125-186, 318-375): f0 dev [frames] (one value per mel
 * frame; the reference's nearest up-sampling by `scale` is implicit) -> source_out dev [frames * scale] = tanh(l_linear(sine waves)).  phase0: dev
 * [harmonics + 1] initial phases (the reference draws Uniform(-pi, pi), first entry 0); noise: dev [harmonics + 1][frames * scale] N(0, 1) variates
 * (parity hook) or NULL -> in-kernel counter RNG keyed by `seed`.  The running phase is the exact per-frame closed form of the reference's fp32 cumsum.
 * workspace: dev, frames * (harmonics + 1) * 12 bytes, 8-byte aligned. */
int cv_sinegen1_source(const float* f0, int32_t frames, int32_t scale, int32_t harmonics, float sr, const float* phase0, const float* noise, uint64_t seed,
                       const float* lin_w, const float* lin_b, float amp, float sigma, float thr, float* source_out, void* workspace, void* stream);
/* One Euler step of ConditionalCFM.solve_euler with classifier-free guidance (flow/flow_matching.py:94-123): x[i] += dt * ((1 + rate) * d[i] - rate * d[n + i]). */
int cv_cfg_euler(float* x, const float* d, int64_t n, float dt, float rate, void* stream);
/* The estimator's input for the guidance pair (flow_matching.py:101-108 + decoder.py:225-231): x, mu, cond dev [T][mel], spks dev [mel] ->
 * h dev [2][T][4 mel] = (x | mu | spks | cond) for row 0 and (x | 0 | 0 | 0) for row 1. */
int cv_pack_cfg_input(const float* x, const float* mu, const float* spks, const float* cond, float* h, int32_t T, int32_t mel, void* stream);
/* matcha SinusoidalPosEmb(dim)(t, scale = 1000): t dev [n] -> out dev [n][dim] = (sin | cos). */
int cv_time_sinusoid(const float* t, float* out, int32_t n, int32_t dim, void* stream);
/* out[b][t] = (a[b][t][0:ca] | bb[b][t][0:cb]) for t < T; a_batch / b_batch: floats between the batch rows of a / bb (their own lengths may exceed T:
 * the U-Net cuts the up-sampled stream to the skip connection's length, flow/decoder.py:275). */
int cv_concat_cols(const float* a, int32_t ca, int64_t a_batch, const float* b, int32_t cb, int64_t b_batch, float* out, int32_t T, int32_t B, void* stream);
/* F.interpolate(mode="linear") over time for channel-last rows: x dev [T][C] -> y rows 0 .. Tn - 1 of pitch ldy (length_regulator.py:52-70). */
int cv_interp_rows(const float* x, float* y, int32_t C, int32_t T, int32_t Tn, int32_t ldy, void* stream);
/* out [cols][rows] = in [rows][cols] transposed (channel-last <-> channel-first at the API boundary). */
int cv_transpose(const float* in, float* out, int32_t rows, int32_t cols, void* stream);

/* The decode step of CosyVoice-300M's TransformerLM as ONE call (round 4): replaces, per generated token, `self.llm.forward_chunk(lm_input, ..., att_cache, cnn_cache)`
 * + `self.llm_decoder(y_pred[:, -1])` of TransformerLM.inference (cosyvoice/llm/llm.py:206-212; transformer/encoder.py:267-327 forward_chunk,
 * encoder_layer.py:60-119, attention.py:200-330 RelPositionMultiHeadedAttention with its key / value cache).  Prefill (the first forward_chunk over the whole
 * prompt) stays on cv_gemm_conv / cv_norm_rows / cv_attention and fills the same cache buffers; the sampler (ras_sampling, cosyvoice/utils/common.py:109-131)
 * stays with the caller, as in the reference.  All weights: dev fp32, row pitch = K rounded up to 32 floats, zero padded (cv_gemm_conv's fp32 layout), owned by the caller
 * and kept alive as long as the handle.
 *   w_qkv / b_qkv : [4 d][d] rows (linear_q | linear_q | linear_k | linear_v), bias (b_q + pos_bias_u | b_q + pos_bias_v | b_k | b_v) - the layer's cache row is
 *                   (q + u | q + v | k | v), what attention.py:300-318 forms from q;
 *   cache rows    : per layer dev [cap][4 d] (cv_lm1_bind), rows 0 .. pos - 1 filled by the prefill and the earlier steps;
 *   tabs          : per layer dev [2 n_tab - 1][d] = linear_pos(pos_emb), row m = relative position n_tab - 1 - m (embedding.py:143-160), n_tab >= cap.
 * cv_lm1_step(x_row dev [d_in], pos) : input layer Linear -> LayerNorm(1e-5) -> act -> * xscale, n_layers pre-norm layers writing cache row `pos` and attending to
 * rows 0 .. pos, after_norm, decoder -> logits dev [n_out].  73 launches for 14 layers; with option "graph" all but the first are one hipGraph captured once per handle:
 * position, cache and table addresses are read on the device from a block the first kernel / cv_lm1_bind update.  Results: every GEMV and LayerNorm has the bits of
 * the cv_gemm_conv (M = 1) / cv_norm_rows launches it replaces; the one-query attention sums exact fp32 products in another order (fp32 rounding). */
/* A handle serves one step at a time (not thread-safe; the host side serialises requests per LM object); bind and step on the SAME stream. */
typedef struct cv_lm1 cv_lm1;
typedef struct cv_lm1_layer_weights {
    const float *ln1_g, *ln1_b, *w_qkv, *b_qkv, *w_out, *b_out, *ln2_g, *ln2_b, *w1, *b1, *w2, *b2;
} cv_lm1_layer_weights;
typedef struct cv_lm1_config {
    int32_t n_layers, d, heads, ffn, d_in, n_out;
    int32_t act;                 /* CV_ACT_RELU (TransformerEncoder of CosyVoice-300M): the input layer's and the feed-forward activation */
    float xscale;                /* sqrt(d): the positional encoding's input scale (embedding.py:74) */
    const float *embed_w, *embed_b, *embed_g, *embed_beta, *after_g, *after_b, *dec_w, *dec_b;
} cv_lm1_config;
cv_lm1* cv_lm1_create(const cv_lm1_config* cfg, const cv_lm1_layer_weights* layers /* [n_layers] */);      /* NULL on error (cv_last_error) */
void cv_lm1_destroy(cv_lm1* m);
/* The model's fp16 mode (the reference: cli/cosyvoice.py:27-56 `fp16=True` -> cli/model.py:60-63 `self.llm.half()`), here W16A32: the decode step reads the SAME matrices
 * as bf16 [N][Kp] (row pitch Kp = round_up(K, 32) elements, like the fp32 form) - half the bytes per token; activations, biases, norms, cache and logits stay fp32 and
 * every product is the fp32 product of the bf16 weight, in the fp32 kernels' order (bit-identical to the fp32 step over bf16-ROUNDED fp32 matrices).  embed_w / dec_w may
 * be NULL (that product then stays on the fp32 matrix).  layers == NULL switches back to fp32.  The pointers must outlive the handle. */
typedef struct cv_lm1_layer_bf16 { const void *w_qkv, *w_out, *w1, *w2; } cv_lm1_layer_bf16;
int cv_lm1_use_bf16(cv_lm1* m, const cv_lm1_layer_bf16* layers /* [n_layers] */, const void* embed_w, const void* dec_w);
/* A request's buffers: rows / tabs are HOST arrays of n_layers device pointers.  Call again whenever a buffer is reallocated (a grown cache, grown tables). */
int cv_lm1_bind(cv_lm1* m, float* const* rows, const float* const* tabs, int32_t n_tab, int32_t cap, void* stream);
int cv_lm1_step(cv_lm1* m, const float* x_row, int32_t pos, float* logits, void* stream);
/* The decode LOOP on the device (round 5; replaces the python loop of TransformerLM.inference, llm/llm.py:196-223, which takes the logits to the host, samples there and
 * sends the embedding row of the sampled token back once per token).  cv_lm1_decode_begin: the first input row (dev [d_in]), the position it will be written at, the
 * request's sampling parameters (cv_sampling: greedy or repetition-aware sampling with the device's counter RNG - or `host_uniforms`, two per step, when
 * sp->use_uniforms: the parity hook), the fp32 speech_embedding table (dev [n_out][d_in]) and the token capacity.  cv_lm1_decode: n_steps steps of [input Linear]
 * [72 launches of cv_lm1_step] [sampler + embedding row of the sampled token] [advance]; returns the tokens emitted by this call and whether the request ended (a stop
 * id or max_len).  The logits of a step never leave the device.  cv_lm1_step stays: it is the launch sequence inside, and the host-sampler path (parity hook). */
int cv_lm1_decode_begin(cv_lm1* m, const float* x_row, int32_t pos, const cv_sampling* sp, const float* emb_table, int32_t max_tokens, const float* host_uniforms,
                        int32_t n_uniforms, void* stream);
int cv_lm1_decode(cv_lm1* m, int32_t n_steps, int32_t* out_tokens, int32_t* n_out, int32_t* finished, void* stream);
int64_t cv_lm1_stat(const cv_lm1* m, const char* name);            /* "steps", "graph_replays", "launches_per_step"; -1 for an unknown name */
int cv_lm1_set_option(cv_lm1* m, const char* name, int32_t value); /* "graph" (env CV_LM1_GRAPH): 1 replays the step as a hipGraph, 0 (default: measured 3 % faster per token) launches it kernel by kernel;
                                                                     * "gemv_rows" / "gemv_rows16" (1 | 2 | 4; env CV_LM1_ROWS / CV_LM1_ROWS16): output rows per 16-lane group of the decode GEMVs over
                                                                     * fp32 / bf16 matrices - the same bits, a different split of the rows over workgroups */

#ifdef __cplusplus
}
#endif
#endif
