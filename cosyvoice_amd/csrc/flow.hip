// Stage driver: token -> mel flow-matching decoder (boundaries B3, B4, B5 of SURVEY.md §8b).
//   cv_flow_encoder     <- UpsampleConformerEncoder.forward            (transformer/upsample_encoder.py:244-307)
//   cv_flow_estimator   <- CausalConditionalDecoder.forward            (flow/decoder.py:405-494)
//   cv_flow_inference   <- CausalMaskedDiffWithXvec.inference + CausalConditionalCFM.forward/solve_euler
//                          (flow/flow.py:235-281, flow/flow_matching.py:71-124,203-227)
// Everything is channel-last fp32 with bf16 (or fp32) weights; the Euler loop, CFG batching and time embeddings stay on
// the device — the host only enqueues.
#include <vector>
#include <map>
#include <tuple>
#include <cmath>
#include <cstdlib>
#include "ops.h"
#include "tensor_map.h"
#include "flow_kernels.h"
#include "flow_fused.h"
#include "flow_attn32.h"
#include "flow_big.h"
#ifdef CV_BUILD_EXPERIMENTS
#include "experiments/flow_tail.h"      // measured no-go (profiles/r3_flow_tail_ab.txt): built, and its option accepted, only with -DCV_BUILD_EXPERIMENTS
#endif
#include "flow_band.h"
#include "group_norm.h"

using namespace cv;

namespace {

struct Lin { const void* w = nullptr; const float* b = nullptr; int N = 0, K = 0, Kp = 0, taps = 1; bool bf16 = true; };
struct LN { const float* g = nullptr; const float* b = nullptr; };

struct ConformerW { LN norm_mha, norm_ff; Lin qkv, pos, out, ff1, ff2; const float* bias_u; const float* bias_v; };
struct ResnetW { Lin mlp, conv1, conv2, res; LN ln1, ln2; };
struct TBlockW { LN norm1, norm3; Lin qkv, out, ff1, ff2; const u32x4_t* tail = nullptr; const float* tail_prm = nullptr; bool tail_qkv = false; const u32x4_t* band = nullptr; const u32x4_t* bandq = nullptr; const u32x4_t* lnqkv = nullptr; };   // lnqkv: the stream of flow_lnqkv_kernel (first block of a stage); tail: fragment-ordered stream of flow_tail_kernel (+ the next block's QKV); band: the stream of flow_band_kernel (flow_band.h)
struct StageW { ResnetW res; std::vector<TBlockW> tf; };
struct DitBlockW { Lin mod, qkv, out, ff1, ff2; };       // DiTBlock (flow/DiT/modules.py:500-530)
// cfg.estimator == 2, the U-Net of CosyVoice-300M (flow/decoder.py:88-291): what a stage is and what follows its transformer blocks.  kind 0 = down (its output is kept as a
// skip connection), 1 = mid, 2 = up (its input is the stream cut to the skip's length ++ the skip); post 0 = nothing, 1 = Conv1d(k 3, pad 1), 2 = Downsample1D
// (Conv1d k 3, stride 2, pad 1 as a Linear over rows of pitch 2 C: post_lin.K = 3 C), 3 = Upsample1D (ConvTranspose1d k 4, stride 2, pad 1 in polyphase form: N = 2 C, 2 taps)
struct UStageW { int kind = 1, post = 0; Lin post_lin; };

inline unsigned nblk(long long n) { return (unsigned)((n + 255) / 256); }

}  // namespace

// Options that select a measured no-go (kept as evidence, built only with -DCV_BUILD_EXPERIMENTS: VERDICT r5 item 9)
#ifdef CV_BUILD_EXPERIMENTS
static constexpr bool kExperiments = true;
#else
static constexpr bool kExperiments = false;
#endif
static void need_experiments(bool wanted, const char* what) {
    if (wanted && !kExperiments) throw Error(std::string(what) + " selects an experiment this library was built without (rebuild with CV_BUILD_EXPERIMENTS=1)");
}

struct cv_flow {
    cv_flow_config cfg{};
    TensorMap tm;
    bool finalized = false, wbf16 = true;
    // encoder
    Lin embed_lin, up_embed_lin, pre1, pre2, upconv, enc_proj, spk_affine;
    LN embed_ln, up_embed_ln, after_norm;
    std::vector<ConformerW> enc, enc_up;
    const void* input_embedding = nullptr;
    // DiT estimator + front of CausalMaskedDiffWithDiT (cfg.estimator == 1; Fun-CosyVoice3, flow/flow.py:284-414, flow/DiT/dit.py:104-176)
    Lin d_pre1, d_pre2, d_time1, d_time2, d_inproj, d_pos1, d_pos2, d_fmod, d_proj;
    std::vector<DitBlockW> dit;
    DevBuf d_x, d_y, d_n, d_qkv, d_att, d_ff, d_mod, d_fm, d_temb, d_tsin, d_th;
    // estimator
    Lin time1, time2, down_conv, up_conv, final_conv, final_proj;
    LN final_ln;
    std::vector<StageW> stages;
    std::vector<UStageW> ust;                                                                 // cfg.estimator == 2: one per stage
    DevBuf u_skip[4]; DevBuf u_gn;                                                       // ... its skip connections (one per down stage) and the GroupNorm partial sums
    int solve_cap = 0;                                                                        // cv_flow_solve: frames the f_* staging buffers hold
    // workspaces
    DevBuf e_x, e_xe, e_n, e_qkv, e_qu, e_qv, e_pe, e_p, e_bd, e_att, e_ff, e_x2, e_ctx;    // encoder
    DevBuf s_in, s_a, s_b, s_c, s_n, s_qkv, s_att, s_ff, s_skip, s_cat, s_out;                    // estimator
    DevBuf h_qk, h_vt, h_att, h_ff;                                                           // estimator, fused bf16 pipeline (flow_fused.h)
    DevBuf gemm_dbg; int gemm_dbg_on = 0;                                                     // dev tool: the same for the large-M GEMMs (option gemm_dbg = 1 QKV | 2 FF1 | 3 out-projection | 4 FF2; stats gemm_phase_<k>)
    DevBuf attn_dbg; int attn_dbg_on = 0, attn_dbg_blocks = 0;                                  // dev tool: phase stamps of the LAST attention launch (option attn_dbg, stats attn_phase_<k>)
    DevBuf h_zero;                                                                            // 64 zero bytes (the LDS-DMA source of a convolution's padded rows)
    DevBuf h_xn, h_cur;                                                                       // LayerNorm'd rows / a ResNet block's input as bf16 (flow_big.h)
    // Round 4, bf16 mode: the large-M kernel set of flow_big.h for passes of at least `big_rows` estimator rows (0 = never) - LayerNorm once per row -> bf16,
    // bf16 GEMMs with K streamed through a swizzled LDS ring, the ResNet convolutions on the same tiles.  Bit-identical to the small-tile path, so the threshold is
    // a pure speed knob.  Measured on MI355X (profiles/r4_flow_big_ab.txt, ms per shared pass small -> big): 2 utterances (M = 2696) 49.8 -> 66.9 with 128-wide
    // tiles, 4 (M = 5392) 77.9 -> 74.4, 8 (M = 10 784) 135.0 -> 109.4 with 64 x 64 tiles - which beat 128 x 64 (116.8) and 128 x 128 (136.6): these launches
    // are bound by per-workgroup latency chains (load -> LDS -> MFMA -> epilogue stores), not by re-read traffic, and more, smaller workgroups overlap them better.
    // "big_tile0" / "big_tile1": tile of the bf16-out / fp32-residual GEMMs, 0 = by measurement (64 x 64), 1 = 128x128, 2 = 128x64, 3 = 64x64, 4 = 32x64 (the N = 256 residual GEMMs and convolutions are only 676 workgroups at M = 10 784 as 64 x 64 tiles).
    // `attn2_rows`: attention with 32 queries per wave (attn_flow_kernel<.., QG = 2>) from that many rows on; 0 = never, the default: at M = 10 784 it measured
    // 54.1 us per launch against 43.0 for QG = 1 (164 registers: one 8-wave workgroup per CU instead of two).
    int big_rows = 2000, attn2_rows = 0, big_tile0 = 0, big_tile1 = 0;       // 2000 (round 5, with the 32-row band launch): from 2 utterances of U10 per pass - 48.1 -> 43.8 ms at 2, 36.3 vs 36.9 at 1 (profiles/r5_flow_band.txt); round 4 without the band: 4000 (61.6 -> 57.4 ms at 3, 48.4 vs 50.8 at 2)
    int big_lds_epi = 1;               // "big_lds_epi": the large-M GEMMs store their output tile row-wise through LDS (flow_big.h, epilogue_lds); 0 = per-lane stores from the accumulator layout
    int big_glds = 0;                  // "big_glds": the large-M GEMM stages go global -> LDS by DMA (1: global_load_lds_dwordx4, common.h CV_GLDS16) or through registers + ds_write (0).
                                       // Off: as hipcc compiles it the DMA does not overlap the MFMAs (a vmcnt(0) lands in front of the fragment reads, flow_big.h)
    int big_grid_cap = 0;              // "big_grid_cap": test hook - at most this many workgroups per persistent launch (0 = no cap)
    int big_persist = -1;              // "big_persist": -1 = one tile per workgroup; 0 = persistent workgroups, as many as the LDS admits per CU (at most 4), each walking a run of tiles with
                                       // the stage pipeline running across tile boundaries; n > 0 = n per CU.  Measured on MI355X at 8 utterances per pass (profiles/r4_flow_big_ab.txt):
                                       // 108.5 ms (-1) / 110.7 (0) / 173.8 (1) / 136.2 (2) / 116.8 (3) with 64 x 64 tiles - a tile's first-load wait and store tail are NOT what bounds these
                                       // launches (removing them bought nothing); what does is the staging itself, which more co-resident waves overlap better.
    int vt_pitch = 0;                  // row pitch of V^T = round_up(T capacity, 64)
    int tail_ring = 8;                 // weight fragments (1 KB each) a wave of flow_tail_kernel keeps in flight: 8 or 16 (option "tail_ring", env CV_FLOW_TAIL_RING)
    int fused_tail = 0;                // bf16 mode: 1 = everything after a block's attention in ONE launch per 16-row band (flow_tail.h).  Measured on MI355X
                                       // (profiles/r3_flow_tail_ab.txt): 46.3 vs 38.5 ms per flow.inference at batch 1 - a workgroup pulls its 2 MB of weights through one CU's
                                       // L1 at ~45 B/clk (~32 KB in flight, ~700 cycles), 33-40 us per launch whatever the band count, against 37 us for the four launches it
                                       // replaces.  Off by default; bit-identical to the five-launch form, tested both ways.
    int ln_qkv = 1;                    // with band_qkv: the FIRST block of a stage gets its LayerNorm + QKV GEMM from one band launch too (flow_lnqkv_kernel, round 6; option "ln_qkv", env CV_FLOW_LN_QKV)
    int band_qkv = 1;                  // with fused_band: the band launch also runs the NEXT block's QKV GEMM (flow_band_kernel<.., HAS_QKV>): a block of a large pass is two launches
                                       // (attention, band); bit-identical; option "band_qkv", env CV_FLOW_BAND_QKV
    int band_pipe = 2;                 // option "band_pipe" (env CV_FLOW_BAND_PIPE): the FF1 -> GELU -> FF2 chunks of a band as a software pipeline (flow_band_kernel<.., PIPE>): 0 = never,
                                       // 1 = 48-row bands, 2 = 48- and 32-row bands (default; a 32-row band then holds 96 KB of LDS - one workgroup per CU instead of two - and is
                                       // still faster: 8 / 6 / 4 utterances of U10 per pass 79.1 -> 77.6 / 64.9 -> 63.4 / 52.1 -> 50.8 ms, profiles/r5_band_qkv.txt section 4)
    int band_stagger = 0;              // experiments builds only, option "band_stagger" (env CV_FLOW_BAND_STAGGER): FlowBandArgs::stagger; 0 = all bands of a launch start together
    int band_bm = 0;                   // option "band_bm" (env CV_FLOW_BAND_BM): rows per band 32 / 48 / 64 whatever the row count; 0 = by the row count of the pass (band_rows_for)
    int fused_band = 1;                // bf16 mode, large passes (big_rows): everything between a block's attention and the next block's QKV GEMM in ONE launch per 64-row band
                                       // (flow_band.h) instead of five (out-projection, LayerNorm, FF1, FF2, LayerNorm); bit-identical; option "fused_band", env CV_FLOW_BAND
    int fused = 1;                     // bf16 mode: LN-prologue GEMMs + bf16 activations + bf16 flash attention for the transformer blocks
    // tuning knobs of the fused pipeline.  "flow_tile": 0 = by size (one round of workgroups, see ln_gemm_bf16), 1 = 64x64, 2 = 64x128, 3 = 32x64,
    // 4 = 64x192; "attn_waves": 2 | 4 waves (32 | 64 queries) per workgroup; "attn_kt": 64-key tiles per iteration (1 | 2)
    int flow_ntile = 0;                // "flow_ntile": N tiles per workgroup of the LayerNorm-prologue GEMMs: 0 = by grid size (two when that makes one round), 1, 2 (env CV_FLOW_NTILE)
    int flow_tile = 0, attn_waves = 4, attn_kt = 1, attn_ks = 2;   // "attn_ks": key splits inside a 64-query workgroup (2 = 8 waves, 128 keys per iteration)
    int res_tile = 1;      // tile of the small-pass residual GEMMs (out-projection, FF2): 1 = 32 x 32 (344 workgroups at M = 1348: 37.5 -> 36.1 ms per flow.inference, round 6), 0 = 32 x 64 (option "res_tile", env CV_FLOW_RES_TILE)
    int attn32_waves = 0;  // waves (x 32 queries) per workgroup of attn_flow32_kernel: 0 = 4 (the measured choice at every size), 2 | 4 (option "attn32_waves", env CV_FLOW_ATTN32_WAVES)
    int attn32 = 1;        // round 6: every bf16 attention of the estimator on attn_flow32_kernel (flow_attn32.h: 32 queries per wave on 32x32x16 tiles); 0 = attn_flow_kernel as configured above
    DevBuf t_val, t_sin, t_h, t_emb, t_mlp;                                                   // time embeddings
    DevBuf f_tok, f_h, f_mu, f_spk, f_spkn, f_cond, f_x, f_ones;                              // inference glue
    long long enc_rows = 0;            // row capacity (2 x tokens x utterances) of the encoder's row-wise workspaces
    int enc_nu = 0, enc_batch = 1;        // enc_batch: encode the utterances of an equal-length pass together (option "enc_batch", CV_FLOW_ENC_BATCH)
    int enc_cap = 0, est_cap = 0, est_nz = 0, t_cap = 0, inf_cap = 0; long long inf_rows = 0;
    // padded batches (cv_flow_inference_ragged): key counts per estimator batch row, fixed device address (captured graphs read the current content)
    DevBuf klen; std::vector<int> host_klen; const int* cur_klen = nullptr;
    // hipGraph cache of the whole Euler solve, keyed by (T, n_steps, streaming): ~5000 launches per utterance become one replay.
    // A key is captured the second time it is seen (streaming requests change T every chunk and would only pay the instantiation).
    bool use_graph = true;
    int graph_max_rows = 1000;         // "graph_max_rows": passes of at least this many estimator rows (2 x utterances x T) are not captured (0 = capture everything); CV_FLOW_GRAPH_MAX_ROWS.
                                       // 3000 until round 6; 1000 since: at 1348 rows (one utterance of U10) the launches average 9 us against the host's ~4 us per launch, the host stays
                                       // ahead and replaying the 3640-node graph costs more than issuing it - flow.inference 36.3 -> 34.9 ms, the bench 177.3 -> 176.0 ms per U10; the
                                       // chunks of a streaming request (460 rows and up: 5 - 7 us launches) keep their graphs (8 clients: 217.7 vs 217.7 audio-s/s, p50 102.7 vs 103.1 ms);
                                       // no graphs at all costs the 8-client run 3 ms of p50 (profiles/r6_flow_graph_threshold.txt)
    int bf16_mfma = 0;                 // 1: Linear / Conv1d products on the bf16 MFMA (activations rounded to bf16 in LDS), 0: fp32-accurate (three-term split for bf16 weights, fp32 MFMA chain otherwise)
    std::map<std::tuple<int, int, int>, hipGraphExec_t> graphs;
    std::map<std::tuple<int, int, int>, unsigned long long> graph_used; unsigned long long graph_clock = 0;   // last use per key (least-recently-used eviction)
    size_t graph_cap = 32;             // option "graph_cap": shared passes of a serving scheduler see (utterances per pass) x (chunk shapes) keys per lane: 16 for U10-like traffic
    int graph_captures = 0;            // captures so far (cv_flow_get_stat: tests of the eviction rule)
    std::map<std::tuple<int, int, int>, int> seen;
    hipStream_t own_stream = nullptr;
    // Round 3, est_streams = 2: the estimator's batch rows as TWO launch chains (rows only meet in the CFG combine of an Euler step) - the second half of
    // the batch rows is forked onto side_stream for every estimator evaluation and joined before the Euler update (graph edges inside a captured solve).
    // Same kernels on the same rows: bit-identical to one chain (tests/test_flow.py).  Measured on MI355X (profiles/r3_flow_two_stream_ab.txt): an isolated
    // pass gains 1 % at batch 1 (38.8 -> 38.4 ms), 7 - 10 % for 2 - 8 utterances per pass, but a hipGraph with parallel branches leaves the runtime's
    // single-batch replay path: hipGraphLaunch then costs ~40 ms of HOST time per solve (bench step 178 -> 218 ms once nothing overlaps it).  Off by default.
    int est_streams = 1;
    // Round 5: passes that run EAGER (graph_max_rows: every shared pass of 3000 estimator rows or more) can take the two-chain form without that cost - the host stays
    // ~10 ms of enqueueing ahead of an 80 ms pass: option "eager_streams" = 2 (env CV_FLOW_EAGER_STREAMS).  An ISOLATED pass gains 4 - 8 % (8 utterances of U10 81.1 ->
    // 74.5 ms, 6: 67.6 -> 62.4, 4: 54.2 -> 50.9, 2: 44.0 -> 42.3), but the token2wav LANES of the model already are two chains at a coarser grain, and four busy vocoder
    // streams next to the LM lose: batch 16 393 -> 329, batch 32 494 -> 423, mixed64 478 -> 443 audio-s/s with two lanes; with one lane two chains beat one (340 -> 353,
    // 408 -> 427, 413 -> 429) and still lose to two lanes of one chain (profiles/r5_band_qkv.txt section 3).  Default 1.
    int eager_streams = 1;
    bool two_chains_now = false;       // set by solve_euler around an eager body
    long long rule_rows = 0;           // rows of the whole pass while its halves run as two chains (band_rows_for sees what the chip holds, not one chain's share)
    hipStream_t side_stream = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    std::vector<float> host_t;
    ~cv_flow() {
        for (auto& g : graphs) (void)hipGraphExecDestroy(g.second);
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_join) (void)hipEventDestroy(ev_join);
        if (side_stream) (void)hipStreamDestroy(side_stream);
        if (own_stream) (void)hipStreamDestroy(own_stream);
    }
};

// Graph capture is illegal on the legacy default stream: NULL maps to a handle-owned blocking stream (implicitly ordered with it).
static hipStream_t resolve(cv_flow* m, void* s) {
    if (s) return as_stream(s);
    if (!m->own_stream) CV_HIP(hipStreamCreate(&m->own_stream));
    return m->own_stream;
}

static Lin get_lin(const cv_flow* m, const std::string& name, int N, int K, int taps, bool bias) {
    Lin l; l.N = N; l.K = K; l.Kp = round_up32(K); l.taps = taps; l.bf16 = m->wbf16;
    const long long numel = (long long)N * taps * l.Kp;
    l.w = m->tm.get(name + ".w", m->wbf16 ? CV_BF16 : CV_F32, numel).p;
    l.b = bias ? m->tm.f32(name + ".b", N) : nullptr;
    return l;
}
static LN get_ln(const cv_flow* m, const std::string& name, int C) { LN n; n.g = m->tm.f32(name + ".g", C); n.b = m->tm.f32(name + ".b", C); return n; }

static void flow_finalize(cv_flow* m) {
    const auto& c = m->cfg;
    CV_CHECK(c.mel == 80, "flow: mel must be 80 (solve_euler hard-codes it, flow_matching.py:95)");
    CV_CHECK(c.estimator >= 0 && c.estimator <= 2, "flow: estimator must be 0 (causal U-Net), 1 (DiT) or 2 (the U-Net of CosyVoice-300M)");
    const bool unet1 = c.estimator == 2;      // ConditionalDecoder alone: the rest of MaskedDiffWithXvec (another encoder family, length regulator, flow cache) stays with the caller
    CV_CHECK(c.estimator == 1 || (unet1 ? c.est_ch % 32 == 0 : (c.dim % 64 == 0 && c.dim / c.enc_heads == 64 && c.est_ch % 32 == 0)), "flow: head_dim is fixed at 64");
    const char* probe = unet1 ? "est.time1.w" : "spk_affine.w";
    CV_CHECK(m->tm.has(probe), std::string("flow: missing tensor '") + probe + "'");
    m->wbf16 = m->tm.t.at(probe).dtype == CV_BF16;
    const int d = c.dim, C = c.est_ch, inner = c.est_heads * 64, tdim = 4 * C, cin = 4 * c.mel;
    if (!unet1) {
        m->input_embedding = m->tm.get("input_embedding", m->wbf16 ? CV_BF16 : CV_F32, (long long)c.vocab * d).p;
        m->spk_affine = get_lin(m, "spk_affine", c.mel, c.spk_dim, 1, true);
    }
    if (c.estimator == 1) {            // CausalMaskedDiffWithDiT: no conformer encoder, DiT estimator
        const int D = c.est_ch, Cp = c.ffn, ffi = c.est_mid * D;
        CV_CHECK(d == c.mel && D % 64 == 0 && D / c.est_heads == 64 && D % 16 == 0 && (D / 16) % 4 == 0 && c.enc_blocks == 0 && c.up_blocks == 0,
                 "flow(dit): input_size must equal mel, head_dim 64, 16 position-conv groups of a multiple of 4 channels");
        m->d_pre1 = get_lin(m, "dit.pre.conv1", Cp, d, c.pre_lookahead + 1, true); m->d_pre2 = get_lin(m, "dit.pre.conv2", d, Cp, 3, true);
        m->d_time1 = get_lin(m, "dit.time1", D, 256, 1, true); m->d_time2 = get_lin(m, "dit.time2", D, D, 1, true);
        m->d_inproj = get_lin(m, "dit.in_proj", D, 4 * c.mel, 1, true);
        m->d_pos1 = get_lin(m, "dit.pos.conv1", D, D / 16, 31, true); m->d_pos2 = get_lin(m, "dit.pos.conv2", D, D / 16, 31, true);
        for (int i = 0; i < c.est_blocks; ++i) {
            const std::string p = "dit.blk." + std::to_string(i) + ".";
            DitBlockW b;
            b.mod = get_lin(m, p + "mod", 6 * D, D, 1, true); b.qkv = get_lin(m, p + "qkv", 3 * D, D, 1, true); b.out = get_lin(m, p + "out", D, D, 1, true);
            b.ff1 = get_lin(m, p + "ff1", ffi, D, 1, true); b.ff2 = get_lin(m, p + "ff2", D, ffi, 1, true);
            m->dit.push_back(b);
        }
        m->d_fmod = get_lin(m, "dit.final_mod", 2 * D, D, 1, true); m->d_proj = get_lin(m, "dit.proj_out", c.mel, D, 1, true);
        m->finalized = true;
        return;
    }
    if (!unet1) {
    m->embed_lin = get_lin(m, "enc.embed.lin", d, d, 1, true); m->embed_ln = get_ln(m, "enc.embed.ln", d);
    m->up_embed_lin = get_lin(m, "enc.up_embed.lin", d, d, 1, true); m->up_embed_ln = get_ln(m, "enc.up_embed.ln", d);
    m->after_norm = get_ln(m, "enc.after_norm", d);
    m->pre1 = get_lin(m, "enc.pre.conv1", d, d, c.pre_lookahead + 1, true);
    m->pre2 = get_lin(m, "enc.pre.conv2", d, d, 3, true);
    m->upconv = get_lin(m, "enc.up.conv", d, d, 5, true);
    m->enc_proj = get_lin(m, "encoder_proj", c.mel, d, 1, true);
    auto conformer = [&](const std::string& p) {
        ConformerW w;
        w.norm_mha = get_ln(m, p + "norm_mha", d); w.norm_ff = get_ln(m, p + "norm_ff", d);
        w.qkv = get_lin(m, p + "qkv", 3 * d, d, 1, true); w.pos = get_lin(m, p + "pos", d, d, 1, false);
        w.out = get_lin(m, p + "out", d, d, 1, true);
        w.ff1 = get_lin(m, p + "ff1", c.ffn, d, 1, true); w.ff2 = get_lin(m, p + "ff2", d, c.ffn, 1, true);
        w.bias_u = m->tm.f32(p + "bias_u", d); w.bias_v = m->tm.f32(p + "bias_v", d);
        return w;
    };
    for (int i = 0; i < c.enc_blocks; ++i) m->enc.push_back(conformer("enc.layers." + std::to_string(i) + "."));
    for (int i = 0; i < c.up_blocks; ++i) m->enc_up.push_back(conformer("enc.up_layers." + std::to_string(i) + "."));
    }
    m->time1 = get_lin(m, "est.time1", tdim, cin, 1, true); m->time2 = get_lin(m, "est.time2", tdim, tdim, 1, true);
    int nst = c.est_mid + 2, n_down = 1;
    if (unet1) {                       // len(channels) down stages, est_mid mid stages, len(channels) up stages - counted from the tensors that were set
        nst = 0;
        while (m->tm.has("est.stage." + std::to_string(nst) + ".res.mlp.w")) ++nst;
        n_down = (nst - c.est_mid) / 2;
        CV_CHECK(n_down >= 1 && n_down <= 4 && nst == 2 * n_down + c.est_mid, "flow(unet1): stages must be n down + est_mid mid + n up, n <= 4");
    }
    for (int s = 0; s < nst; ++s) {
        const std::string p = "est.stage." + std::to_string(s) + ".";
        const int din = s == 0 ? cin : (s >= nst - n_down ? 2 * C : C);
        StageW st;
        st.res.mlp = get_lin(m, p + "res.mlp", C, tdim, 1, true);
        st.res.conv1 = get_lin(m, p + "res.block1.conv", C, din, 3, true); st.res.ln1 = get_ln(m, p + "res.block1.ln", C);
        st.res.conv2 = get_lin(m, p + "res.block2.conv", C, C, 3, true); st.res.ln2 = get_ln(m, p + "res.block2.ln", C);
        st.res.res = get_lin(m, p + "res.res", C, din, 1, true);
        for (int j = 0; j < c.est_blocks; ++j) {
            const std::string q = p + "tf." + std::to_string(j) + ".";
            TBlockW t;
            t.norm1 = get_ln(m, q + "norm1", C); t.norm3 = get_ln(m, q + "norm3", C);
            t.qkv = get_lin(m, q + "qkv", 3 * inner, C, 1, false); t.out = get_lin(m, q + "out", C, inner, 1, true);
            t.ff1 = get_lin(m, q + "ff1", 4 * C, C, 1, true); t.ff2 = get_lin(m, q + "ff2", C, 4 * C, 1, true);
            if (m->wbf16 && m->tm.has(q + "tail_prm") && ((C == 256 && inner == 512) || (C == 64 && inner == 64))) {
                t.tail_qkv = j + 1 < c.est_blocks;
                const long long frags = (long long)(C / 64) * (inner / 32) + (long long)(4 * C / 64) * (C / 32) + (long long)(C / 64) * (4 * C / 32) +
                                        (t.tail_qkv ? 3LL * (inner / 64) * (C / 32) : 0);
                if (m->tm.has(q + "tail"))                         // (the 16-row tail stream: packed by weights.py for CV_BUILD_EXPERIMENTS libraries only)
                    t.tail = reinterpret_cast<const u32x4_t*>(m->tm.get(q + "tail", CV_BF16, 4 * frags * 64 * 8).p);
                t.tail_prm = m->tm.f32(q + "tail_prm", 6LL * C + 4 * C);
                if (m->tm.has(q + "band")) {                       // the 64-row band form of large passes (flow_band.h): out-projection + FF1 + FF2 fragments, 8 waves at C = 256, 4 at C = 64
                    const long long bfr = (long long)(C / 16) * (inner / 32) + 2LL * (C / 16) * (4 * C / 32);
                    t.band = reinterpret_cast<const u32x4_t*>(m->tm.get(q + "band", CV_BF16, bfr * 64 * 8).p);
                    if (m->tm.has(q + "lnqkv"))                    // LayerNorm + QKV of THIS block as one band launch (flow_lnqkv_kernel: the first block of a stage)
                        t.lnqkv = reinterpret_cast<const u32x4_t*>(m->tm.get(q + "lnqkv", CV_BF16, 3LL * (inner / 16) * (C / 32) * 64 * 8).p);
                    if (t.tail_qkv && m->tm.has(q + "bandq"))      // the same stream + the NEXT block's QKV GEMM (flow_band_kernel<.., HAS_QKV>)
                        t.bandq = reinterpret_cast<const u32x4_t*>(m->tm.get(q + "bandq", CV_BF16, (bfr + 3LL * (inner / 16) * (C / 32)) * 64 * 8).p);
                }
            }
            st.tf.push_back(t);
        }
        m->stages.push_back(st);
        if (unet1) {
            UStageW u;
            u.kind = s < n_down ? 0 : (s < n_down + c.est_mid ? 1 : 2);
            if (u.kind == 0) { u.post = s == n_down - 1 ? 1 : 2; u.post_lin = u.post == 1 ? get_lin(m, p + "post", C, C, 3, true) : get_lin(m, p + "post", C, 3 * C, 1, true); }
            if (u.kind == 2) { u.post = s == nst - 1 ? 1 : 3; u.post_lin = u.post == 1 ? get_lin(m, p + "post", C, C, 3, true) : get_lin(m, p + "post", 2 * C, C, 2, true); }
            m->ust.push_back(u);
        }
    }
    if (const char* e = getenv("CV_FLOW_NTILE")) m->flow_ntile = atoi(e);
    // dev knobs of the large-M kernel set for A/B runs through bench.py (options of the same names without the prefix)
    if (const char* e = getenv("CV_FLOW_GRAPH_MAX_ROWS")) m->graph_max_rows = atoi(e);
    if (const char* e = getenv("CV_FLOW_BIG_ROWS")) m->big_rows = atoi(e);
    if (const char* e = getenv("CV_FLOW_BIG_PERSIST")) m->big_persist = atoi(e);
    if (const char* e = getenv("CV_FLOW_BIG_TILE0")) m->big_tile0 = atoi(e);
    if (const char* e = getenv("CV_FLOW_BIG_TILE1")) m->big_tile1 = atoi(e);
    if (const char* e = getenv("CV_FLOW_BIG_GLDS")) m->big_glds = atoi(e) != 0;
    if (const char* e = getenv("CV_FLOW_BIG_LDS_EPI")) m->big_lds_epi = atoi(e) != 0;
    if (const char* e = getenv("CV_FLOW_ENC_BATCH")) m->enc_batch = atoi(e) != 0;
    if (const char* e = getenv("CV_FLOW_ATTN2_ROWS")) m->attn2_rows = atoi(e);
    if (kExperiments) {                                                              // A/B knobs of experiment builds
        if (const char* e = getenv("CV_FLOW_ATTN32")) m->attn32 = e[0] != '0';
        if (const char* e = getenv("CV_FLOW_EAGER_STREAMS")) m->eager_streams = atoi(e) >= 2 ? 2 : 1;
        if (const char* e = getenv("CV_FLOW_TAIL")) m->fused_tail = e[0] != '0';
        if (const char* e = getenv("CV_FLOW_BAND_STAGGER")) m->band_stagger = atoi(e) > 0 ? atoi(e) : 0;
    }
    if (const char* e = getenv("CV_FLOW_RES_TILE")) m->res_tile = atoi(e) == 1;
    if (const char* e = getenv("CV_FLOW_ATTN32_WAVES")) { const int v = atoi(e); if (v == 0 || v == 2 || v == 4) m->attn32_waves = v; }
    if (const char* e = getenv("CV_FLOW_BAND_QKV")) m->band_qkv = e[0] != '0';
    if (const char* e = getenv("CV_FLOW_LN_QKV")) m->ln_qkv = e[0] != '0';
    if (const char* e = getenv("CV_FLOW_BAND_BM")) m->band_bm = atoi(e);
    if (const char* e = getenv("CV_FLOW_BAND_PIPE")) m->band_pipe = atoi(e);
    if (const char* e = getenv("CV_FLOW_BAND")) m->fused_band = e[0] != '0';        // dev knob for A/B runs (also: option "fused_band")
    if (const char* e = getenv("CV_FLOW_TAIL_RING")) m->tail_ring = atoi(e) == 16 ? 16 : 8;
    if (!unet1) { m->down_conv = get_lin(m, "est.down_conv", C, C, 3, true); m->up_conv = get_lin(m, "est.up_conv", C, C, 3, true); }
    m->final_conv = get_lin(m, "est.final.conv", C, C, 3, true); m->final_ln = get_ln(m, "est.final.ln", C);
    m->final_proj = get_lin(m, "est.final_proj", c.mel, C, 1, true);
    m->finalized = true;
}

// precision of the Linear / Conv1d products issued by the current entry point (set from the handle's option for the duration of a call)
static thread_local int tl_bf16_mfma = 0;
static thread_local int tl_flow_tile = 0, tl_attn_waves = 4, tl_attn_kt = 2, tl_attn_ks = 1, tl_flow_ntile = 0, tl_attn32 = 1, tl_attn32_waves = 0, tl_res_tile = 1;     // tuning knobs of the fused pipeline, per call like the precision
static thread_local long long* tl_attn_dbg = nullptr; static thread_local long long* tl_gemm_dbg = nullptr; static thread_local int tl_gemm_dbg_which = 0;
static thread_local int tl_big_tile0 = 0, tl_big_tile1 = 0, tl_big_persist = -1, tl_big_grid_cap = 0, tl_big_glds = 0, tl_big_lds_epi = 1;
struct PrecisionScope {
    int prev, pt, pw, pk, ps, pn, p32, p32w, prt, pb0, pb1, pbp, pbc, pbg, pbe;
    explicit PrecisionScope(const cv_flow* m) : prev(tl_bf16_mfma), pt(tl_flow_tile), pw(tl_attn_waves), pk(tl_attn_kt), ps(tl_attn_ks), pn(tl_flow_ntile), p32(tl_attn32), p32w(tl_attn32_waves), prt(tl_res_tile), pb0(tl_big_tile0), pb1(tl_big_tile1), pbp(tl_big_persist), pbc(tl_big_grid_cap), pbg(tl_big_glds), pbe(tl_big_lds_epi) {
        tl_bf16_mfma = m->bf16_mfma; tl_flow_tile = m->flow_tile; tl_attn_waves = m->attn_waves; tl_attn_kt = m->attn_kt; tl_attn_ks = m->attn_ks; tl_flow_ntile = m->flow_ntile; tl_attn32 = m->attn32; tl_attn32_waves = m->attn32_waves; tl_res_tile = m->res_tile;
        tl_big_tile0 = m->big_tile0; tl_big_tile1 = m->big_tile1; tl_big_persist = m->big_persist; tl_big_grid_cap = m->big_grid_cap; tl_big_glds = m->big_glds; tl_big_lds_epi = m->big_lds_epi; tl_attn_dbg = m->attn_dbg_on ? const_cast<cv_flow*>(m)->attn_dbg.as<long long>() : nullptr;
        tl_gemm_dbg = m->gemm_dbg_on ? const_cast<cv_flow*>(m)->gemm_dbg.as<long long>() : nullptr; tl_gemm_dbg_which = m->gemm_dbg_on;
    }
    ~PrecisionScope() { tl_bf16_mfma = prev; tl_flow_tile = pt; tl_attn_waves = pw; tl_attn_kt = pk; tl_attn_ks = ps; tl_flow_ntile = pn; tl_attn32 = p32; tl_attn32_waves = p32w; tl_res_tile = prt; tl_big_tile0 = pb0; tl_big_tile1 = pb1; tl_big_persist = pbp; tl_big_grid_cap = pbc; tl_big_glds = pbg; tl_big_lds_epi = pbe; tl_attn_dbg = nullptr; tl_gemm_dbg = nullptr; tl_gemm_dbg_which = 0; }
};

// ---- generic conv/linear on channel-last activations -----------------------------------------------------------------
// rows: M per batch, `a_rows` valid input rows per batch (zero padding outside), tap j reads row (m + j*dil - pad_left)
static void conv_cl(const Lin& l, const float* A, int a_rows, int M, int batch, int pad_left, int dil, float* C, int act, float act_p,
                    const float* res, hipStream_t s, int pro = ACT_NONE, float pro_p = 0.f, const float* row_scale = nullptr, int ldc = -1) {
    GemmConvArgs a{};
    a.A = A; a.a_batch = (long long)a_rows * l.K; a.a_len = (long long)a_rows * l.K; a.lda = l.K; a.a_off0 = -pad_left * l.K;
    a.tap_step = dil * l.K; a.taps = l.taps; a.K = l.K; a.pro = pro; a.pro_p = pro_p; a.pro_alpha = nullptr;
    a.W = l.w; a.Kp = l.Kp; a.ldw = 0; a.w_batch = 0; a.bias = l.b;
    if (ldc < 0) ldc = l.N;
    a.C = C; a.c_batch = (long long)M * ldc; a.c_len = (long long)M * ldc; a.ldc = ldc; a.c_off = 0; a.M = M; a.N = l.N;
    a.act = act; a.act_p = act_p; a.res = res; a.res_batch = (long long)M * ldc; a.out_scale = 1.f;
    a.row_scale = row_scale; a.row_scale_batch = M; a.accumulate = 0;
    a.a_bf16 = tl_bf16_mfma && l.bf16;
    gemm_conv(a, l.bf16, batch, s);
}
static void lin_cl(const Lin& l, const float* A, long long rows, float* C, int act, const float* res, hipStream_t s, int pro = ACT_NONE) {
    conv_cl(l, A, (int)rows, (int)rows, 1, 0, 1, C, act, 0.f, res, s, pro);
}
static void ln_rows(const LN& n, const float* x, float* y, long long rows, int C, float eps, hipStream_t s, int act = ACT_NONE, float scale = 1.f,
                    const float* col_add = nullptr, long long rows_per_batch = 0) {
    norm_rows(NormArgs{x, y, rows, C, n.g, n.b, eps, 0, act, scale, nullptr, col_add, rows_per_batch > 0 ? rows_per_batch : rows}, s);
}

// ---- encoder ---------------------------------------------------------------------------------------------------------
static void enc_reserve(cv_flow* m, int T, int nu = 1) {      // T = token count before upsampling, nu = utterances of that length encoded together
    // Two capacities: the per-utterance length T sizes the position tables and the rel-pos score buffer; the ROW count 2 T nu sizes everything row-wise.
    // Kept as independent maxima they made one long utterance after an 8-utterance pass of short chunks reserve 8 x the rows it needs.
    const long long rows = 2LL * T * nu;
    if (T <= m->enc_cap && nu <= m->enc_nu && rows <= m->enc_rows) return;
    T = std::max(T, m->enc_cap); nu = std::max(nu, m->enc_nu);
    const auto& c = m->cfg; const size_t d = c.dim, T2 = 2 * (size_t)T, R2 = (size_t)std::max(rows, m->enc_rows), f = 4;
    m->e_x.ensure(R2 * d * f); m->e_xe.ensure((R2 + 8 * nu) * d * f); m->e_n.ensure(R2 * d * f); m->e_qkv.ensure(R2 * 3 * d * f);
    m->e_qu.ensure(R2 * d * f); m->e_qv.ensure(R2 * d * f); m->e_pe.ensure((2 * T2) * d * f); m->e_p.ensure((2 * T2) * d * f);
    m->e_bd.ensure((size_t)c.enc_heads * T2 * (2 * T2) * f); m->e_att.ensure(R2 * d * f); m->e_ff.ensure(R2 * c.ffn * f);
    m->e_x2.ensure(R2 * d * f); m->e_ctx.ensure(8 * nu * d * f);
    m->enc_cap = T; m->enc_nu = nu; m->enc_rows = (long long)R2;
}

// x: [nu][T][d] stacked.  Everything row-wise runs ONCE over the nu * T rows; linear_pos(pe) once per layer (the reference recomputes it per call); matrix_bd and
// the attention per utterance (one [H][T][2T-1] score-bias tensor at a time: 58 MB at T = 674).  Per row the arithmetic is the single-utterance call's.
static void conformer_layer(cv_flow* m, const ConformerW& w, float* x, int T, const float* pe, int chunk, hipStream_t s, int nu = 1) {
    const auto& c = m->cfg; const int d = c.dim, H = c.enc_heads, P = 2 * T - 1; const long long R = (long long)nu * T;
    float* n = m->e_n.as<float>(); float* qkv = m->e_qkv.as<float>(); float* qu = m->e_qu.as<float>(); float* qv = m->e_qv.as<float>();
    float* pp = m->e_p.as<float>(); float* bd = m->e_bd.as<float>(); float* att = m->e_att.as<float>(); float* ff = m->e_ff.as<float>();
    ln_rows(w.norm_mha, x, n, R, d, 1e-12f, s);
    lin_cl(w.qkv, n, R, qkv, ACT_NONE, nullptr, s);
    lin_cl(w.pos, pe, P, pp, ACT_NONE, nullptr, s);
    hipLaunchKernelGGL(add_pos_bias_kernel, dim3(nblk(R * d)), dim3(256), 0, s, qkv, 3 * d, w.bias_u, w.bias_v, qu, qv, (int)R, d);
    for (int u = 0; u < nu; ++u) {
        const long long r0 = (long long)u * T;
        {   // matrix_bd[h] = (q + v)[h] @ p[h]^T  -> [H][T][2T-1]   (attention.py:318)
            GemmConvArgs a{};
            a.A = qv + r0 * d; a.a_batch = 64; a.a_len = (long long)T * d; a.lda = d; a.a_off0 = 0; a.tap_step = 0; a.taps = 1; a.K = 64;
            a.pro = ACT_NONE; a.W = pp; a.Kp = 64; a.ldw = d; a.w_batch = 64; a.bias = nullptr;
            a.C = bd; a.c_batch = (long long)T * P; a.c_len = (long long)T * P; a.ldc = P; a.c_off = 0; a.M = T; a.N = P;
            a.act = ACT_NONE; a.res = nullptr; a.out_scale = 1.f; a.row_scale = nullptr; a.accumulate = 0;
            // a_len is a range check on the flat index of batch b (base + b*64): the last head reads up to T*d - 1 from ITS base
            a.a_len = (long long)(T - 1) * d + 64;
            gemm_conv(a, false, H, s);
        }
        AttnArgs at{};
        at.q = qu + r0 * d; at.q_batch = 0; at.q_row = d; at.q_head = 64;
        at.k = qkv + r0 * 3 * d + d; at.k_batch = 0; at.k_row = 3 * d; at.k_head = 64;
        at.v = qkv + r0 * 3 * d + 2 * d; at.v_batch = 0; at.v_row = 3 * d; at.v_head = 64;
        at.o = att + r0 * d; at.o_batch = 0; at.o_row = d; at.o_head = 64;
        at.B = 1; at.H = H; at.kv_group = 1; at.Tq = T; at.Tk = T; at.scale = 0.125f;
        at.mask_mode = chunk > 0 ? MASK_CHUNK : MASK_NONE; at.chunk = chunk;
        at.rel_bd = bd; at.bd_batch = 0; at.bd_head = (long long)T * P; at.bd_row = P;
        at.bf16 = tl_bf16_mfma;
        attention(at, s);
    }
    lin_cl(w.out, att, R, x, ACT_NONE, x, s);
    ln_rows(w.norm_ff, x, n, R, d, 1e-12f, s);
    lin_cl(w.ff1, n, R, ff, ACT_SILU, nullptr, s);
    lin_cl(w.ff2, ff, R, x, ACT_NONE, x, s);
}

// tok_emb [nu][tok_pitch rows][d] (already masked; the first T rows of every utterance are encoded), ctx: the pre_lookahead rows that follow them (row T .. T + la - 1
// of every utterance) when `has_ctx`, zeros otherwise  ->  h_out [nu][2T][d]
static void flow_encoder(cv_flow* m, const float* tok_emb, int T, const float* ctx, int streaming, float* h_out, hipStream_t s, int nu = 1, int tok_pitch = 0) {
    const auto& c = m->cfg; const int d = c.dim, la = c.pre_lookahead;
    CV_CHECK(T > 0 && nu >= 1, "flow_encoder: empty input");
    if (tok_pitch <= 0) tok_pitch = T;
    enc_reserve(m, T, nu);
    float* x = m->e_x.as<float>(); float* xe = m->e_xe.as<float>(); float* n = m->e_n.as<float>(); float* pe = m->e_pe.as<float>();
    const float xscale = sqrtf((float)d);
    const long long R = (long long)nu * T;
    const size_t rowb = (size_t)d * 4;
    // embed: Linear -> LayerNorm(1e-5) -> * sqrt(d)     (subsampling.py:83-113, embedding.py:256-270)
    const float* in = tok_emb;
    if (nu > 1 && tok_pitch != T) {        // compact the utterances' first T rows (the GEMM takes contiguous rows)
        CV_HIP(hipMemcpy2DAsync(m->e_x2.p, (size_t)T * rowb, tok_emb, (size_t)tok_pitch * rowb, (size_t)T * rowb, nu, hipMemcpyDeviceToDevice, s));
        in = m->e_x2.as<float>();
    }
    lin_cl(m->embed_lin, in, R, n, ACT_NONE, nullptr, s);
    ln_rows(m->embed_ln, n, x, R, d, 1e-5f, s, ACT_NONE, xscale);
    hipLaunchKernelGGL(rel_pos_emb_kernel, dim3(2 * T - 1), dim3(256), 0, s, pe, T, d);
    // PreLookaheadLayer (upsample_encoder.py:82-103): xe = [x ; ctx or zeros] per utterance, conv1 k=la+1 looks right, leaky_relu(0.01),
    // causal conv2 k3, + residual
    CV_HIP(hipMemcpy2DAsync(xe, (size_t)(T + la) * rowb, x, (size_t)T * rowb, (size_t)T * rowb, nu, hipMemcpyDeviceToDevice, s));
    if (ctx) {
        float* cx = m->e_ctx.as<float>();
        if (nu > 1) { CV_HIP(hipMemcpy2DAsync(cx, (size_t)la * rowb, ctx, (size_t)tok_pitch * rowb, (size_t)la * rowb, nu, hipMemcpyDeviceToDevice, s)); }
        lin_cl(m->embed_lin, nu > 1 ? cx : ctx, (long long)nu * la, n, ACT_NONE, nullptr, s);
        if (nu == 1) ln_rows(m->embed_ln, n, xe + (size_t)T * d, la, d, 1e-5f, s, ACT_NONE, xscale);
        else {
            ln_rows(m->embed_ln, n, cx, (long long)nu * la, d, 1e-5f, s, ACT_NONE, xscale);
            CV_HIP(hipMemcpy2DAsync(xe + (size_t)T * d, (size_t)(T + la) * rowb, cx, (size_t)la * rowb, (size_t)la * rowb, nu, hipMemcpyDeviceToDevice, s));
        }
    } else {
        CV_HIP(hipMemset2DAsync(xe + (size_t)T * d, (size_t)(T + la) * rowb, 0, (size_t)la * rowb, nu, s));
    }
    float* y1 = m->e_x2.as<float>();
    conv_cl(m->pre1, xe, T + la, T, nu, 0, 1, y1, ACT_LEAKY, 0.01f, nullptr, s);
    conv_cl(m->pre2, y1, T, T, nu, 2, 1, x, ACT_NONE, 0.f, x, s);          // in-place residual: each element read once, then written
    const int chunk1 = streaming ? c.chunk : 0;
    for (auto& w : m->enc) conformer_layer(m, w, x, T, pe, chunk1, s, nu);
    // Upsample1D: nearest x2 -> left pad 4 -> Conv1d k5  (upsample_encoder.py:59-63); stacked utterances of equal length: output row 2 r + k <- row r
    const int T2 = 2 * T;
    hipLaunchKernelGGL(upsample2x_kernel, dim3(nblk(2 * R * d)), dim3(256), 0, s, x, xe, (int)R, d);
    conv_cl(m->upconv, xe, T2, T2, nu, 4, 1, n, ACT_NONE, 0.f, nullptr, s);
    lin_cl(m->up_embed_lin, n, 2 * R, xe, ACT_NONE, nullptr, s);
    ln_rows(m->up_embed_ln, xe, x, 2 * R, d, 1e-5f, s, ACT_NONE, xscale);
    hipLaunchKernelGGL(rel_pos_emb_kernel, dim3(2 * T2 - 1), dim3(256), 0, s, pe, T2, d);
    const int chunk2 = streaming ? 2 * c.chunk : 0;
    for (auto& w : m->enc_up) conformer_layer(m, w, x, T2, pe, chunk2, s, nu);
    ln_rows(m->after_norm, x, h_out, 2 * R, d, 1e-5f, s);
}

// ---- estimator ---------------------------------------------------------------------------------------------------------
static void drop_graphs(cv_flow* m) { std::lock_guard<std::recursive_mutex> lk(runtime_lock()); for (auto& g : m->graphs) (void)hipGraphExecDestroy(g.second); m->graphs.clear(); m->seen.clear(); m->graph_used.clear(); }

// nz = batch rows of the estimator: 2 (the CFG pair of one utterance) or 2 x the utterances of a batched solve
static void est_reserve(cv_flow* m, int T, int nz = 2) {
    if (m->cfg.estimator == 2) T += 2;    // the up-sampled stream is 2 ceil(T / 2) rows before it is cut to the skip's length (flow/decoder.py:275)
    if (T <= m->est_cap && nz <= m->est_nz) return;
    drop_graphs(m);                       // captured kernels hold the old workspace addresses
    T = std::max(T, m->est_cap); nz = std::max(nz, m->est_nz);
    const auto& c = m->cfg; const size_t R = (size_t)nz * T, C = c.est_ch, f = 4;
    m->s_in.ensure(R * 4 * c.mel * f); m->s_a.ensure(R * C * f); m->s_b.ensure(R * C * f); m->s_c.ensure(R * C * f); m->s_n.ensure(R * C * f);
    m->s_qkv.ensure(R * 3 * c.est_heads * 64 * f); m->s_att.ensure(R * c.est_heads * 64 * f); m->s_ff.ensure(R * 4 * C * f);
    m->s_skip.ensure(R * C * f); m->s_cat.ensure(R * 2 * C * f); m->s_out.ensure(R * c.mel * f);
    if (c.estimator == 2) {
        size_t nd = 0; for (const auto& u : m->ust) nd += u.kind == 0;
        for (size_t i = 0; i < nd; ++i) m->u_skip[i].ensure(R * C * f);
        m->u_gn.ensure((size_t)nz * 8 * 64 * sizeof(double));
    }
    {   // bf16 activations of the fused pipeline; V^T is [nz][heads * 64][pitch] and its never-written pad columns must stay finite (0 x P)
        const size_t inner = (size_t)c.est_heads * 64, pitch = (size_t)(T + T / 2 + 63) / 64 * 64;
        m->h_qk.ensure(R * 2 * inner * 2); m->h_att.ensure(R * inner * 2); m->h_ff.ensure(R * 4 * C * 2); if (!m->h_zero.p) { m->h_zero.ensure(64); CV_HIP(hipMemset(m->h_zero.p, 0, 64)); }
        m->h_xn.ensure(R * C * 2); m->h_cur.ensure(R * std::max((size_t)4 * c.mel, 2 * C) * 2);
        const size_t before = m->h_vt.bytes;
        m->h_vt.ensure((size_t)nz * inner * pitch * 2);
        if (m->h_vt.bytes != before || nz != m->est_nz) { CV_HIP(hipMemset(m->h_vt.p, 0, m->h_vt.bytes)); m->vt_pitch = (int)(m->h_vt.bytes / ((size_t)nz * inner * 2) / 64 * 64); }
    }
    m->est_cap = T; m->est_nz = nz;
}
static void time_reserve(cv_flow* m, int n) {
    if (n <= m->t_cap) return;
    drop_graphs(m);
    const auto& c = m->cfg; const size_t tdim = 4 * c.est_ch;
    m->t_val.ensure((size_t)n * 4); m->t_sin.ensure((size_t)n * 4 * c.mel * 4); m->t_h.ensure((size_t)n * tdim * 4); m->t_emb.ensure((size_t)n * tdim * 4);
    m->t_mlp.ensure(m->stages.size() * n * c.est_ch * 4);
    m->t_cap = n;
}
// t_val[n] (device) -> per-resnet time projections t_mlp[stage][n][C]  (SinusoidalPosEmb -> TimestepEmbedding -> Mish -> Linear)
static void time_embed(cv_flow* m, int n, hipStream_t s) {
    const auto& c = m->cfg; const int tdim = 4 * c.est_ch, cin = 4 * c.mel;
    hipLaunchKernelGGL(time_sinusoid_kernel, dim3(n), dim3(256), 0, s, m->t_val.as<float>(), m->t_sin.as<float>(), n, cin);
    lin_cl(m->time1, m->t_sin.as<float>(), n, m->t_h.as<float>(), ACT_SILU, nullptr, s);
    lin_cl(m->time2, m->t_h.as<float>(), n, m->t_emb.as<float>(), ACT_NONE, nullptr, s);
    for (size_t i = 0; i < m->stages.size(); ++i)
        lin_cl(m->stages[i].res.mlp, m->t_emb.as<float>(), n, m->t_mlp.as<float>() + i * (size_t)n * c.est_ch, ACT_NONE, nullptr, s, ACT_MISH);
    (void)tdim;
}


// ---- fused bf16 pipeline (flow_fused.h) --------------------------------------------------------------------------------
// out = act(LN(x) W^T + b) as bf16; columns >= n_row go to the transposed, key-permuted V^T
static void ln_gemm_bf16(const Lin& l, const LN* ln, float eps, const float* x, int M, int act, bf16_t* out, int ldo, int n_row,
                         bf16_t* outT, long long t_batch, int ldt, int rows_per_batch, hipStream_t s) {
    CV_CHECK(l.bf16 && l.K % 32 == 0 && l.K <= 256 && l.taps == 1 && l.N % 4 == 0, "ln_gemm_bf16: needs bf16 weights, K % 32 == 0, K <= 256");
    FlowGemmArgs a{};
    a.A = x; a.lda = l.K; a.gamma = ln ? ln->g : nullptr; a.beta = ln ? ln->b : nullptr; a.eps = eps;
    a.W = reinterpret_cast<const bf16_t*>(l.w); a.Kp = l.Kp; a.bias = l.b; a.M = M; a.N = l.N; a.K = l.K; a.act = act;
    a.out = out; a.ldo = ldo; a.n_row = n_row; a.outT = outT; a.t_batch = t_batch; a.ldt = ldt; a.rows_per_batch = rows_per_batch > 0 ? rows_per_batch : M;
    CV_CHECK(n_row % 16 == 0, "ln_gemm_bf16: the transposed section must start on a 16-column boundary");
    // Tile choice, measured on MI355X at the U10 size (M = 1348, N = 1536 / 1024, K = 256; profiles/r2_probe_flow_2_tiles.txt): per launch
    // 32x64 9.9 us, 64x64 13.9 us, 64x128 ~16 us, 64x192 22.6 us.  A workgroup of this kernel is one dependent chain (load burst -> LayerNorm ->
    // MFMAs -> epilogue) and a CU ingests only a few tens of bytes per cycle, so what helps is MANY small co-resident workgroups whose
    // chains overlap, not fewer re-reads; the big tiles stay selectable for other shapes / later pipelined variants.
    int tile = tl_flow_tile;
    if (tile == 0) tile = 3;
    auto grid = [&](int bm, int bn) { return dim3((unsigned)(((M + bm - 1) / bm) * ((l.N + bn - 1) / bn))); };
    if (tile == 1) hipLaunchKernelGGL((flow_gemm_kernel<64, 64, 1, 0>), grid(64, 64), dim3(256), 0, s, a);
    else if (tile == 2) hipLaunchKernelGGL((flow_gemm_kernel<64, 128, 1, 0>), grid(64, 128), dim3(256), 0, s, a);
    else if (tile == 3) {
        // two N tiles per workgroup when the single-tile grid would not fit the 768 resident workgroups (3 per CU) in one round but half of it does
        // (QKV at T = 674: 1032 -> 516 workgroups; FF1 has 688 and stays single-tile): the LayerNorm prologue is paid once per pair
        const unsigned g1 = grid(32, 64).x;
        if (tl_flow_ntile != 1 && (tl_flow_ntile == 2 || (g1 > 768 && g1 <= 1536)) && l.N % 128 == 0)
            hipLaunchKernelGGL((flow_gemm_kernel<32, 64, 1, 0, 2>), grid(32, 128), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((flow_gemm_kernel<32, 64, 1, 0>), grid(32, 64), dim3(256), 0, s, a);
    }
    else hipLaunchKernelGGL((flow_gemm_kernel<64, 192, 1, 0>), grid(64, 192), dim3(256), 0, s, a);
}
// C = A_bf16 W^T + b (+ res), fp32
static void gemm_bf16_res(const Lin& l, const bf16_t* A, int lda, int M, float* C, const float* res, hipStream_t s) {
    CV_CHECK(l.bf16 && l.K % 32 == 0 && l.taps == 1 && l.N % 4 == 0 && lda % 8 == 0, "gemm_bf16_res: needs bf16 weights, K % 32 == 0");
    FlowGemmArgs a{};
    a.A = A; a.lda = lda; a.W = reinterpret_cast<const bf16_t*>(l.w); a.Kp = l.Kp; a.bias = l.b; a.M = M; a.N = l.N; a.K = l.K;
    a.C = C; a.ldc = l.N; a.res = res; a.n_row = l.N;
    // 32 x 64 tiles; 32 x 32 (option "res_tile" = 1; the same k order per element: the same bits) doubles the workgroups of these N = 256 GEMMs (172 at M = 1348)
    if (tl_res_tile == 1 && l.N % 32 == 0) { hipLaunchKernelGGL((flow_gemm_kernel<32, 32, 0, 1>), dim3(((M + 31) / 32) * ((l.N + 31) / 32)), dim3(256), 0, s, a); return; }
    const unsigned g = ((M + 31) / 32) * ((l.N + 63) / 64);
    hipLaunchKernelGGL((flow_gemm_kernel<32, 64, 0, 1>), dim3(g), dim3(256), 0, s, a);
}
// ---- large-M forms (flow_big.h): LayerNorm -> bf16 once per row, then plain bf16 GEMMs on 128-wide tiles
static void ln_bf16(const LN& ln, float eps, const float* x, int M, int K, bf16_t* y, hipStream_t s) {
    CV_CHECK(K % 4 == 0 && K <= 256, "ln_bf16: K % 4 == 0, K <= 256");
    LnBf16Args a{x, K, ln.g, ln.b, eps, y, K, M, K};
    hipLaunchKernelGGL(ln_bf16_kernel, dim3((unsigned)((M + 15) / 16)), dim3(256), 0, s, a);
}
// persistent grid (flow_big.h): at most the workgroups that are resident at once - a workgroup then walks its run of tiles with the stage pipeline running across them
static unsigned big_grid(int M, int N, int bm, int bn) {
    const long long tiles = (long long)((M + bm - 1) / bm) * ((N + bn - 1) / bn);
    if (tl_big_persist < 0) return (unsigned)tiles;            // option big_persist = -1: one tile per workgroup (the first form, kept for A/B runs)
    const int lds = 2 * (bm + bn) * 32 * 4;
    const int per_cu = tl_big_persist > 0 ? tl_big_persist : std::max(1, std::min(160 * 1024 / lds, 4));       // 0: LDS-limited residency, at most 4 (16 waves) per CU
    const long long g = std::min<long long>(tiles, (long long)per_cu * 256);
    return (unsigned)(tl_big_grid_cap > 0 ? std::min<long long>(g, tl_big_grid_cap) : g);     // test hook: a handful of workgroups walk many tiles each
}
template <int OMODE, bool GLDS>
static void gemm_big_launch2(const FlowGemmArgs& a, int tile, hipStream_t s) {
    if (tile == 1) hipLaunchKernelGGL((flow_gemm_big_kernel<128, 128, OMODE, false, GLDS>), dim3(big_grid(a.M, a.N, 128, 128)), dim3(256), 0, s, a);
    else if (tile == 2) hipLaunchKernelGGL((flow_gemm_big_kernel<128, 64, OMODE, false, GLDS>), dim3(big_grid(a.M, a.N, 128, 64)), dim3(256), 0, s, a);
    else if (tile == 4) hipLaunchKernelGGL((flow_gemm_big_kernel<32, 64, OMODE, false, GLDS>), dim3(big_grid(a.M, a.N, 32, 64)), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((flow_gemm_big_kernel<64, 64, OMODE, false, GLDS>), dim3(big_grid(a.M, a.N, 64, 64)), dim3(256), 0, s, a);
}
template <int OMODE>
static void gemm_big_launch(const FlowGemmArgs& a, int tile, hipStream_t s) {
    if (tl_big_glds) gemm_big_launch2<OMODE, true>(a, tile, s); else gemm_big_launch2<OMODE, false>(a, tile, s);
}
// out = act(A W^T + b) as bf16 (columns >= n_row to the transposed, key-permuted V^T), A = bf16 rows (LayerNorm already applied)
static void gemm_big_bf16(const Lin& l, const bf16_t* A, int M, int act, bf16_t* out, int ldo, int n_row, bf16_t* outT, long long t_batch, int ldt, int rows_per_batch,
                          hipStream_t s) {
    CV_CHECK(l.bf16 && l.K % 32 == 0 && l.taps == 1 && l.N % 4 == 0 && n_row % 16 == 0, "gemm_big_bf16: needs bf16 weights, K % 32 == 0");
    FlowGemmArgs a{};
    a.A = A; a.lda = l.K; a.W = reinterpret_cast<const bf16_t*>(l.w); a.Kp = l.Kp; a.bias = l.b; a.M = M; a.N = l.N; a.K = l.K; a.act = act;
    a.out = out; a.ldo = ldo; a.n_row = n_row; a.outT = outT; a.t_batch = t_batch; a.ldt = ldt; a.rows_per_batch = rows_per_batch > 0 ? rows_per_batch : M;
    a.lds_epilogue = tl_big_lds_epi;
    if (tl_gemm_dbg && tl_gemm_dbg_which == (outT ? 1 : 2)) a.dbg = tl_gemm_dbg;
    gemm_big_launch<0>(a, tl_big_tile0 ? tl_big_tile0 : 3, s);
}
// C = A_bf16 W^T + b (+ res), fp32
static void gemm_big_res(const Lin& l, const bf16_t* A, int lda, int M, float* C, const float* res, hipStream_t s) {
    CV_CHECK(l.bf16 && l.K % 32 == 0 && l.taps == 1 && l.N % 4 == 0 && lda % 8 == 0, "gemm_big_res: needs bf16 weights, K % 32 == 0");
    FlowGemmArgs a{};
    a.A = A; a.lda = lda; a.W = reinterpret_cast<const bf16_t*>(l.w); a.Kp = l.Kp; a.bias = l.b; a.M = M; a.N = l.N; a.K = l.K;
    a.C = C; a.ldc = l.N; a.res = res; a.n_row = l.N; a.lds_epilogue = tl_big_lds_epi;
    if (tl_gemm_dbg && tl_gemm_dbg_which == (l.K <= 512 ? 3 : 4)) a.dbg = tl_gemm_dbg;
    gemm_big_launch<1>(a, tl_big_tile1 ? tl_big_tile1 : 3, s);
}
// causal Conv1d / Linear over bf16 rows of nz requests of T rows each (ResNet blocks of a large pass): C = conv(A) + b (+ res), fp32
template <bool GLDS>
static void conv_big_launch(const FlowGemmArgs& a, int tile, hipStream_t s) {
    if (tile == 1) hipLaunchKernelGGL((flow_gemm_big_kernel<128, 128, 1, true, GLDS>), dim3(big_grid(a.M, a.N, 128, 128)), dim3(256), 0, s, a);
    else if (tile == 2) hipLaunchKernelGGL((flow_gemm_big_kernel<128, 64, 1, true, GLDS>), dim3(big_grid(a.M, a.N, 128, 64)), dim3(256), 0, s, a);
    else if (tile == 4) hipLaunchKernelGGL((flow_gemm_big_kernel<32, 64, 1, true, GLDS>), dim3(big_grid(a.M, a.N, 32, 64)), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((flow_gemm_big_kernel<64, 64, 1, true, GLDS>), dim3(big_grid(a.M, a.N, 64, 64)), dim3(256), 0, s, a);
}
static void conv_big(const Lin& l, const bf16_t* A, int T, int nz, int pad_left, float* C, const float* res, const void* zeros, hipStream_t s) {
    CV_CHECK(l.bf16 && l.K % 64 == 0 && l.N % 4 == 0 && l.Kp == l.K, "conv_big: needs bf16 weights and K % 64 == 0");
    FlowGemmArgs a{};
    a.A = A; a.lda = l.K; a.W = reinterpret_cast<const bf16_t*>(l.w); a.Kp = l.Kp; a.bias = l.b; a.M = nz * T; a.N = l.N; a.K = l.K;
    a.C = C; a.ldc = l.N; a.res = res; a.n_row = l.N; a.taps = l.taps; a.pad_left = pad_left; a.rows_per_batch = T; a.zeros = zeros; a.lds_epilogue = tl_big_lds_epi;
    const int tile = tl_big_tile1 ? tl_big_tile1 : 3;
    if (tl_big_glds) conv_big_launch<true>(a, tile, s); else conv_big_launch<false>(a, tile, s);
}
// everything between the attention of block `t` and the QKV GEMM of the next block - or, with `q`, up to and including that GEMM - in one launch, 64 / 48 / 32 rows
// per workgroup (flow_band.h)
// Rows per band of a pass of M rows (every height computes the same bits): a band is one dependent chain of ~35 - 55 us, so what matters is how many ROUNDS of bands
// the chip runs (one workgroup per CU at every height: the pipelined 32-row form holds 96 KB of LDS) and how tall a band of the last round is.  Launch time of
// flow_band_kernel<.., HAS_QKV> by height and workgroup count k, from tools/ubench/bandq_probe (profiles/r5_band_qkv.txt section 4: after the staging arrays left
// scratch memory): t64 = 50 + 0.026 k, t48 = 41.5 + 0.015 k (pipelined), t32 = 30 + 0.03 k (pipelined).  8 utterances of U10 (10 784 rows) -> 48 rows (225 workgroups,
// one round), up to 8192 rows -> 32, 12 - 16 k rows -> 64.
static int band_rows_for(int M) {
    auto t = [](int bm, int k) { return bm == 64 ? 50.f + 0.026f * k : bm == 48 ? 41.5f + 0.015f * k : 30.f + 0.03f * k; };
    int best = 32; float best_t = 1e30f;
    for (int bm : {32, 48, 64}) {
        const int n = (M + bm - 1) / bm, cap = 256, full = n / cap, rem = n % cap;
        const float est = full * t(bm, cap) + (rem ? t(bm, rem) : 0.f);
        if (est < best_t) { best_t = est; best = bm; }
    }
    return best;
}
struct BandQkv { bf16_t* qk; int ld_qk; bf16_t* vt; long long vt_batch; int ldt; int rows_per_batch; };
template <int C, int INNER, int FF, int NW>
static void flow_band_launch(const FlowBandArgs& a, bool has_next, bool qkv, int bm, bool pipe, hipStream_t s) {
    const dim3 g((unsigned)((a.M + bm - 1) / bm)), b(NW * 64);
#define CV_BAND(BM_, PIPE_)                                                                                                                 \
    if (qkv) hipLaunchKernelGGL((flow_band_kernel<C, INNER, FF, true, NW, 0, BM_, true, PIPE_>), g, b, 0, s, a);                            \
    else if (has_next) hipLaunchKernelGGL((flow_band_kernel<C, INNER, FF, true, NW, 0, BM_, false, PIPE_>), g, b, 0, s, a);                 \
    else hipLaunchKernelGGL((flow_band_kernel<C, INNER, FF, false, NW, 0, BM_, false, PIPE_>), g, b, 0, s, a);
    if (bm == 64) { CV_BAND(64, false) }
    else if (bm == 48) { if (pipe) { CV_BAND(48, true) } else { CV_BAND(48, false) } }
    else { if (pipe) { CV_BAND(32, true) } else { CV_BAND(32, false) } }
#undef CV_BAND
}
static void flow_band(const cv_flow* m, const TBlockW& t, bool has_next, const BandQkv* q, const bf16_t* att, int inner, float* x, int C, int M, bf16_t* xn, hipStream_t s) {
    FlowBandArgs a{};
    a.att = att; a.ld_att = inner; a.x = x; a.ldx = C; a.wstream = q ? t.bandq : t.band; a.prm = t.tail_prm; a.eps = 1e-5f; a.M = M; a.xn = xn; a.ld_xn = C; a.stagger = m->band_stagger;
    CV_CHECK(t.band && t.tail_prm && (!has_next || t.tail_qkv) && (!q || (has_next && t.bandq)), "flow_band: block was not packed for this call");
    if (q) { a.qk = q->qk; a.ld_qk = q->ld_qk; a.vt = q->vt; a.vt_batch = q->vt_batch; a.ldt = q->ldt; a.rows_per_batch = q->rows_per_batch > 0 ? q->rows_per_batch : M; }
    const int bm = m->band_bm ? m->band_bm : band_rows_for(m->rule_rows > M ? (int)m->rule_rows : M);
    const bool pipe = bm == 48 ? m->band_pipe >= 1 : bm == 32 ? m->band_pipe >= 2 : false;
    if (C == 256 && inner == 512) flow_band_launch<256, 512, 1024, 8>(a, has_next, q != nullptr, bm, pipe, s);
    else if (C == 64 && inner == 64) flow_band_launch<64, 64, 256, 4>(a, has_next, q != nullptr, bm, pipe, s);
    else throw Error("flow_band: no instantiation for these dimensions");
}

// LayerNorm(norm1) + QKV GEMM of a stage's first block in one launch per row band (flow_lnqkv_kernel; the band height of the pass, like flow_band)
static void flow_lnqkv(const cv_flow* m, const TBlockW& t, const BandQkv& q, const float* x, int C, int inner, int M, hipStream_t s) {
    FlowLnQkvArgs a{};
    a.x = x; a.ldx = C; a.wstream = t.lnqkv; a.gamma = t.norm1.g; a.beta = t.norm1.b; a.eps = 1e-5f; a.M = M;
    a.qk = q.qk; a.ld_qk = q.ld_qk; a.vt = q.vt; a.vt_batch = q.vt_batch; a.ldt = q.ldt; a.rows_per_batch = q.rows_per_batch > 0 ? q.rows_per_batch : M;
    CV_CHECK(t.lnqkv && t.norm1.g && t.norm1.b, "flow_lnqkv: block was not packed for this call");
    const int bm = m->band_bm ? m->band_bm : band_rows_for(m->rule_rows > M ? (int)m->rule_rows : M);
    const dim3 g((unsigned)((M + bm - 1) / bm));
    if (C == 256 && inner == 512) {
        if (bm == 64) hipLaunchKernelGGL((flow_lnqkv_kernel<256, 512, 8, 64>), g, dim3(512), 0, s, a);
        else if (bm == 48) hipLaunchKernelGGL((flow_lnqkv_kernel<256, 512, 8, 48>), g, dim3(512), 0, s, a);
        else hipLaunchKernelGGL((flow_lnqkv_kernel<256, 512, 8, 32>), g, dim3(512), 0, s, a);
    } else if (C == 64 && inner == 64) {
        if (bm == 64) hipLaunchKernelGGL((flow_lnqkv_kernel<64, 64, 4, 64>), g, dim3(256), 0, s, a);
        else if (bm == 48) hipLaunchKernelGGL((flow_lnqkv_kernel<64, 64, 4, 48>), g, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((flow_lnqkv_kernel<64, 64, 4, 32>), g, dim3(256), 0, s, a);
    } else throw Error("flow_lnqkv: no instantiation for these dimensions");
}

#ifdef CV_BUILD_EXPERIMENTS
// everything after the attention of block `t` (+ LayerNorm and QKV of `next`) in one launch, 16 rows per workgroup (flow_tail.h)
static void flow_tail(const TBlockW& t, const TBlockW* next, const bf16_t* att, int inner, float* x, int C, int M, bf16_t* qk, bf16_t* vt, long long vt_batch, int ldt,
                      int rows_per_batch, int depth, hipStream_t s) {
    FlowTailArgs a{};
    a.att = att; a.ld_att = inner; a.x = x; a.ldx = C; a.wstream = t.tail;
    a.prm = t.tail_prm; a.eps = 1e-5f; a.M = M;
    CV_CHECK(t.tail && t.tail_prm && t.tail_qkv == (next != nullptr), "flow_tail: block was not packed for this call");
    if (next) { a.qk = qk; a.ld_qk = 2 * inner; a.vt = vt; a.vt_batch = vt_batch; a.ldt = ldt; a.rows_per_batch = rows_per_batch; }
    const dim3 g((unsigned)((M + 15) / 16));
    if (C == 256 && inner == 512) {
        if (depth == 16) { if (next) hipLaunchKernelGGL((flow_tail_kernel<256, 512, 1024, true, 16>), g, dim3(256), 0, s, a); else hipLaunchKernelGGL((flow_tail_kernel<256, 512, 1024, false, 16>), g, dim3(256), 0, s, a); }
        else { if (next) hipLaunchKernelGGL((flow_tail_kernel<256, 512, 1024, true, 8>), g, dim3(256), 0, s, a); else hipLaunchKernelGGL((flow_tail_kernel<256, 512, 1024, false, 8>), g, dim3(256), 0, s, a); }
    } else if (C == 64 && inner == 64) {
        if (next) hipLaunchKernelGGL((flow_tail_kernel<64, 64, 256, true, 8>), g, dim3(256), 0, s, a); else hipLaunchKernelGGL((flow_tail_kernel<64, 64, 256, false, 8>), g, dim3(256), 0, s, a);
    } else throw Error("flow_tail: no instantiation for these dimensions");
}
#endif
static void attn_flow(const bf16_t* qk, int ld, int inner, const bf16_t* vt, long long vt_batch, int ldt, bf16_t* o, int B, int H, int T, int chunk, hipStream_t s, const int* klen = nullptr,
                      bool big = false) {
    AttnFlowArgs a{};
    a.klen = klen; a.dbg = tl_attn_dbg;
    a.q = qk; a.k = qk + inner; a.ld = ld; a.vt = vt; a.vt_batch = vt_batch; a.ldt = ldt; a.o = o; a.ldo = inner;
    a.B = B; a.H = H; a.T = T; a.scale = 0.125f; a.mask_mode = chunk > 0 ? MASK_CHUNK : MASK_NONE; a.chunk = chunk;
    const dim3 g2((unsigned)(((T + 31) / 32) * H * B)), g4((unsigned)(((T + 63) / 64) * H * B));
    if (tl_attn32) {
        // 128 queries per workgroup (4 waves).  64 (2 waves; option "attn32_waves" = 2) fill more CUs for a single utterance (176 instead of 96 workgroups at T = 674) and
        // measured SLOWER: 39.2 vs 37.5 ms per flow.inference (tools/probe_flow_r6b.py) - each K / V tile then serves half the queries.  Same bits either way.
        const unsigned g128 = (unsigned)(((T + 127) / 128) * H * B);
        const int nw = tl_attn32_waves == 2 ? 2 : 4;
        if (nw == 2) hipLaunchKernelGGL((attn_flow32_kernel<2, 3>), dim3((unsigned)(((T + 63) / 64) * H * B)), dim3(128), 0, s, a);
        else hipLaunchKernelGGL((attn_flow32_kernel<4, 3>), dim3(g128), dim3(256), 0, s, a);
        return;
    }
#ifdef CV_BUILD_EXPERIMENTS      // attn_flow_kernel (flow_fused.h), the attention of rounds 2-5 in its workgroup shapes: superseded by attn_flow32_kernel, kept for A/B builds (option attn32 = 0)
    // 128-query workgroups (QG = 2): the same arithmetic per query as attn_flow_kernel<4, 2, 2>, so only that default may be replaced
    if (big && tl_attn_ks == 2) { hipLaunchKernelGGL((attn_flow_kernel<4, 2, 2, 2>), dim3((unsigned)(((T + 127) / 128) * H * B)), dim3(512), 0, s, a); return; }
    if (tl_attn_ks == 2) { hipLaunchKernelGGL((attn_flow_kernel<4, 2, 2>), g4, dim3(512), 0, s, a); return; }
    // 3 / 4 key splits: 12 / 16 waves on the same 64 queries, 192 / 256 keys per iteration - 4 / 3 iterations instead of 6 at T = 674 (round 3 probe)
    if (tl_attn_ks == 3) { hipLaunchKernelGGL((attn_flow_kernel<4, 3, 3>), g4, dim3(768), 0, s, a); return; }
    if (tl_attn_ks == 4) { hipLaunchKernelGGL((attn_flow_kernel<4, 4, 4>), g4, dim3(1024), 0, s, a); return; }
    if (tl_attn_waves == 2) {
        if (tl_attn_kt == 2) hipLaunchKernelGGL((attn_flow_kernel<2, 2>), g2, dim3(128), 0, s, a);
        else hipLaunchKernelGGL((attn_flow_kernel<2, 1>), g2, dim3(128), 0, s, a);
    } else {
        if (tl_attn_kt == 2) hipLaunchKernelGGL((attn_flow_kernel<4, 2>), g4, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((attn_flow_kernel<4, 1>), g4, dim3(256), 0, s, a);
    }
#else
    (void)big;
    throw Error("flow: attn32 = 0 (attn_flow_kernel) needs a library built with CV_BUILD_EXPERIMENTS");
#endif
}

// s_in: packed [2][T][4*mel]; t_row: which row of the time tables; t_shared: both CFG rows use the same row;
// result in s_out [2][T][mel] (not masked).  Buffer discipline: a stage never writes the buffer it reads its input from
// (res_conv re-reads the stage input after block1/block2), outputs ping-pong between s_a and s_c, s_b is scratch.
// nz batch rows starting at batch row b0 of the packed workspaces (b0 > 0: the second half of a two-stream evaluation, estimator_eval)
// The transformer blocks of one U-Net stage on the residual stream x [nz][T][C] (matcha BasicTransformerBlock x n_blocks; flow/decoder.py:455-466 causal, 232-243 CosyVoice-300M):
// rows r0 .. of every workspace (r0 = b0 * T: the second launch chain of a two-chain evaluation works behind the first one's rows).
static void stage_blocks(cv_flow* m, const StageW& st, float* x, int T, int nz, int b0, long long r0, int chunk, const int* klen, bool fused, hipStream_t s) {
    const auto& c = m->cfg; const int C = c.est_ch, H = c.est_heads, inner = H * 64; const long long R = (long long)nz * T;
    float* n = m->s_n.as<float>() + r0 * C; float* qkv = m->s_qkv.as<float>() + r0 * 3 * inner;
    float* att = m->s_att.as<float>() + r0 * inner; float* ff = m->s_ff.as<float>() + r0 * 4 * C;
    for (size_t ti = 0; ti < st.tf.size(); ++ti) {      // matcha BasicTransformerBlock (self-attention + exact-erf GELU feed-forward)
        const TBlockW& t = st.tf[ti];
#ifdef CV_BUILD_EXPERIMENTS
        if (fused && m->fused_tail && t.tail) {   // flow_tail.h: LN + QKV once per stage, then attention + ONE row-band launch per block
            const long long vt_batch = (long long)inner * m->vt_pitch;
            bf16_t* qk = m->h_qk.as<bf16_t>() + r0 * 2 * inner; bf16_t* vt = m->h_vt.as<bf16_t>() + b0 * vt_batch; bf16_t* ab = m->h_att.as<bf16_t>() + r0 * inner;
            if (ti == 0) ln_gemm_bf16(t.qkv, &t.norm1, 1e-5f, x, (int)R, ACT_NONE, qk, 2 * inner, 2 * inner, vt, vt_batch, m->vt_pitch, T, s);
            attn_flow(qk, 2 * inner, inner, vt, vt_batch, m->vt_pitch, ab, nz, H, T, chunk, s, klen);
            flow_tail(t, ti + 1 < st.tf.size() ? &st.tf[ti + 1] : nullptr, ab, inner, x, C, (int)R, qk, vt, vt_batch, m->vt_pitch, T, m->tail_ring, s);
            continue;
        }
#endif
        if (fused) {                      // flow_fused.h: 5 launches, bf16 activations, same rounding points as the path below
            const long long vt_batch = (long long)inner * m->vt_pitch;
            bf16_t* qk = m->h_qk.as<bf16_t>() + r0 * 2 * inner; bf16_t* vt = m->h_vt.as<bf16_t>() + b0 * vt_batch; bf16_t* ab = m->h_att.as<bf16_t>() + r0 * inner;
            bf16_t* fb = m->h_ff.as<bf16_t>() + r0 * 4 * C;
            const bool big = m->big_rows > 0 && R >= m->big_rows, big_attn = m->attn2_rows > 0 && R >= m->attn2_rows;
            if (big && m->fused_band && t.band) {     // flow_band.h: QKV GEMM, attention, then ONE launch per 64-row band up to the next block's LayerNorm; bit-identical to the forms below
                bf16_t* xn = m->h_xn.as<bf16_t>() + r0 * C;
                const bool had_qkv = ti > 0 && m->band_qkv && st.tf[ti - 1].bandq;  // later blocks: the previous block's band launch left this block's Q | K and V^T (band_qkv), or LayerNorm(norm1) of its output in xn
                if (ti == 0 && m->band_qkv && m->ln_qkv && t.lnqkv) {                // first block of a stage: LayerNorm + QKV GEMM in ONE launch per row band (round 6; the bits of the two launches below)
                    const BandQkv bq0{qk, 2 * inner, vt, vt_batch, m->vt_pitch, T};
                    flow_lnqkv(m, t, bq0, x, C, inner, (int)R, s);
                } else {
                    if (ti == 0) ln_bf16(t.norm1, 1e-5f, x, (int)R, C, xn, s);
                    if (!had_qkv) gemm_big_bf16(t.qkv, xn, (int)R, ACT_NONE, qk, 2 * inner, 2 * inner, vt, vt_batch, m->vt_pitch, T, s);
                }
                attn_flow(qk, 2 * inner, inner, vt, vt_batch, m->vt_pitch, ab, nz, H, T, chunk, s, klen, big_attn);
                const bool has_next = ti + 1 < st.tf.size();
                const BandQkv bq{qk, 2 * inner, vt, vt_batch, m->vt_pitch, T};
                flow_band(m, t, has_next, has_next && m->band_qkv && t.bandq ? &bq : nullptr, ab, inner, x, C, (int)R, xn, s);
                continue;
            }
            if (big) {                    // flow_big.h: 7 launches of large tiles, bit-identical to the 5 below
                bf16_t* xn = m->h_xn.as<bf16_t>() + r0 * C;
                ln_bf16(t.norm1, 1e-5f, x, (int)R, C, xn, s);
                gemm_big_bf16(t.qkv, xn, (int)R, ACT_NONE, qk, 2 * inner, 2 * inner, vt, vt_batch, m->vt_pitch, T, s);
                attn_flow(qk, 2 * inner, inner, vt, vt_batch, m->vt_pitch, ab, nz, H, T, chunk, s, klen, big_attn);
                gemm_big_res(t.out, ab, inner, (int)R, x, x, s);
                ln_bf16(t.norm3, 1e-5f, x, (int)R, C, xn, s);
                gemm_big_bf16(t.ff1, xn, (int)R, ACT_GELU_ERF, fb, 4 * C, 4 * C, nullptr, 0, 0, 0, s);
                gemm_big_res(t.ff2, fb, 4 * C, (int)R, x, x, s);
                continue;
            }
            ln_gemm_bf16(t.qkv, &t.norm1, 1e-5f, x, (int)R, ACT_NONE, qk, 2 * inner, 2 * inner, vt, vt_batch, m->vt_pitch, T, s);
            attn_flow(qk, 2 * inner, inner, vt, vt_batch, m->vt_pitch, ab, nz, H, T, chunk, s, klen, big_attn);
            gemm_bf16_res(t.out, ab, inner, (int)R, x, x, s);
            ln_gemm_bf16(t.ff1, &t.norm3, 1e-5f, x, (int)R, ACT_GELU_ERF, fb, 4 * C, 4 * C, nullptr, 0, 0, 0, s);
            gemm_bf16_res(t.ff2, fb, 4 * C, (int)R, x, x, s);
            continue;
        }
        ln_rows(t.norm1, x, n, R, C, 1e-5f, s);
        lin_cl(t.qkv, n, R, qkv, ACT_NONE, nullptr, s);
        AttnArgs at{};
        at.q = qkv; at.q_batch = (long long)T * 3 * inner; at.q_row = 3 * inner; at.q_head = 64;
        at.k = qkv + inner; at.k_batch = at.q_batch; at.k_row = 3 * inner; at.k_head = 64;
        at.v = qkv + 2 * inner; at.v_batch = at.q_batch; at.v_row = 3 * inner; at.v_head = 64;
        at.o = att; at.o_batch = (long long)T * inner; at.o_row = inner; at.o_head = 64;
        at.B = nz; at.H = H; at.kv_group = 1; at.Tq = T; at.Tk = T; at.scale = 0.125f;
        at.mask_mode = chunk > 0 ? MASK_CHUNK : MASK_NONE; at.chunk = chunk; at.rel_bd = nullptr;
        at.bf16 = tl_bf16_mfma; at.klen = klen;
        attention(at, s);
        lin_cl(t.out, att, R, x, ACT_NONE, x, s);
        ln_rows(t.norm3, x, n, R, C, 1e-5f, s);
        lin_cl(t.ff1, n, R, ff, ACT_GELU_ERF, nullptr, s);
        lin_cl(t.ff2, ff, R, x, ACT_NONE, x, s);
    }
}

static void estimator_forward(cv_flow* m, int T, int t_row, int t_rows_total, bool t_shared, int streaming, hipStream_t s, int nz = 2, int b0 = 0) {
    const auto& c = m->cfg; const int C = c.est_ch; const long long R = (long long)nz * T;
    const long long r0 = (long long)b0 * T;              // first row of this call in every [batch rows x T][...] workspace
    float* pp[2] = {m->s_a.as<float>() + r0 * C, m->s_c.as<float>() + r0 * C};
    float* xb = m->s_b.as<float>() + r0 * C; float* skip = m->s_skip.as<float>() + r0 * C;
    float* cat = m->s_cat.as<float>() + r0 * 2 * C;
    const int* klen = m->cur_klen ? m->cur_klen + b0 : nullptr;
    const int chunk = streaming ? 2 * c.chunk : 0;
    const float* cur = m->s_in.as<float>() + r0 * 4 * c.mel;
    int flip = 0;
    const int nst = (int)m->stages.size();
    for (int si = 0; si < nst; ++si) {
        const StageW& st = m->stages[si];
        const float* tm = m->t_mlp.as<float>() + ((size_t)si * t_rows_total + t_row + (t_shared ? 0 : b0)) * C;
        const long long rpb = t_shared ? R : T;          // rows per time-embedding row
        float* x = pp[flip];                             // stage output (cur never aliases it)
        // CausalResnetBlock1D (decoder.py:65-85 + matcha ResnetBlock1D): block1 -> + mlp(t) -> block2 -> + res_conv(x)
        const bool fused = tl_bf16_mfma && m->fused && C <= 256 && m->wbf16;
        const int din = st.res.conv1.K;
        if (fused && m->big_rows > 0 && R >= m->big_rows && din % 64 == 0 && C % 64 == 0) {
            // large pass (flow_big.h): the block's input and Mish(LN(block1)) are rounded to bf16 ONCE (the small bf16 tiles round them per tap and per N tile
            // when they stage them - same values), the convolutions run on 128-row tiles: bit-identical to the five launches below
            // (the chain's share of h_cur starts at the WIDEST stage input's pitch: with a per-stage pitch r0 * din the second chain of a two-chain evaluation, a stage
            // behind or ahead of the first, would sit inside the first chain's rows of a wider stage - found in round 5 as a changed mixed64 hash under two lanes)
            bf16_t* cb = m->h_cur.as<bf16_t>() + r0 * std::max(4 * c.mel, 2 * C); bf16_t* hb = m->h_xn.as<bf16_t>() + r0 * C;
            hipLaunchKernelGGL(cvt_bf16_kernel, dim3(nblk(R * din / 8)), dim3(256), 0, s, cur, cb, R * din / 8);
            conv_big(st.res.conv1, cb, T, nz, 2, x, nullptr, m->h_zero.p, s);
            { NormArgs na{x, nullptr, R, C, st.res.ln1.g, st.res.ln1.b, 1e-5f, 0, ACT_MISH, 1.f, nullptr, tm, rpb}; na.y16 = hb; norm_rows(na, s); }
            conv_big(st.res.conv2, hb, T, nz, 2, x, nullptr, m->h_zero.p, s);
            ln_rows(st.res.ln2, x, xb, R, C, 1e-5f, s, ACT_MISH);
            conv_big(st.res.res, cb, T, nz, 0, x, xb, m->h_zero.p, s);
        } else {
        conv_cl(st.res.conv1, cur, T, T, nz, 2, 1, x, ACT_NONE, 0.f, nullptr, s);
        ln_rows(st.res.ln1, x, xb, R, C, 1e-5f, s, ACT_MISH, 1.f, tm, rpb);
        conv_cl(st.res.conv2, xb, T, T, nz, 2, 1, x, ACT_NONE, 0.f, nullptr, s);
        ln_rows(st.res.ln2, x, xb, R, C, 1e-5f, s, ACT_MISH);
        conv_cl(st.res.res, cur, T, T, nz, 0, 1, x, ACT_NONE, 0.f, xb, s);           // x = res_conv(input) + h
        }
        stage_blocks(m, st, x, T, nz, b0, r0, chunk, klen, fused, s);
        flip ^= 1;
        if (si == 0) {                         // keep the skip, then the stride-1 "downsample" CausalConv1d (decoder.py:452-453)
            CV_HIP(hipMemcpyAsync(skip, x, (size_t)R * C * 4, hipMemcpyDeviceToDevice, s));
            conv_cl(m->down_conv, x, T, T, nz, 2, 1, pp[flip], ACT_NONE, 0.f, nullptr, s);
            cur = pp[flip]; flip ^= 1;
        } else if (si == nst - 2) {            // last mid block: concat with the skip for the up block (decoder.py:476)
            hipLaunchKernelGGL(concat_cols_kernel, dim3(nblk(R * 2 * C)), dim3(256), 0, s, x, C, skip, C, cat, R);
            cur = cat;
        } else if (si == nst - 1) {            // up block's trailing CausalConv1d (decoder.py:490)
            conv_cl(m->up_conv, x, T, T, nz, 2, 1, pp[flip], ACT_NONE, 0.f, nullptr, s);
            cur = pp[flip]; flip ^= 1;
        } else {
            cur = x;
        }
    }
    // final_block (CausalBlock1D) + final_proj (decoder.py:492-494)
    float* y = pp[flip];
    conv_cl(m->final_conv, cur, T, T, nz, 2, 1, y, ACT_NONE, 0.f, nullptr, s);
    ln_rows(m->final_ln, y, xb, R, C, 1e-5f, s, ACT_MISH);
    conv_cl(m->final_proj, xb, T, T, nz, 0, 1, m->s_out.as<float>() + r0 * c.mel, ACT_NONE, 0.f, nullptr, s);
}

// ---- ConditionalDecoder of CosyVoice-300M (cfg.estimator == 2; flow/decoder.py:88-291 over matcha's Block1D / ResnetBlock1D / Downsample1D / Upsample1D) ------------
// The same stage as above with three differences: the convolutions are centred (pad 1 on both sides), a Block1D normalises with GroupNorm(8) over (channels of the
// group x TIME) instead of a LayerNorm per frame, and the stream changes resolution (stride-2 convolution down, transposed convolution up, cut to the skip's length).
// The transformer blocks are the ones of the causal U-Net: stage_blocks() with full attention - in bf16 mode the fused kernels of flow_fused.h / flow_big.h / flow_band.h.
// Batch-1 requests: the mask is all ones.
static void unet1_forward(cv_flow* m, int T, int t_row, int t_rows_total, bool t_shared, hipStream_t s, int nz = 2) {
    const auto& c = m->cfg; const int C = c.est_ch, G = 8;
    float* pp[2] = {m->s_a.as<float>(), m->s_c.as<float>()};
    float* xb = m->s_b.as<float>(); float* cat = m->s_cat.as<float>(); double* gn = m->u_gn.as<double>();
    const bool fused = tl_bf16_mfma && m->fused && C <= 256 && m->wbf16;
    auto conv = [&](const Lin& l, const float* in, int Tn, float* out, const float* res) { conv_cl(l, in, Tn, Tn, nz, l.taps / 2, 1, out, ACT_NONE, 0.f, res, s); };
    const float* cur = m->s_in.as<float>();
    int Tc = T, flip = 0, level = 0;
    std::vector<int> skip_t;
    for (size_t si = 0; si < m->stages.size(); ++si) {
        const StageW& st = m->stages[si]; const UStageW& u = m->ust[si];
        if (u.kind == 2) {                   // x[:, :, :skip length] ++ skip along the channels (decoder.py:275-276)
            const int Ts = skip_t.back(); skip_t.pop_back(); --level;
            hipLaunchKernelGGL(concat_cols_batched_kernel, dim3(nblk((long long)nz * Ts * 2 * C)), dim3(256), 0, s, cur, C, (long long)Tc * C, m->u_skip[level].as<float>(), C,
                               (long long)Ts * C, cat, Ts, nz);
            cur = cat; Tc = Ts;
        }
        const long long R = (long long)nz * Tc;
        const float* tm = m->t_mlp.as<float>() + ((size_t)si * t_rows_total + t_row) * C;
        float* x = pp[flip];                 // stage output (cur never aliases it)
        // ResnetBlock1D (matcha decoder.py): block1 -> + mlp(t) -> block2 -> + res_conv(x)
        conv(st.res.conv1, cur, Tc, x, nullptr);
        group_norm(x, xb, nz, Tc, C, G, st.res.ln1.g, st.res.ln1.b, 1e-5f, ACT_MISH, tm, t_shared ? 0 : C, gn, s);
        conv(st.res.conv2, xb, Tc, x, nullptr);
        group_norm(x, xb, nz, Tc, C, G, st.res.ln2.g, st.res.ln2.b, 1e-5f, ACT_MISH, nullptr, 0, gn, s);
        conv(st.res.res, cur, Tc, x, xb);
        stage_blocks(m, st, x, Tc, nz, 0, 0, 0, nullptr, fused, s);
        flip ^= 1;
        if (u.kind == 0) { CV_HIP(hipMemcpyAsync(m->u_skip[level].p, x, (size_t)R * C * 4, hipMemcpyDeviceToDevice, s)); skip_t.push_back(Tc); ++level; }
        if (u.post == 0) { cur = x; continue; }
        float* y = pp[flip];
        const Lin& l = u.post_lin;
        if (u.post == 1) conv(l, x, Tc, y, nullptr);
        else {
            GemmConvArgs a{};
            a.A = x; a.a_batch = (long long)Tc * C; a.a_len = (long long)Tc * C; a.K = l.K; a.taps = l.taps; a.pro = ACT_NONE; a.W = l.w; a.Kp = l.Kp; a.bias = l.b;
            a.C = y; a.N = l.N; a.act = ACT_NONE; a.out_scale = 1.f; a.a_bf16 = tl_bf16_mfma && l.bf16;
            int Tn;
            if (u.post == 2) {               // Downsample1D: output row j = rows 2 j - 1 .. 2 j + 1, contiguous in a channel-last sequence
                Tn = (Tc - 1) / 2 + 1;
                a.lda = 2 * C; a.a_off0 = -C; a.tap_step = 0; a.M = Tn; a.ldc = l.N; a.c_batch = (long long)Tn * l.N; a.c_len = a.c_batch;
            } else {                         // Upsample1D: GEMM row j = output rows 2 j - 1, 2 j (phase r of the kernel in columns r C .. r C + C), tap q reads row j - q
                Tn = 2 * Tc;
                a.lda = C; a.a_off0 = 0; a.tap_step = -C; a.M = Tc + 1; a.ldc = l.N; a.c_off = -C; a.c_batch = (long long)Tn * C; a.c_len = a.c_batch;
            }
            a.res_batch = a.c_batch; a.row_scale_batch = a.M;
            gemm_conv(a, l.bf16, nz, s);
            Tc = Tn;
        }
        cur = y; flip ^= 1;
    }
    CV_CHECK(Tc == T && skip_t.empty(), "flow(unet1): the up path must come back to the input length");
    float* y = pp[flip];
    conv(m->final_conv, cur, T, y, nullptr);
    group_norm(y, xb, nz, T, C, G, m->final_ln.g, m->final_ln.b, 1e-5f, ACT_MISH, nullptr, 0, gn, s);
    conv(m->final_proj, xb, T, m->s_out.as<float>(), nullptr);
}

// One estimator evaluation over nz batch rows: with est_streams = 2 (and an even nz) the two halves of the batch rows run as independent launch chains
// on `s` and on the handle's side stream, forked and joined through events (graph edges when `s` is being captured).
static void estimator_eval(cv_flow* m, int T, int t_row, int t_rows_total, bool t_shared, int streaming, hipStream_t s, int nz = 2) {
    if (m->cfg.estimator == 2) { unet1_forward(m, T, t_row, t_rows_total, t_shared, s, nz); return; }
    if ((m->est_streams < 2 && !m->two_chains_now) || nz < 2 || (nz & 1)) { estimator_forward(m, T, t_row, t_rows_total, t_shared, streaming, s, nz, 0); return; }
    if (!m->side_stream) {
        CV_HIP(hipStreamCreateWithFlags(&m->side_stream, hipStreamNonBlocking));
        CV_HIP(hipEventCreateWithFlags(&m->ev_fork, hipEventDisableTiming)); CV_HIP(hipEventCreateWithFlags(&m->ev_join, hipEventDisableTiming));
    }
    CV_HIP(hipEventRecord(m->ev_fork, s)); CV_HIP(hipStreamWaitEvent(m->side_stream, m->ev_fork, 0));
    m->rule_rows = (long long)nz * T;
    try {
        estimator_forward(m, T, t_row, t_rows_total, t_shared, streaming, s, nz / 2, 0);
        estimator_forward(m, T, t_row, t_rows_total, t_shared, streaming, m->side_stream, nz / 2, nz / 2);
    } catch (...) { m->rule_rows = 0; throw; }
    m->rule_rows = 0;
    CV_HIP(hipEventRecord(m->ev_join, m->side_stream)); CV_HIP(hipStreamWaitEvent(s, m->ev_join, 0));
}


// ---- DiT estimator (Fun-CosyVoice3) ---------------------------------------------------------------------------------------
// head-0 rotary of x_transformers as the reference applies it (flow/DiT/modules.py:363-373: on the un-split projection with 64-dim freqs):
// channels 0..63 of q and of k turn as interleaved pairs by angle pos * 10000^(-2i/64); every other head carries no rotary signal
static __global__ __launch_bounds__(256) void dit_rope_kernel(float* qkv, long long R, int T, int ld, int k_off) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= R * 64) return;
    const long long row = i >> 6; const int j = (int)(i & 63), which = j >> 5, pr = j & 31;
    const float inv = 1.0f / powf(10000.f, (float)(2 * pr) / 64.f);
    const float ang = (float)(row % T) * inv, c = cosf(ang), s = sinf(ang);
    float* p = qkv + row * ld + (which ? k_off : 0) + 2 * pr;
    const float a = p[0], b = p[1];
    p[0] = a * c - b * s; p[1] = b * c + a * s;
}
// h [n][T'][mel] -> out [2 T'][mel] (repeat_interleave(2) along time, flow/flow.py:391)
static __global__ __launch_bounds__(256) void repeat2_rows_kernel(const float* x, float* y, long long rows, int C) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= 2 * rows * C) return;
    y[i] = x[(i / C / 2) * C + i % C];
}

static void dit_reserve(cv_flow* m, int T, int nz = 2) {
    if (T <= m->est_cap && nz <= m->est_nz && m->d_x.p) return;
    T = std::max(T, m->est_cap); nz = std::max(nz, m->est_nz);
    const auto& c = m->cfg; const size_t R = (size_t)nz * T, D = c.est_ch, f = 4;
    drop_graphs(m);
    m->s_in.ensure(R * 4 * c.mel * f); m->s_out.ensure(R * c.mel * f);
    m->d_x.ensure(R * D * f); m->d_y.ensure(R * D * f); m->d_n.ensure(R * D * f); m->d_qkv.ensure(R * 3 * D * f); m->d_att.ensure(R * D * f);
    m->d_ff.ensure(R * c.est_mid * D * f);
    m->est_cap = T; m->est_nz = nz;
}
static void dit_time_reserve(cv_flow* m, int n) {
    if (n <= m->t_cap && m->d_mod.p) return;
    drop_graphs(m);
    const auto& c = m->cfg; const size_t D = c.est_ch;
    m->t_val.ensure((size_t)n * 4); m->d_tsin.ensure((size_t)n * 256 * 4); m->d_th.ensure((size_t)n * D * 4); m->d_temb.ensure((size_t)n * D * 4);
    m->d_mod.ensure((size_t)c.est_blocks * n * 6 * D * 4); m->d_fm.ensure((size_t)n * 2 * D * 4);
    m->t_cap = n;
}
// t_val[n] -> time embedding (modules.py:606-616) -> adaLN modulations of every block and of the final norm (SiLU -> Linear, :238-241,:266-268);
// the "+1" of (1 + scale) is folded into the bias of the scale chunks by the weight repacker
static void dit_time_embed(cv_flow* m, int n, hipStream_t s) {
    const auto& c = m->cfg; const int D = c.est_ch;
    hipLaunchKernelGGL(time_sinusoid_kernel, dim3(n), dim3(256), 0, s, m->t_val.as<float>(), m->d_tsin.as<float>(), n, 256);
    lin_cl(m->d_time1, m->d_tsin.as<float>(), n, m->d_th.as<float>(), ACT_SILU, nullptr, s);
    lin_cl(m->d_time2, m->d_th.as<float>(), n, m->d_temb.as<float>(), ACT_NONE, nullptr, s);
    for (int i = 0; i < c.est_blocks; ++i)
        lin_cl(m->dit[i].mod, m->d_temb.as<float>(), n, m->d_mod.as<float>() + (size_t)i * n * 6 * D, ACT_NONE, nullptr, s, ACT_SILU);
    lin_cl(m->d_fmod, m->d_temb.as<float>(), n, m->d_fm.as<float>(), ACT_NONE, nullptr, s, ACT_SILU);
}
// grouped causal Conv1d(D, D, k = 31, groups = 16) + Mish of one request (T rows): one batched implicit GEMM, batch = group
static void dit_group_conv(const Lin& l, const float* A, float* C, const float* res, int T, int D, hipStream_t s) {
    const int G = 16, gc = D / G;
    GemmConvArgs a{};
    a.A = A; a.a_batch = gc; a.a_len = (long long)(T - 1) * D + gc; a.lda = D; a.a_off0 = -(l.taps - 1) * D; a.tap_step = D; a.taps = l.taps; a.K = gc;
    a.pro = ACT_NONE; a.W = l.w; a.Kp = l.Kp; a.ldw = 0; a.w_batch = (long long)gc * l.taps * l.Kp; a.bias = l.b; a.bias_batch = gc;
    a.C = C; a.c_batch = gc; a.c_len = (long long)(T - 1) * D + gc; a.ldc = D; a.c_off = 0; a.M = T; a.N = gc;
    a.act = ACT_MISH; a.res = res; a.res_batch = gc; a.out_scale = 1.f; a.row_scale = nullptr; a.accumulate = 0;
    a.a_bf16 = tl_bf16_mfma && l.bf16;
    gemm_conv(a, l.bf16, G, s);
}
// Linear with the adaLN-zero gate:  C = res + gate[m / rows_per_gate] * (A W^T + b)
static void lin_gated(const Lin& l, const float* A, long long rows, float* C, const float* res, const float* gate, long long rows_per_gate, long long gate_stride,
                      hipStream_t s) {
    GemmConvArgs a{};
    a.A = A; a.a_batch = 0; a.a_len = rows * l.K; a.lda = l.K; a.a_off0 = 0; a.tap_step = 0; a.taps = 1; a.K = l.K; a.pro = ACT_NONE;
    a.W = l.w; a.Kp = l.Kp; a.ldw = 0; a.w_batch = 0; a.bias = l.b;
    a.C = C; a.c_batch = 0; a.c_len = rows * l.N; a.ldc = l.N; a.c_off = 0; a.M = (int)rows; a.N = l.N;
    a.act = ACT_NONE; a.res = res; a.res_batch = 0; a.out_scale = 1.f; a.row_scale = nullptr; a.accumulate = 0;
    a.col_scale = gate; a.col_scale_rows = (int)rows_per_gate; a.col_scale_stride = gate_stride;
    a.a_bf16 = tl_bf16_mfma && l.bf16;
    gemm_conv(a, l.bf16, 1, s);
}
// s_in packed [nz][T][4 mel] as [x | mu | spks | cond] (the in_proj columns are permuted to this order by the repacker) -> s_out [nz][T][mel];
// nz = 2 (conditional | unconditional) x the utterances solved together: rows never mix outside their own attention / position-conv window
static void dit_forward(cv_flow* m, int T, int t_row, int t_rows_total, bool t_shared, int streaming, hipStream_t s, int nz = 2) {
    const auto& c = m->cfg; const int D = c.est_ch, H = c.est_heads; const long long R = (long long)nz * T;
    CV_CHECK(t_shared || nz == 2, "dit_forward: per-row time steps are the two-row estimator call only");
    float* x = m->d_x.as<float>(); float* y = m->d_y.as<float>(); float* n = m->d_n.as<float>(); float* qkv = m->d_qkv.as<float>();
    float* att = m->d_att.as<float>(); float* ff = m->d_ff.as<float>();
    const int chunk = streaming ? 2 * c.chunk : 0;
    const long long rpb = t_shared ? R : T, gbs = t_shared ? 0 : 6LL * D;          // rows per modulation row; stride between the two rows' modulations
    // InputEmbedding (dit.py:76-98): proj, then x + CausalConvPositionEmbedding(x)
    lin_cl(m->d_inproj, m->s_in.as<float>(), R, x, ACT_NONE, nullptr, s);
    for (int b = 0; b < nz; ++b) {
        dit_group_conv(m->d_pos1, x + (size_t)b * T * D, y + (size_t)b * T * D, nullptr, T, D, s);
        dit_group_conv(m->d_pos2, y + (size_t)b * T * D, n + (size_t)b * T * D, x + (size_t)b * T * D, T, D, s);     // n = Mish(conv2) + x
    }
    float* cur = n; float* other = x;                                               // residual stream ping-pongs so no GEMM reads what it writes
    for (int i = 0; i < c.est_blocks; ++i) {
        const DitBlockW& w = m->dit[i];
        const float* mod = m->d_mod.as<float>() + ((size_t)i * t_rows_total + t_row) * 6 * D;     // [shift_msa | 1+scale_msa | gate_msa | shift_mlp | 1+scale_mlp | gate_mlp]
        NormArgs na{cur, y, R, D, mod + D, mod, 1e-6f, 0, ACT_NONE, 1.f, nullptr, nullptr, rpb}; na.gb_batch = gbs;
        norm_rows(na, s);
        lin_cl(w.qkv, y, R, qkv, ACT_NONE, nullptr, s);
        hipLaunchKernelGGL(dit_rope_kernel, dim3(nblk(R * 64)), dim3(256), 0, s, qkv, R, T, 3 * D, D);
        AttnArgs at{};
        at.q = qkv; at.q_batch = (long long)T * 3 * D; at.q_row = 3 * D; at.q_head = 64;
        at.k = qkv + D; at.k_batch = at.q_batch; at.k_row = 3 * D; at.k_head = 64;
        at.v = qkv + 2 * D; at.v_batch = at.q_batch; at.v_row = 3 * D; at.v_head = 64;
        at.o = att; at.o_batch = (long long)T * D; at.o_row = D; at.o_head = 64;
        at.B = nz; at.H = H; at.kv_group = 1; at.Tq = T; at.Tk = T; at.scale = 0.125f;
        at.mask_mode = chunk > 0 ? MASK_CHUNK : MASK_NONE; at.chunk = chunk; at.rel_bd = nullptr; at.bf16 = tl_bf16_mfma; at.klen = m->cur_klen;
        attention(at, s);
        lin_gated(w.out, att, R, other, cur, mod + 2 * D, rpb, gbs, s);              // x + gate_msa * to_out(attn)
        std::swap(cur, other);
        NormArgs nb{cur, y, R, D, mod + 4 * D, mod + 3 * D, 1e-6f, 0, ACT_NONE, 1.f, nullptr, nullptr, rpb}; nb.gb_batch = gbs;
        norm_rows(nb, s);
        lin_cl(w.ff1, y, R, ff, ACT_GELU_TANH, nullptr, s);
        lin_gated(w.ff2, ff, R, other, cur, mod + 5 * D, rpb, gbs, s);               // x + gate_mlp * ff(...)
        std::swap(cur, other);
    }
    const float* fm = m->d_fm.as<float>() + (size_t)t_row * 2 * D;                  // [1 + scale | shift]
    NormArgs nf{cur, y, R, D, fm, fm + D, 1e-6f, 0, ACT_NONE, 1.f, nullptr, nullptr, rpb}; nf.gb_batch = t_shared ? 0 : 2LL * D;
    norm_rows(nf, s);
    lin_cl(m->d_proj, y, R, m->s_out.as<float>(), ACT_NONE, nullptr, s);
}
// front of CausalMaskedDiffWithDiT.inference (flow/flow.py:381-392): PreLookaheadLayer on the token embeddings, then repeat_interleave(2) = mu
static void dit_front(cv_flow* m, const float* tok_emb, int n_enc, const float* ctx, float* mu_out, hipStream_t s) {
    const auto& c = m->cfg; const int d = c.dim, la = c.pre_lookahead, Cp = c.ffn;
    m->e_xe.ensure((size_t)(n_enc + la) * d * 4); m->e_x2.ensure((size_t)n_enc * Cp * 4); m->e_x.ensure((size_t)n_enc * d * 4);
    float* xe = m->e_xe.as<float>(); float* y1 = m->e_x2.as<float>(); float* h = m->e_x.as<float>();
    CV_HIP(hipMemcpyAsync(xe, tok_emb, (size_t)n_enc * d * 4, hipMemcpyDeviceToDevice, s));
    if (ctx) CV_HIP(hipMemcpyAsync(xe + (size_t)n_enc * d, ctx, (size_t)la * d * 4, hipMemcpyDeviceToDevice, s));
    else CV_HIP(hipMemsetAsync(xe + (size_t)n_enc * d, 0, (size_t)la * d * 4, s));
    conv_cl(m->d_pre1, xe, n_enc + la, n_enc, 1, 0, 1, y1, ACT_LEAKY, 0.01f, nullptr, s);
    conv_cl(m->d_pre2, y1, n_enc, n_enc, 1, 2, 1, h, ACT_NONE, 0.f, tok_emb, s);     // + inputs (residual)
    hipLaunchKernelGGL(repeat2_rows_kernel, dim3(nblk(2LL * n_enc * d)), dim3(256), 0, s, h, mu_out, (long long)n_enc, d);
}

// ---- solve_euler + inference -------------------------------------------------------------------------------------------
// nu utterances of the same length solved together (x, mu, cond [nu][T][mel], spk [nu][mel]): the estimator sees 2 * nu batch rows, every GEMM
// of a step runs once over all of them (same arithmetic per row as nu separate solves: rows never mix outside their own attention / conv window)
static void solve_euler(cv_flow* m, float* x /*[nu][T][mel] in/out*/, const float* mu, const float* spk, const float* cond, int T,
                        int n_steps, int streaming, hipStream_t s, int nu = 1) {
    const auto& c = m->cfg;
    const bool dit = c.estimator == 1;
    if (dit) { dit_reserve(m, T, 2 * nu); dit_time_reserve(m, n_steps); } else { est_reserve(m, T, 2 * nu); time_reserve(m, n_steps); }
    // cosine schedule and the t / dt recurrences of solve_euler, in fp32 like torch (flow_matching.py:89-122, 223-226)
    std::vector<float> span(n_steps + 1), tv(n_steps), dts(n_steps);
    for (int i = 0; i <= n_steps; ++i) {
        const float lin = n_steps == 0 ? 0.f : (float)i / (float)n_steps;      // torch.linspace(0, 1, n+1)
        span[i] = 1.f - cosf(lin * 0.5f * 3.14159265358979323846f);
    }
    float t = span[0], dt = span[1] - span[0];
    for (int st = 1; st <= n_steps; ++st) {
        tv[st - 1] = t; dts[st - 1] = dt;
        t = t + dt;
        if (st < n_steps) dt = span[st + 1] - t;
    }
    m->host_t = tv;                           // kept alive for the async copy
    CV_HIP(hipMemcpyAsync(m->t_val.p, m->host_t.data(), (size_t)n_steps * 4, hipMemcpyHostToDevice, s));
    const long long n = (long long)nu * T * c.mel;
    auto body = [&]() {
        if (dit) dit_time_embed(m, n_steps, s); else time_embed(m, n_steps, s);
        for (int st = 0; st < n_steps; ++st) {
            hipLaunchKernelGGL(pack_est_input_kernel, dim3(nblk(2 * n * 4)), dim3(256), 0, s, x, mu, spk, cond, m->s_in.as<float>(), T, c.mel, 1, nu);
            if (dit) dit_forward(m, T, st, n_steps, true, streaming, s, 2 * nu); else estimator_eval(m, T, st, n_steps, true, streaming, s, 2 * nu);
            hipLaunchKernelGGL(cfg_euler_kernel, dim3(nblk(n)), dim3(256), 0, s, x, m->s_out.as<float>(), n, dts[st], c.cfg_rate);
        }
    };
    const auto key = std::make_tuple(T, n_steps, (streaming ? 1 : 0) + 2 * nu + (m->cur_klen ? 64 : 0));     // a padded batch bakes the klen pointer into its launches
    // Graphs pay where the launches are short next to the host's ~4 us per launch (one utterance: 3640 launches of ~10 us).  A shared pass over several
    // utterances is GPU-bound launch by launch - the host stays ahead - while capturing and instantiating its ~5000 nodes costs tens of milliseconds per new
    // shape, and offline batches bring a new Tmax with almost every group: passes of at least graph_max_rows estimator rows run eager.
    const bool graphable = m->use_graph && (m->graph_max_rows <= 0 || 2LL * nu * T < m->graph_max_rows);
    auto it = m->graphs.find(key);
    if (graphable && it != m->graphs.end() && x == m->f_x.as<float>()) {
        m->graph_used[key] = ++m->graph_clock;
        { std::lock_guard<std::recursive_mutex> lk(runtime_lock()); CV_HIP(hipGraphLaunch(it->second, s)); }
        return;
    }
    // buffers may have been re-allocated by *_reserve since a capture: graphs are dropped whenever a workspace grows (see est_reserve)
    if (graphable && x == m->f_x.as<float>() && ++m->seen[key] == 2) {
        std::lock_guard<std::recursive_mutex> lk(runtime_lock());
        if (m->graphs.size() >= m->graph_cap) {          // evict the least recently used shape; it may be captured again later (its sighting count restarts)
            auto victim = m->graphs.begin();
            for (auto g = m->graphs.begin(); g != m->graphs.end(); ++g) if (m->graph_used[g->first] < m->graph_used[victim->first]) victim = g;
            (void)hipGraphExecDestroy(victim->second);
            m->seen.erase(victim->first); m->graph_used.erase(victim->first); m->graphs.erase(victim);
        }
        hipGraphExec_t ge = nullptr;
        hipGraph_t g = capture_graph(s, body);
        CV_HIP(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CV_HIP(hipGraphDestroy(g));
        m->graphs[key] = ge; m->graph_used[key] = ++m->graph_clock; ++m->graph_captures;
        CV_HIP(hipGraphLaunch(ge, s));
        return;
    }
    // an eager pass of a large shape: its batch rows as two launch chains (eager_streams); a first-sighting eager run of a graphable shape stays one chain
    m->two_chains_now = !graphable && !dit && m->eager_streams >= 2;
    try { body(); } catch (...) { m->two_chains_now = false; throw; }
    m->two_chains_now = false;
}

extern "C" {

int cv_flow_create(cv_flow** out, const cv_flow_config* cfg) {
    return guarded([&] { CV_CHECK(out && cfg, "cv_flow_create: null argument"); auto* m = new cv_flow(); m->cfg = *cfg; *out = m; });
}
int cv_flow_set_tensor(cv_flow* m, const char* name, const void* dev_ptr, int32_t dtype, int64_t numel) {
    return guarded([&] { CV_CHECK(m, "null handle"); m->tm.set(name, dev_ptr, dtype, numel); });
}
int cv_flow_finalize(cv_flow* m) { return guarded([&] { CV_CHECK(m, "null handle"); flow_finalize(m); }); }
int cv_flow_set_option(cv_flow* m, const char* name, int32_t value) {
    return guarded([&] {
        CV_CHECK(m && name, "null argument");
        if (std::string(name) == "use_graph") { m->use_graph = value != 0; drop_graphs(m); }
        else if (std::string(name) == "bf16_mfma") { m->bf16_mfma = value != 0; drop_graphs(m); }      // captured graphs bake the kernel choice
        else if (std::string(name) == "flow_tile") { CV_CHECK(value >= 0 && value <= 4, "flow_tile must be 0..4"); m->flow_tile = value; drop_graphs(m); }
        else if (std::string(name) == "attn32") { need_experiments(value == 0, "flow option attn32 = 0"); m->attn32 = value != 0; drop_graphs(m); }
        else if (std::string(name) == "res_tile") { CV_CHECK(value == 0 || value == 1, "res_tile must be 0 or 1"); m->res_tile = value; drop_graphs(m); }
        else if (std::string(name) == "attn32_waves") { CV_CHECK(value == 0 || value == 2 || value == 4, "attn32_waves must be 0, 2 or 4"); m->attn32_waves = value; drop_graphs(m); }
        else if (std::string(name) == "attn_ks") { CV_CHECK(value >= 1 && value <= 4, "attn_ks must be 1 .. 4"); m->attn_ks = value; drop_graphs(m); }
        else if (std::string(name) == "attn_kt") { CV_CHECK(value == 1 || value == 2, "attn_kt must be 1 or 2"); m->attn_kt = value; drop_graphs(m); }
        else if (std::string(name) == "attn_waves") { CV_CHECK(value == 2 || value == 4, "attn_waves must be 2 or 4"); m->attn_waves = value; drop_graphs(m); }
        else if (std::string(name) == "eager_streams") { CV_CHECK(value == 1 || value == 2, "eager_streams must be 1 or 2"); need_experiments(value == 2, "flow option eager_streams = 2"); m->eager_streams = value; }
        else if (std::string(name) == "est_streams") { CV_CHECK(value == 1 || value == 2, "est_streams must be 1 or 2"); m->est_streams = value; drop_graphs(m); }
        else if (std::string(name) == "graph_max_rows") { CV_CHECK(value >= 0, "graph_max_rows must be >= 0"); m->graph_max_rows = value; drop_graphs(m); }
        else if (std::string(name) == "graph_cap") { CV_CHECK(value >= 1 && value <= 256, "graph_cap must be 1 .. 256"); drop_graphs(m); m->graph_cap = (size_t)value; }
        else if (std::string(name) == "fused_tail") { need_experiments(value != 0, "flow option fused_tail"); m->fused_tail = value != 0; drop_graphs(m); }
        else if (std::string(name) == "band_qkv") { m->band_qkv = value != 0; drop_graphs(m); }
        else if (std::string(name) == "ln_qkv") { m->ln_qkv = value != 0; drop_graphs(m); }
        else if (std::string(name) == "band_stagger") { CV_CHECK(value >= 0 && value <= 64, "band_stagger must be 0 .. 64"); need_experiments(value != 0, "flow option band_stagger"); m->band_stagger = value; drop_graphs(m); }
        else if (std::string(name) == "band_pipe") { CV_CHECK(value >= 0 && value <= 2, "band_pipe must be 0, 1 or 2"); m->band_pipe = value; drop_graphs(m); }
        else if (std::string(name) == "band_bm") { CV_CHECK(value == 0 || value == 32 || value == 48 || value == 64, "band_bm must be 0, 32, 48 or 64"); m->band_bm = value; drop_graphs(m); }
        else if (std::string(name) == "fused_band") { m->fused_band = value != 0; drop_graphs(m); }      // bf16 mode, large passes: one 64-row band launch between attention and the next QKV GEMM (flow_band.h) on / off
        else if (std::string(name) == "flow_ntile") { CV_CHECK(value >= 0 && value <= 2, "flow_ntile must be 0, 1 or 2"); m->flow_ntile = value; drop_graphs(m); }
        else if (std::string(name) == "tail_ring") { m->tail_ring = value == 16 ? 16 : 8; drop_graphs(m); }      // bf16 mode: one row-band launch after each attention (flow_tail.h) on / off
        else if (std::string(name) == "big_rows") { CV_CHECK(value >= 0, "big_rows must be >= 0"); m->big_rows = value; drop_graphs(m); }
        else if (std::string(name) == "attn2_rows") { CV_CHECK(value >= 0, "attn2_rows must be >= 0"); m->attn2_rows = value; drop_graphs(m); }
        else if (std::string(name) == "big_tile0") { CV_CHECK(value >= 0 && value <= 4, "big_tile0 must be 0..4"); m->big_tile0 = value; drop_graphs(m); }
        else if (std::string(name) == "big_persist") { CV_CHECK(value >= -1 && value <= 8, "big_persist must be -1..8"); m->big_persist = value; drop_graphs(m); }
        else if (std::string(name) == "gemm_dbg") { m->gemm_dbg_on = value; if (value) { m->gemm_dbg.ensure((size_t)65536 * 8 * 8); CV_HIP(hipMemset(m->gemm_dbg.p, 0, m->gemm_dbg.bytes)); } drop_graphs(m); }
        else if (std::string(name) == "attn_dbg") { m->attn_dbg_on = value != 0; if (value) { m->attn_dbg.ensure((size_t)65536 * 8 * 8); CV_HIP(hipMemset(m->attn_dbg.p, 0, m->attn_dbg.bytes)); } drop_graphs(m); }
        else if (std::string(name) == "big_lds_epi") { m->big_lds_epi = value != 0; drop_graphs(m); }
        else if (std::string(name) == "enc_batch") m->enc_batch = value != 0;
        else if (std::string(name) == "big_glds") { m->big_glds = value != 0; drop_graphs(m); }
        else if (std::string(name) == "big_grid_cap") { CV_CHECK(value >= 0, "big_grid_cap must be >= 0"); m->big_grid_cap = value; drop_graphs(m); }
        else if (std::string(name) == "big_tile1") { CV_CHECK(value >= 0 && value <= 4, "big_tile1 must be 0..4"); m->big_tile1 = value; drop_graphs(m); }
        else if (std::string(name) == "fused") { m->fused = value != 0; drop_graphs(m); }              // bf16 mode: fused transformer blocks (flow_fused.h) on / off
        else throw Error(std::string("unknown option ") + name);
    });
}
int cv_flow_get_stat(cv_flow* m, const char* name, int64_t* value) {
    return guarded([&] {
        CV_CHECK(m && name && value, "null argument");
        if (std::string(name) == "graph_captures") *value = m->graph_captures;
        else if (std::string(name) == "graphs_cached") *value = (int64_t)m->graphs.size();
        else if (std::string(name).rfind("attn_phase_", 0) == 0 || std::string(name).rfind("gemm_phase_", 0) == 0) {
            // dev tool: mean shader clocks between stamps k and k + 1 of the LAST stamped launch over its workgroups (up to 65536); k = 7: workgroups,
            // k = 8: first start -> last end, k = 9: mean start offset of a workgroup after the first one, k = 10: mean end offset
            const bool at = name[0] == 'a';
            CV_CHECK(at ? (m->attn_dbg_on && m->attn_dbg.p) : (m->gemm_dbg_on && m->gemm_dbg.p), "<attn|gemm>_phase_<k>: set option attn_dbg / gemm_dbg first");
            CV_HIP(hipDeviceSynchronize());
            std::vector<long long> h(65536 * 8);
            CV_HIP(hipMemcpy(h.data(), at ? m->attn_dbg.p : m->gemm_dbg.p, h.size() * 8, hipMemcpyDeviceToHost));
            const int k = atoi(name + 11), last = 4;
            long long t0 = 0, tend = 0, n = 0; double acc = 0.0, acc_start = 0.0, acc_end = 0.0;
            for (int w = 0; w < 65536; ++w) { if (!h[(size_t)w * 8 + last]) continue; if (!n || h[(size_t)w * 8] < t0) t0 = h[(size_t)w * 8]; tend = std::max(tend, h[(size_t)w * 8 + last]); ++n; }
            for (int w = 0; w < 65536; ++w) {
                if (!h[(size_t)w * 8 + last]) continue;
                acc_start += (double)(h[(size_t)w * 8] - t0); acc_end += (double)(h[(size_t)w * 8 + last] - t0);
                if (k >= 0 && k < 4) acc += (double)(h[(size_t)w * 8 + k + 1] - h[(size_t)w * 8 + k]);
            }
            *value = n == 0 ? 0 : k == 8 ? tend - t0 : k == 9 ? (long long)(acc_start / n) : k == 10 ? (long long)(acc_end / n) : k == 7 ? n : (long long)(acc / n);
        }
        else throw Error(std::string("unknown statistic ") + name);
    });
}
void cv_flow_destroy(cv_flow* m) {
    if (!m) return;
    // may be called from a garbage-collector finaliser on ANY thread while another thread drives a different handle: quiesce the
    // device and hold the runtime lock so stream / graph / buffer destruction never overlaps a capture, a launch burst or a realloc
    std::lock_guard<std::recursive_mutex> lk(runtime_lock());
    (void)hipDeviceSynchronize();
    delete m;
}

int cv_flow_encoder(cv_flow* m, const float* tok_emb, int32_t n_tok, const float* context, int32_t streaming, float* h_out, void* stream) {
    return guarded([&] { CV_CHECK(m && m->finalized && tok_emb && h_out, "cv_flow_encoder: bad arguments");
                         CV_CHECK(m->cfg.estimator == 0, "cv_flow_encoder: CausalMaskedDiffWithDiT has no conformer encoder (its front is PreLookaheadLayer, inside cv_flow_inference)");
                         PrecisionScope prec(m);
                         flow_encoder(m, tok_emb, n_tok, context, streaming, h_out, as_stream(stream)); });
}

static void flow_estimator_call(cv_flow* m, const float* x, const float* mask, const int32_t* key_len, const float* mu, const float* t, const float* spks,
                                const float* cond, int32_t T, int32_t streaming, float* out, void* stream) {
    CV_CHECK(m && m->finalized && x && mu && t && spks && cond && out && T > 0, "cv_flow_estimator: bad arguments");
    PrecisionScope prec(m);
    hipStream_t s = as_stream(stream);
    const auto& c = m->cfg;
    const bool dit = c.estimator == 1;
    if (dit) { dit_reserve(m, T); dit_time_reserve(m, 2); } else { est_reserve(m, T); time_reserve(m, 2); }
    if (key_len) {                         // padded mask: the rows' key counts, read by the attention kernels (AttnArgs::klen) - what a padded pass of several utterances uses
        CV_CHECK(mask && key_len[0] >= 1 && key_len[0] <= T && key_len[1] >= 1 && key_len[1] <= T, "cv_flow_estimator_masked: key_len must be 1..T and needs the mask");
        m->klen.ensure(64 * sizeof(int));
        m->host_klen.assign(key_len, key_len + 2);
        CV_HIP(hipMemcpyAsync(m->klen.p, m->host_klen.data(), 2 * sizeof(int), hipMemcpyHostToDevice, s));
        m->cur_klen = m->klen.as<int>();
    }
    try {
        CV_HIP(hipMemcpyAsync(m->t_val.p, t, 8, hipMemcpyDeviceToDevice, s));
        if (dit) dit_time_embed(m, 2, s); else time_embed(m, 2, s);
        hipLaunchKernelGGL(pack_est_input_kernel, dim3(nblk(2LL * T * 4 * c.mel)), dim3(256), 0, s, x, mu, spks, cond, m->s_in.as<float>(), T, c.mel, 0, 1);
        if (dit) dit_forward(m, T, 0, 2, false, streaming, s); else estimator_eval(m, T, 0, 2, false, streaming, s);
    } catch (...) { m->cur_klen = nullptr; throw; }
    m->cur_klen = nullptr;
    // without key_len `mask` must be all ones for the in-kernel (index-computed) attention masks to be exact; either way it is applied to the output
    hipLaunchKernelGGL(to_channel_first_kernel, dim3(nblk(2LL * T * c.mel)), dim3(256), 0, s, m->s_out.as<float>(), out, 2, T, c.mel, 0, mask);
}
// Measurement hook of bench.py's MFMA roofline record: the three launches a transformer block of a LARGE pass is made of (QKV GEMM, flash attention, the 64-row band
// launch of flow_band.h), each as `reps` back-to-back launches of stage 1 / block 0's weights between one HIP-event pair on `stream`, on the workspaces of a pass
// over nz estimator batch rows of T frames (contents: whatever the last pass left - the timings do not depend on the values).  us3: microseconds per launch.
static void flow_profile_block(cv_flow* m, int nz, int T, int reps, float* us3, hipStream_t s) {
    const auto& c = m->cfg; const int C = c.est_ch, H = c.est_heads, inner = H * 64; const long long R = (long long)nz * T;
    CV_CHECK(m->finalized && c.estimator == 0 && m->wbf16 && nz >= 2 && T > 0 && reps > 0 && us3, "cv_flow_profile_block: needs the bf16-mode U-Net estimator");
    CV_CHECK(m->stages.size() >= 2 && m->stages[1].tf.size() >= 2 && m->stages[1].tf[0].band, "cv_flow_profile_block: no band stream for this configuration");
    PrecisionScope prec(m);
    est_reserve(m, T, nz);
    const TBlockW& t = m->stages[1].tf[0];
    const long long vt_batch = (long long)inner * m->vt_pitch;
    bf16_t* qk = m->h_qk.as<bf16_t>(); bf16_t* vt = m->h_vt.as<bf16_t>(); bf16_t* ab = m->h_att.as<bf16_t>(); bf16_t* xn = m->h_xn.as<bf16_t>(); float* x = m->s_a.as<float>();
    CV_HIP(hipMemsetAsync(x, 0, (size_t)R * C * 4, s)); CV_HIP(hipMemsetAsync(xn, 0, (size_t)R * C * 2, s));
    hipEvent_t e0, e1; CV_HIP(hipEventCreate(&e0)); CV_HIP(hipEventCreate(&e1));
    for (int which = 0; which < 3; ++which) {
        float best = 1e30f;
        for (int pass = 0; pass < 4; ++pass) {               // pass 0: warm-up (freshly reserved workspaces: first touch); then the best of three
            CV_HIP(hipEventRecord(e0, s));
            for (int i = 0; i < reps; ++i) {
                if (which == 0) gemm_big_bf16(t.qkv, xn, (int)R, ACT_NONE, qk, 2 * inner, 2 * inner, vt, vt_batch, m->vt_pitch, T, s);
                else if (which == 1) attn_flow(qk, 2 * inner, inner, vt, vt_batch, m->vt_pitch, ab, nz, H, T, 0, s, nullptr, m->attn2_rows > 0 && R >= m->attn2_rows);
                else { const BandQkv bq{qk, 2 * inner, vt, vt_batch, m->vt_pitch, T}; flow_band(m, t, true, m->band_qkv && t.bandq ? &bq : nullptr, ab, inner, x, C, (int)R, xn, s); }      // as a pass runs it: with the next block's QKV GEMM when band_qkv is on
            }
            CV_HIP(hipEventRecord(e1, s)); CV_HIP(hipEventSynchronize(e1));
            float ms = 0.f; CV_HIP(hipEventElapsedTime(&ms, e0, e1));
            if (pass > 0) best = std::min(best, 1e3f * ms / reps);
            us3[which] = best;
        }
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
}
int cv_flow_profile_block(cv_flow* m, int32_t nz, int32_t T, int32_t reps, float* us3, void* stream) {
    return guarded([&] { CV_CHECK(m, "cv_flow_profile_block: null handle"); flow_profile_block(m, nz, T, reps, us3, as_stream(stream)); });
}
int cv_flow_estimator(cv_flow* m, const float* x, const float* mask, const float* mu, const float* t, const float* spks, const float* cond,
                      int32_t T, int32_t streaming, float* out, void* stream) {
    return guarded([&] { flow_estimator_call(m, x, mask, nullptr, mu, t, spks, cond, T, streaming, out, stream); });
}
int cv_flow_estimator_masked(cv_flow* m, const float* x, const float* mask, const int32_t* key_len, const float* mu, const float* t, const float* spks,
                             const float* cond, int32_t T, int32_t streaming, float* out, void* stream) {
    return guarded([&] { flow_estimator_call(m, x, mask, key_len, mu, t, spks, cond, T, streaming, out, stream); });
}

// nu utterances with the SAME token count and prompt length through one flow pass: x-vector projection, token embedding and encoder per utterance
// (the encoder is ~8 % of the stage), the CFM Euler solve over all of them at once.  token_ids [nu][n_tok], prompt_feat [nu][mel_len1][mel],
// embedding [nu][spk_dim], mel_out [nu][mel][T - mel_len1].  nu = 1 is cv_flow_inference.
static void flow_inference(cv_flow* m, int nu, const int32_t* token_ids, int n_tok, const float* prompt_feat, int mel_len1, const float* embedding,
                           const float* noise_cl, int streaming, int finalize, int n_timesteps, float* mel_out, int32_t* mel_len2_out, void* stream) {
    PrecisionScope prec(m);
    const auto& c = m->cfg; const int d = c.dim;
    hipStream_t s = resolve(m, stream);
    void* stream_r = reinterpret_cast<void*>(s);
    const int n_enc = finalize ? n_tok : n_tok - c.pre_lookahead;
    CV_CHECK(n_enc > 0 && n_timesteps > 0, "cv_flow_inference: too few tokens");
    const int T = 2 * n_enc, mel_len2 = T - mel_len1;
    CV_CHECK(mel_len2 > 0 && mel_len1 >= 0, "cv_flow_inference: prompt longer than the sequence");
    CV_CHECK(nu >= 1 && nu <= 8, "cv_flow_inference_batch: 1..8 utterances");
    if (n_tok > m->inf_cap || (long long)n_tok * nu > m->inf_rows) {        // per-utterance buffers follow n_tok, the stacked ones n_tok * nu
        drop_graphs(m);
        const size_t tk = (size_t)std::max(n_tok, m->inf_cap), rows = std::max((size_t)n_tok * nu, (size_t)m->inf_rows);
        m->f_tok.ensure(rows * d * 4); m->f_h.ensure(2 * rows * d * 4);          // stacked: the encoder takes all utterances of a pass at once
        m->f_mu.ensure(2 * rows * c.mel * 4); m->f_cond.ensure(2 * rows * c.mel * 4); m->f_x.ensure(2 * rows * c.mel * 4);
        m->f_spk.ensure((size_t)8 * c.mel * 4); m->f_spkn.ensure((size_t)c.spk_dim * 4);
        m->inf_cap = (int)tk; m->inf_rows = (long long)rows;
    }
    const size_t per = (size_t)T * c.mel;
    // the conformer encoder over all utterances at once (rows stacked; its attention per utterance): same arithmetic per row as the loop below, 1/nu of the launches
    const bool enc_together = nu > 1 && c.estimator != 1 && m->enc_batch;
    if (enc_together) {
        CV_CHECK(cv_gather_rows(m->input_embedding, m->wbf16 ? CV_BF16 : CV_F32, c.vocab, d, token_ids, (long long)nu * n_tok, m->f_tok.as<float>(), 1.f, stream_r) == 0, cv_last_error());
        const float* ctx = finalize ? nullptr : m->f_tok.as<float>() + (size_t)n_enc * d;
        flow_encoder(m, m->f_tok.as<float>(), n_enc, ctx, streaming, m->f_h.as<float>(), s, nu, n_tok);
        lin_cl(m->enc_proj, m->f_h.as<float>(), (long long)nu * T, m->f_mu.as<float>(), ACT_NONE, nullptr, s);
    }
    for (int u = 0; u < nu; ++u) {
        // x-vector: F.normalize -> Linear(192, 80)  (flow.py:248-249)
        hipLaunchKernelGGL(l2_normalize_kernel, dim3(1), dim3(256), 0, s, embedding + (size_t)u * c.spk_dim, m->f_spkn.as<float>(), c.spk_dim);
        lin_cl(m->spk_affine, m->f_spkn.as<float>(), 1, m->f_spk.as<float>() + (size_t)u * c.mel, ACT_NONE, nullptr, s);
        if (!enc_together) {
            // token embedding (mask is all ones for batch 1, flow.py:252-254); gather_rows_kernel lives in llm_kernels.h -> reuse via C ABI
            CV_CHECK(cv_gather_rows(m->input_embedding, m->wbf16 ? CV_BF16 : CV_F32, c.vocab, d, token_ids + (size_t)u * n_tok, n_tok, m->f_tok.as<float>(), 1.f, stream_r) == 0, cv_last_error());
            const float* ctx = finalize ? nullptr : m->f_tok.as<float>() + (size_t)n_enc * d;
            float* mu = m->f_mu.as<float>() + u * per;
            if (c.estimator == 1) dit_front(m, m->f_tok.as<float>(), n_enc, ctx, mu, s);
            else {
                flow_encoder(m, m->f_tok.as<float>(), n_enc, ctx, streaming, m->f_h.as<float>(), s);
                lin_cl(m->enc_proj, m->f_h.as<float>(), T, mu, ACT_NONE, nullptr, s);
            }
        }
        hipLaunchKernelGGL(copy_rows_zero_tail_kernel, dim3(nblk((long long)per)), dim3(256), 0, s, prompt_feat + (size_t)u * mel_len1 * c.mel, m->f_cond.as<float>() + u * per,
                           (long long)mel_len1 * c.mel, (long long)per);
        CV_HIP(hipMemcpyAsync(m->f_x.as<float>() + u * per, noise_cl, per * 4, hipMemcpyDeviceToDevice, s));       // every utterance starts from the same fixed noise
    }
    solve_euler(m, m->f_x.as<float>(), m->f_mu.as<float>(), m->f_spk.as<float>(), m->f_cond.as<float>(), T, n_timesteps, streaming, s, nu);
    hipLaunchKernelGGL(to_channel_first_kernel, dim3(nblk((long long)nu * mel_len2 * c.mel)), dim3(256), 0, s, m->f_x.as<float>(), mel_out, nu, T, c.mel, mel_len1, (const float*)nullptr);
    *mel_len2_out = mel_len2;
}

// Utterances of DIFFERENT lengths in one pass (the reference's padded batch with `mask`, flow/flow.py:236-281, "identical to running each utterance
// alone"): every utterance occupies a block of Tm = max T rows in the stacked [nu][Tm][mel] buffers, zero beyond its own T.  What keeps the padding out
// of the result: the estimator's convolutions are causal (a row only sees rows before it), norms and linears are per row, and attention - the one
// operator that looks to the right - takes a key count per batch row (AttnArgs::klen), ending a row's key loop exactly where it ends for the
// utterance alone.  The valid rows are therefore bit-identical to the single pass; the padded rows compute finite values nobody reads.
static void flow_inference_ragged(cv_flow* m, int nu, const int32_t* token_ids, const int32_t* n_tok, const float* prompt_feat, const int32_t* mel_len1, const float* embedding,
                                  const float* noise_cl, int streaming, int finalize, int n_timesteps, float* mel_out, int32_t* mel_len2_out, void* stream) {
    PrecisionScope prec(m);
    const auto& c = m->cfg; const int d = c.dim;
    hipStream_t s = resolve(m, stream);
    void* stream_r = reinterpret_cast<void*>(s);
    CV_CHECK(nu >= 1 && nu <= 8 && n_timesteps > 0, "cv_flow_inference_ragged: 1..8 utterances");
    std::vector<int> n_enc(nu), T(nu);
    int Tm = 0, tok_max = 0;
    for (int u = 0; u < nu; ++u) {
        n_enc[u] = finalize ? n_tok[u] : n_tok[u] - c.pre_lookahead;
        CV_CHECK(n_enc[u] > 0, "cv_flow_inference_ragged: too few tokens");
        T[u] = 2 * n_enc[u];
        CV_CHECK(mel_len1[u] >= 0 && T[u] - mel_len1[u] > 0, "cv_flow_inference_ragged: prompt longer than the sequence");
        Tm = std::max(Tm, T[u]); tok_max = std::max(tok_max, n_tok[u]);
    }
    if (tok_max > m->inf_cap || (long long)tok_max * nu > m->inf_rows) {
        drop_graphs(m);
        const size_t tk = (size_t)std::max(tok_max, m->inf_cap), rows = std::max((size_t)tok_max * nu, (size_t)m->inf_rows);
        m->f_tok.ensure(rows * d * 4); m->f_h.ensure(2 * rows * d * 4);          // inf_rows promises the stacked size to flow_inference as well
        m->f_mu.ensure(2 * rows * c.mel * 4); m->f_cond.ensure(2 * rows * c.mel * 4); m->f_x.ensure(2 * rows * c.mel * 4);
        m->f_spk.ensure((size_t)8 * c.mel * 4); m->f_spkn.ensure((size_t)c.spk_dim * 4);
        m->inf_cap = (int)tk; m->inf_rows = (long long)rows;
    }
    m->klen.ensure(64 * sizeof(int));
    const size_t per = (size_t)Tm * c.mel;
    CV_HIP(hipMemsetAsync(m->f_mu.p, 0, (size_t)nu * per * 4, s));
    CV_HIP(hipMemsetAsync(m->f_x.p, 0, (size_t)nu * per * 4, s));
    size_t tok_off = 0, feat_off = 0;
    for (int u = 0; u < nu; ++u) {
        hipLaunchKernelGGL(l2_normalize_kernel, dim3(1), dim3(256), 0, s, embedding + (size_t)u * c.spk_dim, m->f_spkn.as<float>(), c.spk_dim);
        lin_cl(m->spk_affine, m->f_spkn.as<float>(), 1, m->f_spk.as<float>() + (size_t)u * c.mel, ACT_NONE, nullptr, s);
        CV_CHECK(cv_gather_rows(m->input_embedding, m->wbf16 ? CV_BF16 : CV_F32, c.vocab, d, token_ids + tok_off, n_tok[u], m->f_tok.as<float>(), 1.f, stream_r) == 0, cv_last_error());
        const float* ctx = finalize ? nullptr : m->f_tok.as<float>() + (size_t)n_enc[u] * d;
        float* mu = m->f_mu.as<float>() + u * per;
        if (c.estimator == 1) dit_front(m, m->f_tok.as<float>(), n_enc[u], ctx, mu, s);
        else {
            flow_encoder(m, m->f_tok.as<float>(), n_enc[u], ctx, streaming, m->f_h.as<float>(), s);
            lin_cl(m->enc_proj, m->f_h.as<float>(), T[u], mu, ACT_NONE, nullptr, s);
        }
        hipLaunchKernelGGL(copy_rows_zero_tail_kernel, dim3(nblk((long long)per)), dim3(256), 0, s, prompt_feat + feat_off, m->f_cond.as<float>() + u * per,
                           (long long)mel_len1[u] * c.mel, (long long)per);
        CV_HIP(hipMemcpyAsync(m->f_x.as<float>() + u * per, noise_cl, (size_t)T[u] * c.mel * 4, hipMemcpyDeviceToDevice, s));
        tok_off += (size_t)n_tok[u]; feat_off += (size_t)mel_len1[u] * c.mel;
    }
    m->host_klen.assign(2 * nu, 0);
    for (int z = 0; z < 2 * nu; ++z) m->host_klen[z] = T[z % nu];                  // estimator batch row z = (cond | uncond) * nu + utterance
    CV_HIP(hipMemcpyAsync(m->klen.p, m->host_klen.data(), (size_t)2 * nu * sizeof(int), hipMemcpyHostToDevice, s));
    m->cur_klen = m->klen.as<int>();
    try {
        solve_euler(m, m->f_x.as<float>(), m->f_mu.as<float>(), m->f_spk.as<float>(), m->f_cond.as<float>(), Tm, n_timesteps, streaming, s, nu);
    } catch (...) { m->cur_klen = nullptr; throw; }
    m->cur_klen = nullptr;
    size_t out_off = 0;
    for (int u = 0; u < nu; ++u) {
        const int len2 = T[u] - mel_len1[u];
        hipLaunchKernelGGL(to_channel_first_kernel, dim3(nblk((long long)len2 * c.mel)), dim3(256), 0, s, m->f_x.as<float>() + u * per, mel_out + out_off, 1, T[u], c.mel,
                           mel_len1[u], (const float*)nullptr);
        mel_len2_out[u] = len2; out_off += (size_t)len2 * c.mel;
    }
}

int cv_flow_solve(cv_flow* m, float* x, const float* mu, const float* spks, const float* cond, int32_t T, int32_t n_timesteps, void* stream) {
    return guarded([&] {
        CV_CHECK(m && m->finalized && x && mu && spks && cond && T > 0 && n_timesteps > 0, "cv_flow_solve: bad arguments");
        PrecisionScope prec(m);
        hipStream_t s = resolve(m, stream);
        const auto& c = m->cfg; const size_t per = (size_t)T * c.mel * 4;
        if (T > m->solve_cap || m->f_x.bytes < per || m->f_mu.bytes < per || m->f_cond.bytes < per || m->f_spk.bytes < (size_t)c.mel * 4) {
            drop_graphs(m);               // captured solves hold the staging addresses
            m->f_x.ensure(per); m->f_mu.ensure(per); m->f_cond.ensure(per); m->f_spk.ensure((size_t)8 * c.mel * 4);
            m->solve_cap = T;
        }
        // the solve runs on the handle's own staging buffers: a captured graph bakes its addresses in, the caller's tensors come and go
        CV_HIP(hipMemcpyAsync(m->f_x.p, x, per, hipMemcpyDeviceToDevice, s)); CV_HIP(hipMemcpyAsync(m->f_mu.p, mu, per, hipMemcpyDeviceToDevice, s));
        CV_HIP(hipMemcpyAsync(m->f_cond.p, cond, per, hipMemcpyDeviceToDevice, s)); CV_HIP(hipMemcpyAsync(m->f_spk.p, spks, (size_t)c.mel * 4, hipMemcpyDeviceToDevice, s));
        solve_euler(m, m->f_x.as<float>(), m->f_mu.as<float>(), m->f_spk.as<float>(), m->f_cond.as<float>(), T, n_timesteps, 0, s, 1);
        CV_HIP(hipMemcpyAsync(x, m->f_x.p, per, hipMemcpyDeviceToDevice, s));
    });
}
int cv_flow_inference_ragged(cv_flow* m, int32_t n_utt, const int32_t* token_ids, const int32_t* n_tok, const float* prompt_feat, const int32_t* mel_len1,
                             const float* embedding, const float* noise_cl, int32_t streaming, int32_t finalize, int32_t n_timesteps, float* mel_out,
                             int32_t* mel_len2_out, void* stream) {
    return guarded([&] {
        CV_CHECK(m && m->finalized && token_ids && n_tok && mel_len1 && embedding && noise_cl && mel_out && mel_len2_out, "cv_flow_inference_ragged: bad arguments");
        flow_inference_ragged(m, n_utt, token_ids, n_tok, prompt_feat, mel_len1, embedding, noise_cl, streaming, finalize, n_timesteps, mel_out, mel_len2_out, stream);
    });
}

int cv_flow_inference(cv_flow* m, const int32_t* token_ids, int32_t n_tok, const float* prompt_feat, int32_t mel_len1, const float* embedding,
                      const float* noise_cl, int32_t streaming, int32_t finalize, int32_t n_timesteps, float* mel_out, int32_t* mel_len2_out, void* stream) {
    return guarded([&] {
        CV_CHECK(m && m->finalized && token_ids && embedding && noise_cl && mel_out && mel_len2_out, "cv_flow_inference: bad arguments");
        flow_inference(m, 1, token_ids, n_tok, prompt_feat, mel_len1, embedding, noise_cl, streaming, finalize, n_timesteps, mel_out, mel_len2_out, stream);
    });
}

int cv_flow_inference_batch(cv_flow* m, int32_t n_utt, const int32_t* token_ids, int32_t n_tok, const float* prompt_feat, int32_t mel_len1, const float* embedding,
                            const float* noise_cl, int32_t streaming, int32_t finalize, int32_t n_timesteps, float* mel_out, int32_t* mel_len2_out, void* stream) {
    return guarded([&] {
        CV_CHECK(m && m->finalized && token_ids && embedding && noise_cl && mel_out && mel_len2_out, "cv_flow_inference_batch: bad arguments");
        flow_inference(m, n_utt, token_ids, n_tok, prompt_feat, mel_len1, embedding, noise_cl, streaming, finalize, n_timesteps, mel_out, mel_len2_out, stream);
    });
}

}  // extern "C"
