// Stage driver: Qwen2LM speech-token language model (replaces Qwen2LM.inference / inference_wrapper and
// Qwen2Encoder.forward_one_step of cosyvoice/llm/llm.py:242-254,458-549 behind boundary B2 of SURVEY.md §8b).
//
// prefill: M = L0 rows through the fp32-accurate GEMM (three-term bf16 split of the activations, gemm_conv.h AX3) + fp32 flash attention, K/V written to a resident fp32 cache.
// decode : one captured hipGraph per handle = {head GEMV + on-device sampling, embed, 24 x (5 kernels), advance};
//          it is replayed once per token; the loop state lives in device memory (DecodeState), so no host sync is
//          needed per token — tokens are read back in chunks.
#include <vector>
#include <memory>
#include <map>
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include "ops.h"
#include "tensor_map.h"
#include "llm_kernels.h"
#include "llm_batch_kernels.h"

using namespace cv;

// Options / environment knobs that select a measured no-go (kept as evidence; their kernels are instantiated and their switches accepted only in a library built with
// -DCV_BUILD_EXPERIMENTS): fused_qkv_attn (qkv_attn_kernel), fused_attn_oproj (attn_oproj_kernel), prefetch (weight-prefetch roles),
// CV_ATTN_BATCH_GQA (attn_decode_batch_gqa_kernel), CV_DOWN_DEEP (skinny_deep_kernel).
#ifdef CV_BUILD_EXPERIMENTS
static constexpr bool kExperiments = true;
#else
static constexpr bool kExperiments = false;
#endif
static void need_experiments(bool wanted, const char* what) {
    if (wanted && !kExperiments) throw cv::Error(std::string(what) + " selects an experiment this library was built without (rebuild with CV_BUILD_EXPERIMENTS=1)");
}

struct cv_llm {
    cv_llm_config cfg{};
    TensorMap tm;
    bool finalized = false;
    struct Layer { const float* ln1; const bf16_t* wqkv; const float* bqkv; const bf16_t* wo; const float* ln2; const bf16_t* wgu; const bf16_t* wdown; };
    std::vector<Layer> layers;
    // optional fp8 (e4m3 + one fp32 scale per row) copies of the matrices for the batched decode ("<name>.f8" / "<name>.f8s", option "batch_fp8")
    struct F8 { const unsigned char* w = nullptr; const float* s = nullptr; };
    struct LayerF8 { F8 qkv, o, gu, down; };
    std::vector<LayerF8> layers_f8; F8 head_f8; bool have_fp8 = false; int batch_fp8 = 0;
    const float* norm = nullptr; const bf16_t* head_w = nullptr; const float* head_b = nullptr; const bf16_t* speech_emb = nullptr;
    int V = 0, qkv_dim = 0;
    // device state
    DevBuf kcache, vcache, rope_cos, rope_sin, state, tokens, uniforms, sparams;
    SampleParams* host_sp = nullptr;
    DevBuf h, qkv, act, logits, attn_part;            // decode activations (+ split-attention partials)
    DevBuf newtok;                                    // fused qkv + attention: roped q [heads][64], roped k / v [kv_heads][64] of the new token
    // option "fused_qkv_attn": 1 = qkv_attn_kernel (4 launches per layer), 0 = separate qkv / attention kernels.  Measured on MI355X
    // (profiles/r2_decode_ab.txt): fused 9.6 us vs 3.4 + 3.9 us - the q rows are recomputed by the 8 key-slice workgroups of a head, which run on
    // 8 different XCDs, so the PMC shows 13.7 MB of HBM reads per launch instead of 2.1 MB.  Kept, tested, off by default.
    int fused_qkv_attn = 0;
    int head_rows = 1;                                // rows per 16-lane group of the head GEMV (1: 411 workgroups, 2: 206)
    int head_waves = 4;                               // head_rows == 1: waves per workgroup of the head GEMV (4: 411 workgroups = 1.6 per CU; 7: 235 workgroups, one per CU) - option "head_waves" / CV_HEAD_WAVES
    int attn_splits = 8;
    // option "prefetch" (env CV_DECODE_PREFETCH): extra workgroups of the short decode kernels read the weights the bandwidth-bound kernels behind them
    // will stream (llm_kernels.h PrefetchArgs).  0 = off; 1 = attention fetches gate / up, o_proj fetches down; 2 = qkv fetches gate / up, attention
    // fetches down.  "prefetch_shift": dev knob (fetch for another consumer workgroup: breaks the XCD match).
    int prefetch = 0, prefetch_shift = 0; DevBuf pf_sink;
    // option "fused_attn_oproj" (env CV_DECODE_FUSED_O): attention + o_proj in one launch (attn_oproj_kernel: the o_proj GEMV split by head, the per-head
    // contributions summed into the residual by gate / up's prologue) - 4 launches per layer.  "oproj_rblocks": row blocks per head (4 | 8).
    int fused_attn_oproj = 0, oproj_rblocks = 4, oproj_waves = 8; DevBuf opart, h2;
    int only_cat = -1;                  // cv_llm_profile_chain: enqueue only the launches of this category (-1 = all)
    DevBuf pf_x, pf_xn, pf_qkv, pf_attn, pf_gu, pf_act, pf_part; // prefill activations (grown on demand; pf_part: split-K partials of down on the weight-stationary path)
    int prefill_rows = [] { const char* e = getenv("CV_PREFILL_ROWS"); return (e && e[0] == '1') ? 1 : 0; }();   // option "prefill_rows" = 1: prompts of <= 160 rows on skinny_rows_kernel (measured neutral: off)
    int pf_rows = 0;
    // decode graph
    hipGraphExec_t graph = nullptr; hipStream_t graph_stream = nullptr; hipStream_t own_stream = nullptr;
    cv_sampling sp{}; bool sp_valid = false; bool use_graph = true;
    int* host_tokens = nullptr; DecodeState* host_state = nullptr;
    // lock-step batched decode (llm_batch_kernels.h): nb slots, each with its own KV cache / state / token history
    struct Batch {
        int nb = 0;
        DevBuf kcache, vcache, state, tokens, sparams, uniforms, h, qkv, act, logits, attn, dpart, apart;   // attn: merged attention [nb][heads*64]; dpart: split-K partials of down
        std::vector<DecodeState> host_state; std::vector<int> host_tokens; std::vector<SampleParams> host_sp;
        hipGraphExec_t graphs[3] = {nullptr, nullptr, nullptr}; hipStream_t graph_stream = nullptr; int graph_nb = 0;      // one captured step per attention form (0 VALU, 1 MFMA, 2 MFMA
        int attn_mode = 1;                                       // with the fixed key partition): all kept - a batch crosses the rule's threshold as its contexts grow, a handle serves pinned and unpinned paths
        void drop_graphs() { for (auto& g : graphs) if (g) { (void)hipGraphExecDestroy(g); g = nullptr; } }
    } bt;
    // Round 3: fragment-ordered copies of the matrices for the batched decode (skinny_pk_kernel, llm_batch_kernels.h), made on the device the first time
    // the batch path is used: [row tile][k tile][lane][8 bf16], so that one wave load is 1 KB contiguous.  Keyed by the row-major tensor's address.
    // +0.73 GB of HBM for Qwen2-0.5B (the 288 GB of an MI355X are what the batched path is sized for).  Option "batch_packed" (default 1).
    // Shared between the handles that register the SAME weight tensors (a decode group's sibling handles, llm.py Qwen2LM.sibling): one copy per tensor address in a
    // process-wide table of weak references - the last handle that lets go frees it (every sibling used to pack its own 0.73 GB).
    std::map<const void*, std::shared_ptr<DevBuf>> packed;
    int batch_packed = [] { const char* e = getenv("CV_BATCH_PACKED"); return (e && e[0] == '0') ? 0 : 1; }();        // env: A/B knob for bench runs
    // Batched decode attention form: -1 = chosen per decode call from the slot count and the longest context (the default, the measured rule in batch_decode), 0 = the
    // per-head VALU kernel, 1 = the fp32-MFMA kernel + merge (slice count by the slot count), 2 = the fp32-MFMA kernel with ONE key partition for every slot count.  The two forms sum the same products in a different order: with the automatic rule a request's logits
    // depend - in the last fp32 bits - on how many slots shared its decode call and on its neighbours' context lengths.  A caller that needs a request's tokens to be
    // independent of the batch composition down to near-tie argmax decisions pins one form (option "batch_attn", read once here from CV_ATTN_BATCH).
    int batch_attn = [] { const char* e = getenv("CV_ATTN_BATCH"); return !e ? -1 : (e[0] == '0' ? 0 : e[0] == '2' ? 2 : 1); }();
    const bf16_t* pk(const bf16_t* w) const { if (!batch_packed) return nullptr; auto it = packed.find(w); return it == packed.end() ? nullptr : it->second->as<bf16_t>(); }
    size_t slot_cache() const { return layer_cache() * cfg.layers; }
    // optional per-kernel HIP-event timing of one eager decode step (bench.py roofline)
    bool profiling = false; std::vector<std::pair<int, std::pair<hipEvent_t, hipEvent_t>>> prof_events;
    size_t layer_cache() const { return (size_t)cfg.kv_heads * cfg.max_len * 64; }
    ~cv_llm() {
        if (graph) (void)hipGraphExecDestroy(graph);
        bt.drop_graphs();
        if (own_stream) (void)hipStreamDestroy(own_stream);
        if (host_tokens) (void)hipHostFree(host_tokens);
        if (host_state) (void)hipHostFree(host_state);
        if (host_sp) (void)hipHostFree(host_sp);
    }
};

static void llm_finalize(cv_llm* m) {
    const auto& c = m->cfg;
    CV_CHECK(c.hidden % 128 == 0 && c.inter % 128 == 0, "llm: hidden and inter must be multiples of 128");
    CV_CHECK(c.heads % c.kv_heads == 0 && c.heads * 64 % 128 == 0, "llm: bad head configuration (head_dim is fixed at 64)");
    CV_CHECK(c.max_len > 0 && c.max_len <= 32768, "llm: max_len (KV capacity) must be in (0, 32768] (Qwen2.5 max_position_embeddings)");
    CV_CHECK(c.hidden <= 896 && c.heads * 64 <= 5120 && c.inter <= 5120, "llm: hidden must fit one wave-row (<= 896, fused RMSNorm) and inter <= 5120");
    m->V = c.speech_vocab;
    CV_CHECK(m->V > 0 && m->V <= 8192, "llm: speech vocab above the sampler limit");
    m->qkv_dim = (c.heads + 2 * c.kv_heads) * 64;
    const long long H = c.hidden;
    m->layers.resize(c.layers);
    for (int i = 0; i < c.layers; ++i) {
        const std::string p = "layers." + std::to_string(i) + ".";
        auto& L = m->layers[i];
        L.ln1 = m->tm.f32(p + "ln1", H);
        L.wqkv = m->tm.bf16(p + "wqkv", (long long)m->qkv_dim * H);
        L.bqkv = m->tm.f32(p + "bqkv", m->qkv_dim);
        L.wo = m->tm.bf16(p + "wo", H * c.heads * 64);
        L.ln2 = m->tm.f32(p + "ln2", H);
        L.wgu = m->tm.bf16(p + "wgu", 2LL * c.inter * H);
        L.wdown = m->tm.bf16(p + "wdown", H * c.inter);
    }
    m->have_fp8 = m->tm.has("head.w.f8");
    if (m->have_fp8) {
        auto f8 = [&](const std::string& name, long long rows, long long cols) {
            cv_llm::F8 f; f.w = reinterpret_cast<const unsigned char*>(m->tm.get(name + ".f8", CV_U8, rows * cols).p); f.s = m->tm.f32(name + ".f8s", rows); return f;
        };
        CV_CHECK(H % 64 == 0 && c.inter % 64 == 0 && c.heads * 64 % 64 == 0, "llm(fp8): K must be a multiple of 64");
        m->layers_f8.resize(c.layers);
        for (int i = 0; i < c.layers; ++i) {
            const std::string p = "layers." + std::to_string(i) + ".";
            m->layers_f8[i].qkv = f8(p + "wqkv", m->qkv_dim, H); m->layers_f8[i].o = f8(p + "wo", H, c.heads * 64);
            m->layers_f8[i].gu = f8(p + "wgu", 2LL * c.inter, H); m->layers_f8[i].down = f8(p + "wdown", H, c.inter);
        }
        m->head_f8 = f8("head.w", m->V, H);
    }
    m->norm = m->tm.f32("norm", H);
    m->head_w = m->tm.bf16("head.w", (long long)m->V * H);
    m->head_b = m->tm.f32("head.b", m->V);
    m->speech_emb = m->tm.bf16("embed.speech", (long long)m->V * H);

    m->kcache.ensure(m->layer_cache() * c.layers * sizeof(float));
    m->vcache.ensure(m->layer_cache() * c.layers * sizeof(float));
    // rotate-half RoPE tables (transformers Qwen2RotaryEmbedding: inv_freq = theta^(-2f/64), fp32)
    std::vector<float> cs((size_t)c.max_len * 32), sn((size_t)c.max_len * 32);
    for (int f = 0; f < 32; ++f) {
        const float inv = 1.0f / powf(c.rope_theta, (float)(2 * f) / 64.0f);
        for (int p = 0; p < c.max_len; ++p) { const float a = (float)p * inv; cs[(size_t)p * 32 + f] = cosf(a); sn[(size_t)p * 32 + f] = sinf(a); }
    }
    m->rope_cos.ensure(cs.size() * 4); m->rope_sin.ensure(sn.size() * 4);
    CV_HIP(hipMemcpy(m->rope_cos.p, cs.data(), cs.size() * 4, hipMemcpyHostToDevice));
    CV_HIP(hipMemcpy(m->rope_sin.p, sn.data(), sn.size() * 4, hipMemcpyHostToDevice));
    m->state.ensure(sizeof(DecodeState));
    m->tokens.ensure((size_t)c.max_len * sizeof(int));
    m->uniforms.ensure((size_t)c.max_len * 2 * sizeof(float));
    m->h.ensure(H * 4); m->qkv.ensure((size_t)m->qkv_dim * 4);
    m->attn_part.ensure((size_t)c.heads * 16 * ATTN_PART * 4);
    m->newtok.ensure((size_t)(c.heads + 2 * c.kv_heads) * 64 * 4);
    if (kExperiments) if (const char* e = getenv("CV_DECODE_FUSED_QKV")) m->fused_qkv_attn = e[0] != '0';     // dev knob for A/B runs (also: option "fused_qkv_attn")
    if (const char* e = getenv("CV_HEAD_WAVES")) m->head_waves = atoi(e) == 7 ? 7 : 4;
    if (kExperiments) if (const char* e = getenv("CV_DECODE_PREFETCH")) m->prefetch = atoi(e);                 // dev knobs for A/B runs (also: options "prefetch", "prefetch_shift")
    if (const char* e = getenv("CV_DECODE_PREFETCH_SHIFT")) m->prefetch_shift = atoi(e);
    m->pf_sink.ensure(64);
    if (kExperiments) if (const char* e = getenv("CV_DECODE_FUSED_O")) m->fused_attn_oproj = e[0] != '0';      // dev knobs for A/B runs (also: options "fused_attn_oproj", "oproj_rblocks")
    if (const char* e = getenv("CV_OPROJ_RBLOCKS")) m->oproj_rblocks = atoi(e) == 8 ? 8 : 4;
    if (const char* e = getenv("CV_OPROJ_WAVES")) m->oproj_waves = atoi(e) == 16 ? 16 : 8;
    m->opart.ensure((size_t)16 * H * 4); m->h2.ensure(H * 4);
    m->act.ensure((size_t)c.inter * 4); m->logits.ensure((size_t)m->V * 4);
    CV_HIP(hipHostMalloc((void**)&m->host_tokens, (size_t)c.max_len * sizeof(int)));
    CV_HIP(hipHostMalloc((void**)&m->host_state, sizeof(DecodeState)));
    CV_HIP(hipHostMalloc((void**)&m->host_sp, sizeof(SampleParams)));
    m->sparams.ensure(sizeof(SampleParams));
    CV_HIP(hipMemset(m->state.p, 0, sizeof(DecodeState)));
    m->finalized = true;
}

// Graph capture is illegal on the legacy default stream: a NULL stream is mapped to a handle-owned *blocking* stream,
// which HIP orders against the default stream implicitly (so torch work on stream 0 stays ordered with ours).
static hipStream_t resolve(cv_llm* m, void* s) {
    if (s) return as_stream(s);
    if (!m->own_stream) CV_HIP(hipStreamCreate(&m->own_stream));
    return m->own_stream;
}

static LinearW lw(const bf16_t* w, const float* b, int N, int K) { LinearW l; l.w = w; l.b = b; l.N = N; l.K = K; l.Kp = round_up32(K); l.bf16 = true; return l; }

// Fragment-ordered copies of the matrices ([row tile][k tile][lane][8 bf16]: one wave load = 1 KB contiguous) for skinny_pk_kernel (batched decode) and
// skinny_rows_kernel (prefill), made on the device once per handle, the first time either path runs.
static void ensure_packed(cv_llm* m, hipStream_t s) {
    if (!m->batch_packed || !m->packed.empty()) return;
    const auto& c = m->cfg;
    static std::map<std::pair<const void*, long long>, std::weak_ptr<DevBuf>> shared;       // (tensor address, element count) -> the live copy, under runtime_lock()
    std::lock_guard<std::recursive_mutex> lk(runtime_lock());
    auto pack = [&](const bf16_t* w, long long N, long long K) {
        if (!w || K % 32 != 0 || m->packed.count(w)) return;
        auto& slot = shared[{w, N * K}];
        std::shared_ptr<DevBuf> buf = slot.lock();
        if (!buf) {
            const long long pieces = ((N + 15) / 16) * (K / 32) * 64;
            buf = std::make_shared<DevBuf>(); buf->ensure((size_t)pieces * 16);
            hipLaunchKernelGGL(pack_frag_kernel, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, s, w, buf->as<bf16_t>(), (int)N, (int)K);
            slot = buf;
        }
        m->packed[w] = std::move(buf);
    };
    const long long H = c.hidden, A = c.heads * 64;
    pack(m->head_w, m->V, H);
    for (const auto& L : m->layers) { pack(L.wqkv, m->qkv_dim, H); pack(L.wo, H, A); pack(L.wgu, 2LL * c.inter, H); pack(L.wdown, H, c.inter); }
    CV_HIP(hipStreamSynchronize(s));
    CV_HIP(hipGetLastError());
}

// the weight-stationary GEMM over the rows of a prefill (skinny_rows_kernel): geometry rules of skinny() for the fragment-ordered path
static bool skinny_rows_fits(int N, int K, int ksplit) {
    const int tiles = K / 32 / ksplit;
    return K % (32 * ksplit) == 0 && tiles >= 4 && (tiles + 3) / 4 <= 7 && N % 4 == 0;
}
static void skinny_rows(const SkinnyArgs& a, int rt, hipStream_t s, const bf16_t* wp) {
    const int tiles = a.K / 32 / a.ksplit, row_tiles = (a.N + 15) / 16;
    CV_CHECK(wp && skinny_rows_fits(a.N, a.K, a.ksplit), "skinny_rows: needs the fragment-ordered weights and a K range of 4 .. 28 tiles per workgroup");
    CV_CHECK(!(a.gamma && a.ksplit != 1) && (a.mode == 2) == (a.ksplit > 1), "skinny_rows: split-K workgroups leave raw partials (mode 2), the fused norm needs the whole row");
    SkinnyArgs b = a; b.W = wp;
    const dim3 grid(((row_tiles + rt - 1) / rt) * a.ksplit);
    const bool deep = (tiles + 3) / 4 > 5;
    if (rt == 1) hipLaunchKernelGGL((skinny_rows_kernel<1, 7>), grid, dim3(256), 0, s, b);
    else if (rt == 4) hipLaunchKernelGGL((skinny_rows_kernel<4, 7>), grid, dim3(256), 0, s, b);
    else if (deep) hipLaunchKernelGGL((skinny_rows_kernel<2, 7>), grid, dim3(256), 0, s, b);
    else hipLaunchKernelGGL((skinny_rows_kernel<2, 5>), grid, dim3(256), 0, s, b);
}
static int down_ksplit(int inter);

// One prompt segment of a prefill: rows [row0, row0 + L) of the stacked input are the positions pos0 .. pos0 + L - 1 of a sequence whose KV cache
// (layout [layers][kv_heads][max_len][64]) starts at kc / vc.  The GEMMs / norms of a prefill run over ALL rows of all segments at once
// (M = sum of the prompt lengths: the weights are read once and the tiles are fuller), RoPE + cache write and the causal attention per segment.
struct PrefillSeg { int row0, L, pos0; float* kc; float* vc; };

static void llm_prefill_rows(cv_llm* m, const float* x_in, int R, const std::vector<PrefillSeg>& segs, hipStream_t s) {
    const auto& c = m->cfg;
    const int H = c.hidden, Q = m->qkv_dim, A = c.heads * 64;
    if (R > m->pf_rows) {
        m->pf_x.ensure((size_t)R * H * 4); m->pf_xn.ensure((size_t)R * H * 4); m->pf_qkv.ensure((size_t)R * Q * 4);
        m->pf_attn.ensure((size_t)R * A * 4); m->pf_gu.ensure((size_t)R * 2 * c.inter * 4); m->pf_act.ensure((size_t)R * c.inter * 4);
        m->pf_rows = R;
    }
    float* x = m->pf_x.as<float>(); float* xn = m->pf_xn.as<float>(); float* qkv = m->pf_qkv.as<float>();
    float* at = m->pf_attn.as<float>(); float* gu = m->pf_gu.as<float>(); float* act = m->pf_act.as<float>();
    CV_HIP(hipMemcpyAsync(x, x_in, (size_t)R * H * 4, hipMemcpyDeviceToDevice, s));
    // Round 3, opt-in (prefill_rows): up to 160 rows (one prompt) the GEMMs run weight-stationary on the fragment-ordered copies (skinny_rows_kernel: RMSNorm
    // in the prologue, SiLU * up in the epilogue, split-K down + sum_partials) - 7 launches per layer instead of 9.  Measured 2.95 vs 2.99 ms: off by default.
    int dks = down_ksplit(c.inter);
    if (!skinny_rows_fits(H, c.inter, dks) && skinny_rows_fits(H, c.inter, 1)) dks = 1;       // small models: the whole K in one workgroup
    const bool rows_path = m->prefill_rows && m->batch_packed && R <= 160 && skinny_rows_fits(Q, H, 1) && skinny_rows_fits(H, A, 1) &&
                           skinny_rows_fits(2 * c.inter, H, 1) && skinny_rows_fits(H, c.inter, dks);
    if (rows_path) {
        ensure_packed(m, s);
        if (dks > 1) m->pf_part.ensure((size_t)dks * R * H * 4);
    }
    for (int i = 0; i < c.layers; ++i) {
        const auto& L = m->layers[i];
        if (rows_path) {
            skinny_rows(SkinnyArgs{L.wqkv, L.bqkv, x, H, qkv, Q, Q, H, L.ln1, c.rms_eps, nullptr, 0, 0, R, 1}, 1, s, m->pk(L.wqkv));
        } else {
            norm_rows(NormArgs{x, xn, R, H, L.ln1, nullptr, c.rms_eps, 1, ACT_NONE, 1.f, nullptr, nullptr, R}, s);
            linear(xn, R, lw(L.wqkv, L.bqkv, Q, H), qkv, ACT_NONE, nullptr, s);
        }
        for (const auto& g : segs) {
            float* kc = g.kc + m->layer_cache() * i; float* vc = g.vc + m->layer_cache() * i;
            float* q = qkv + (size_t)g.row0 * Q;
            hipLaunchKernelGGL(rope_store_kernel, dim3(g.L), dim3(256), 0, s, q, g.L, c.heads, c.kv_heads, g.pos0,
                               m->rope_cos.as<float>(), m->rope_sin.as<float>(), kc, vc, c.max_len);
            AttnArgs a{};
            a.q = q; a.q_batch = 0; a.q_row = Q; a.q_head = 64;
            a.k = kc; a.k_batch = 0; a.k_row = 64; a.k_head = c.max_len * 64;
            a.v = vc; a.v_batch = 0; a.v_row = 64; a.v_head = c.max_len * 64;
            a.o = at + (size_t)g.row0 * A; a.o_batch = 0; a.o_row = A; a.o_head = 64;
            a.B = 1; a.H = c.heads; a.kv_group = c.heads / c.kv_heads; a.Tq = g.L; a.Tk = g.pos0 + g.L;      // causal with offset Tk - Tq
            a.scale = 0.125f; a.mask_mode = MASK_CAUSAL; a.chunk = 0; a.rel_bd = nullptr;
            attention(a, s);
        }
        if (rows_path) {
            skinny_rows(SkinnyArgs{L.wo, nullptr, at, A, x, H, H, A, nullptr, 0.f, x, H, 0, R, 1}, 1, s, m->pk(L.wo));
            skinny_rows(SkinnyArgs{L.wgu, nullptr, x, H, act, c.inter, 2 * c.inter, H, L.ln2, c.rms_eps, nullptr, 0, 1, R, 1}, 4, s, m->pk(L.wgu));
            if (dks > 1) {
                float* part = m->pf_part.as<float>();
                skinny_rows(SkinnyArgs{L.wdown, nullptr, act, c.inter, part, H, H, c.inter, nullptr, 0.f, nullptr, 0, 2, R, dks}, 2, s, m->pk(L.wdown));
                hipLaunchKernelGGL(sum_partials_kernel, dim3((unsigned)(((long long)R * H / 4 + 255) / 256)), dim3(256), 0, s, part, dks, R, H, x, (long long)H, x, (long long)H);
            } else {
                skinny_rows(SkinnyArgs{L.wdown, nullptr, act, c.inter, x, H, H, c.inter, nullptr, 0.f, x, H, 0, R, 1}, 2, s, m->pk(L.wdown));
            }
            continue;
        }
        linear(at, R, lw(L.wo, nullptr, H, A), x, ACT_NONE, x, s);
        norm_rows(NormArgs{x, xn, R, H, L.ln2, nullptr, c.rms_eps, 1, ACT_NONE, 1.f, nullptr, nullptr, R}, s);
        linear(xn, R, lw(L.wgu, nullptr, 2 * c.inter, H), gu, ACT_NONE, nullptr, s);
        const long long n_out = (long long)R * c.inter;
        hipLaunchKernelGGL(silu_mul_kernel, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, s, gu, act, n_out);
        linear(act, R, lw(L.wdown, nullptr, H, c.inter), x, ACT_NONE, x, s);
    }
}

// append = false: a new request, the rows become positions 0 .. L0-1.  append = true (inference_bistream, llm/llm.py:551-661): the rows
// are forwarded on top of the positions already cached, exactly like `forward_one_step(lm_input, cache=cache)` with a multi-row lm_input;
// the running request (tokens emitted so far, step counter) continues and a `done` left by a fill token is cleared.
static void llm_prefill(cv_llm* m, const float* x_in, int L0, hipStream_t s, bool append = false) {
    const auto& c = m->cfg;
    CV_CHECK(m->finalized, "llm: call cv_llm_finalize first");
    const int pos0 = append ? m->host_state->pos : 0;
    CV_CHECK(L0 > 0 && pos0 + L0 < c.max_len, "llm_prefill: prompt length out of range (KV capacity max_len)");
    llm_prefill_rows(m, x_in, L0, {PrefillSeg{0, L0, pos0, m->kcache.as<float>(), m->vcache.as<float>()}}, s);
    CV_HIP(hipMemcpyAsync(m->h.p, m->pf_x.as<float>() + (size_t)(L0 - 1) * c.hidden, (size_t)c.hidden * 4, hipMemcpyDeviceToDevice, s));
    DecodeState st{}; st.pos = pos0 + L0; st.step = 0; st.done = 0; st.n_tokens = 0; st.last_token = 0; st.stop_token = -1;
    if (append) { st.step = m->host_state->step; st.n_tokens = m->host_state->n_tokens; st.last_token = m->host_state->last_token; }
    *m->host_state = st;
    CV_HIP(hipMemcpyAsync(m->state.p, m->host_state, sizeof(DecodeState), hipMemcpyHostToDevice, s));
    CV_HIP(hipStreamSynchronize(s));     // host_state is reused by the next call
}

// kernel categories for cv_llm_profile_step: 0 qkv, 1 attention, 2 o_proj, 3 gate_up, 4 down, 5 head, 6 sample, 7 other
struct ProfScope {
    cv_llm* m; hipStream_t s; int cat; hipEvent_t e0 = nullptr, e1 = nullptr;
    ProfScope(cv_llm* m_, hipStream_t s_, int c) : m(m_), s(s_), cat(c) {
        if (!m->profiling) return;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventRecord(e0, s);
    }
    ~ProfScope() { if (!m->profiling) return; (void)hipEventRecord(e1, s); m->prof_events.push_back({cat, {e0, e1}}); }
};

// dev knob (CV_GEMV_SHARED_NORM=0 restores the per-wave RMSNorm prologue for A/B runs)
static const bool g_gemv_shared_norm = [] { const char* e = getenv("CV_GEMV_SHARED_NORM"); return !(e && e[0] == '0'); }();

// picks the instantiation from K (= 128 * steps): <=7 steps -> one wave per 4*ROWS rows; <=40 steps -> 4-way split-K
static void gemv(const GemvArgs& a, int rows, hipStream_t s, int nsp = 0, int waves = 4) {
    const int steps = a.K / 128;
    if (nsp > 0) {                                   // o_proj over split-attention partials (rows == 1, K = heads * 64 <= 1024)
        CV_CHECK(steps <= 8 && rows == 1 && a.mode == 0 && !a.gamma && a.part, "gemv: partial-combine prologue is for the o_proj shape only");
        const dim3 g4(a.pf.p ? a.pf.first + prefetch_groups(a.pf.cons_bytes, 4, 10) * a.pf.stride : (a.N + 3) / 4);
        CV_CHECK(!a.pf.p || a.pf.first >= (a.N + 3) / 4, "gemv: prefetch workgroups must follow the GEMV's own");
        if (nsp == 4) hipLaunchKernelGGL((gemv_kernel<2, 1, 4, 4>), g4, dim3(256), 0, s, a);
        else if (nsp == 8) hipLaunchKernelGGL((gemv_kernel<2, 1, 4, 8>), g4, dim3(256), 0, s, a);
        else if (nsp == 16) hipLaunchKernelGGL((gemv_kernel<2, 1, 4, 16>), g4, dim3(256), 0, s, a);
        else throw Error("gemv: attn_splits must be 4, 8 or 16");
        return;
    }
    CV_CHECK(a.K % 128 == 0 && steps >= 1 && steps <= 40, "gemv: K must be a multiple of 128 and at most 5120");
    const int units = a.mode == 1 ? a.N / 2 : (a.N + rows - 1) / rows;       // 16-lane groups needed
    const dim3 grid((units + 3) / 4);
    if (a.mode == 1) rows = 2;
    if (steps <= 7) {
        if (a.gamma && g_gemv_shared_norm) {                 // 4 waves share the normalised input through LDS (gemv_norm_kernel)
            const dim3 g16((units + 15) / 16);
            static const bool five = [] { const char* e = getenv("CV_GEMV_GATEUP_WAVES"); return !(e && e[0] == '4'); }();   // dev knob for A/B runs
            if (a.opart) {                                   // gate / up behind attn_oproj_kernel: the per-head o_proj contributions join the residual in the prologue
                CV_CHECK(rows == 2 && a.n_opart >= 1 && a.n_opart <= 16, "gemv: the o_proj-contribution prologue is for the gate / up shape, <= 16 heads");
                hipLaunchKernelGGL((gemv_norm_kernel<7, 2, 5, true, true>), dim3((units + 19) / 20), dim3(320), 0, s, a);
            }
            else if (rows == 2 && five) hipLaunchKernelGGL((gemv_norm_kernel<7, 2, 5>), dim3((units + 19) / 20), dim3(320), 0, s, a);
            else if (rows == 2) hipLaunchKernelGGL((gemv_norm_kernel<7, 2>), g16, dim3(256), 0, s, a);
            else if (waves == 7) hipLaunchKernelGGL((gemv_norm_kernel<7, 1, 7>), dim3((units + 27) / 28), dim3(448), 0, s, a);   // the head: 6564 rows as 235 seven-wave workgroups, one balanced round
            else if (a.pf.p) {                               // rows == 1 host (qkv): own workgroups, padding, then the prefetch workgroups
                CV_CHECK(a.pf.first >= (int)g16.x, "gemv: prefetch workgroups must follow the GEMV's own");
                hipLaunchKernelGGL((gemv_norm_kernel<7, 1>), dim3(a.pf.first + prefetch_groups(a.pf.cons_bytes, 4, 9) * a.pf.stride), dim3(256), 0, s, a);
            }
            else           hipLaunchKernelGGL((gemv_norm_kernel<7, 1>), g16, dim3(256), 0, s, a);
            return;
        }
        if (rows == 2) hipLaunchKernelGGL((gemv_kernel<7, 2, 1>), grid, dim3(64), 0, s, a);
        else           hipLaunchKernelGGL((gemv_kernel<7, 1, 1>), grid, dim3(64), 0, s, a);
    } else {
        CV_CHECK(!a.gamma, "gemv: fused RMSNorm needs the whole row in one wave (K <= 896)");
        if (rows == 2) hipLaunchKernelGGL((gemv_kernel<10, 2, 4>), grid, dim3(256), 0, s, a);
        else           hipLaunchKernelGGL((gemv_kernel<10, 1, 4>), grid, dim3(256), 0, s, a);
    }
}

// one token: head + sample, then (unless done) embed + backbone for the sampled token.  The graph is ONE chain on purpose: a forked branch
// (an L2 warm-up kernel per layer next to qkv / attention / o_proj) was measured at +21 us per layer for the fork / join edges alone on this
// runtime (profiles/r2_batch_decode_ab.txt), and early-launched dependents on a second stream deadlock under hipGraph scheduling.
static void llm_enqueue_step(cv_llm* m, hipStream_t s) {
    const auto& c = m->cfg;
    const DecodeState* st = m->state.as<DecodeState>();
    float* h = m->h.as<float>(); float* qkv = m->qkv.as<float>(); float* act = m->act.as<float>();
    auto want = [&](int cat) { return m->only_cat < 0 || m->only_cat == cat; };
    if (want(5)) { ProfScope ps(m, s, 5); gemv(GemvArgs{m->head_w, m->head_b, h, m->logits.as<float>(), m->V, c.hidden, m->norm, c.rms_eps, nullptr, 0, st}, m->head_rows, s, 0, m->head_rows == 1 ? m->head_waves : 4); }
    SampleArgs sa{};
    sa.logits = m->logits.as<float>(); sa.V = m->V; sa.sp = m->sparams.as<SampleParams>(); sa.uniforms = m->uniforms.as<float>();
    sa.st = m->state.as<DecodeState>(); sa.tokens = m->tokens.as<int>(); sa.max_tokens = c.max_len;
    sa.emb_table = m->speech_emb; sa.emb_dim = c.hidden; sa.h_out = h;             // sampling + embedding of the sampled token: one launch
    if (want(6)) { ProfScope ps(m, s, 6); hipLaunchKernelGGL(sample_kernel, dim3(1), dim3(1024), 0, s, sa); }
    for (int i = 0; i < c.layers; ++i) {
        const auto& L = m->layers[i];
        const int nsp = m->attn_splits;
        GemvArgs go{L.wo, nullptr, nullptr, h, c.hidden, c.heads * 64, nullptr, 0.f, h, 0, st};
        go.part = m->attn_part.as<float>();
        // weight prefetch by the short kernels (PrefetchArgs): consumer geometry of gate / up (gemv_norm_kernel<7,2,5>: 20 row pairs per workgroup) and
        // down (gemv_kernel<10,1,4>: 4 rows per workgroup)
        auto round8 = [](int v) { return (v + 7) / 8 * 8; };
        PrefetchArgs pgu{}, pdn{};
        if (m->prefetch && !m->fused_qkv_attn) {
            const bool five = !(getenv("CV_GEMV_GATEUP_WAVES") && getenv("CV_GEMV_GATEUP_WAVES")[0] == '4');
            const int gu_rows = !g_gemv_shared_norm ? 8 : five ? 40 : 32;
            pgu.p = reinterpret_cast<const char*>(L.wgu); pgu.bytes = 2LL * c.inter * c.hidden * 2; pgu.cons_bytes = gu_rows * c.hidden * 2;
            pgu.n_cons = (2 * c.inter + gu_rows - 1) / gu_rows; pgu.stride = round8(pgu.n_cons); pgu.shift = m->prefetch_shift; pgu.sink = m->pf_sink.as<unsigned>();
            pdn.p = reinterpret_cast<const char*>(L.wdown); pdn.bytes = 2LL * c.hidden * c.inter; pdn.cons_bytes = 4 * c.inter * 2;
            pdn.n_cons = (c.hidden + 3) / 4; pdn.stride = round8(pdn.n_cons); pdn.shift = m->prefetch_shift; pdn.sink = pgu.sink;
        }
#ifdef CV_BUILD_EXPERIMENTS
        if (m->fused_attn_oproj && !m->fused_qkv_attn) {
            // qkv -> attention + per-head o_proj contributions -> gate / up (residual + contributions in its prologue, h2 = the new residual) -> down (+ h2)
            if (want(0)) { ProfScope ps(m, s, 0); gemv(GemvArgs{L.wqkv, L.bqkv, h, qkv, m->qkv_dim, c.hidden, L.ln1, c.rms_eps, nullptr, 0, st}, 1, s); }
            const int R = m->oproj_rblocks, rpb = ((c.hidden + R - 1) / R + 7) / 8 * 8;
            CV_CHECK(rpb <= 256 && c.heads <= 16, "fused_attn_oproj: hidden / oproj_rblocks must be <= 256 rows, heads <= 16");
            AttnOprojArgs ao{qkv, m->kcache.as<float>() + m->layer_cache() * i, m->vcache.as<float>() + m->layer_cache() * i, m->rope_cos.as<float>(), m->rope_sin.as<float>(),
                             c.heads, c.kv_heads, c.max_len, st, L.wo, c.hidden, m->opart.as<float>(), (c.hidden + rpb - 1) / rpb, rpb};
            if (want(1)) {
                ProfScope ps(m, s, 1);
                if (m->oproj_waves == 16) hipLaunchKernelGGL((attn_oproj_kernel<16, 2, 6>), dim3(c.heads * ao.rblocks), dim3(1024), 0, s, ao);   // 384 keys per pass
                else hipLaunchKernelGGL((attn_oproj_kernel<8, 4, 10>), dim3(c.heads * ao.rblocks), dim3(512), 0, s, ao);                         // 320 keys per pass
            }
            GemvArgs gg{L.wgu, nullptr, h, act, 2 * c.inter, c.hidden, L.ln2, c.rms_eps, nullptr, 1, st};
            gg.opart = m->opart.as<float>(); gg.n_opart = c.heads; gg.x_out = m->h2.as<float>();
            if (want(3)) { ProfScope ps(m, s, 3); gemv(gg, 2, s); }
            GemvArgs gd2{L.wdown, nullptr, act, h, c.hidden, c.inter, nullptr, 0.f, m->h2.as<float>(), 0, st};
            if (i == c.layers - 1 && m->only_cat < 0) gd2.advance = m->state.as<DecodeState>();
            if (want(4)) { ProfScope ps(m, s, 4); gemv(gd2, 1, s); }
            continue;
        }
#endif
        if (m->fused_qkv_attn) {
#ifdef CV_BUILD_EXPERIMENTS
            float* qn = m->newtok.as<float>(); float* kn = qn + c.heads * 64; float* vn = kn + c.kv_heads * 64;
            QkvAttnArgs qa{L.wqkv, L.bqkv, h, L.ln1, c.rms_eps, c.hidden, m->kcache.as<float>() + m->layer_cache() * i, m->vcache.as<float>() + m->layer_cache() * i,
                           m->rope_cos.as<float>(), m->rope_sin.as<float>(), c.heads, c.kv_heads, c.max_len, st, m->attn_part.as<float>(), nsp, qn, kn, vn};
            // profiling category 0 (qkv) carries the fused launch; category 1 (attention) stays empty in this mode
            if (want(0)) { ProfScope ps(m, s, 0); hipLaunchKernelGGL((qkv_attn_kernel<7>), dim3(c.heads * nsp + 2 * c.kv_heads), dim3(256), 0, s, qa); }
            go.qnew = qn; go.knew = kn; go.vnew = vn; go.kv_group = c.heads / c.kv_heads;
#endif
        } else {
            GemvArgs gq{L.wqkv, L.bqkv, h, qkv, m->qkv_dim, c.hidden, L.ln1, c.rms_eps, nullptr, 0, st};
            if (m->prefetch == 2 && pgu.p) { gq.pf = pgu; gq.pf.first = round8((m->qkv_dim + 15) / 16); }
            if (want(0)) { ProfScope ps(m, s, 0); gemv(gq, 1, s); }
            AttnDecodeArgs ad{qkv, m->kcache.as<float>() + m->layer_cache() * i, m->vcache.as<float>() + m->layer_cache() * i,
                              m->rope_cos.as<float>(), m->rope_sin.as<float>(), c.heads, c.kv_heads, c.max_len, st,
                              m->attn_part.as<float>(), nsp};
            int attn_grid = c.heads * nsp;
            if (pgu.p) {
                ad.pf = m->prefetch == 2 ? pdn : pgu; ad.pf.first = round8(attn_grid);
                attn_grid = ad.pf.first + prefetch_groups(ad.pf.cons_bytes, 1, 14) * ad.pf.stride;
            }
            if (want(1)) { ProfScope ps(m, s, 1); hipLaunchKernelGGL(attn_decode_kernel, dim3(attn_grid), dim3(64), 0, s, ad); }
            if (m->prefetch == 1 && pdn.p) { go.pf = pdn; go.pf.first = round8((c.hidden + 3) / 4); }
        }
        if (want(2)) { ProfScope ps(m, s, 2); gemv(go, 1, s, nsp); }
        if (want(3)) { ProfScope ps(m, s, 3); gemv(GemvArgs{L.wgu, nullptr, h, act, 2 * c.inter, c.hidden, L.ln2, c.rms_eps, nullptr, 1, st}, 2, s); }
        GemvArgs gd{L.wdown, nullptr, act, h, c.hidden, c.inter, nullptr, 0.f, h, 0, st};
        if (i == c.layers - 1 && m->only_cat < 0) gd.advance = m->state.as<DecodeState>();     // the step's last kernel also advances the KV length
        if (want(4)) { ProfScope ps(m, s, 4); gemv(gd, 1, s); }
    }
}

static bool same_sampling(const cv_sampling& a, const cv_sampling& b) {
    return a.mode == b.mode && a.eos == b.eos && a.n_stop == b.n_stop && a.min_len == b.min_len && a.max_len == b.max_len &&
           a.top_p == b.top_p && a.top_k == b.top_k && a.win_size == b.win_size && a.tau_r == b.tau_r && a.seed == b.seed &&
           a.use_uniforms == b.use_uniforms;
}

static void llm_decode(cv_llm* m, int n_steps, const cv_sampling* sp, int32_t* out_tokens, int32_t* n_out, int32_t* finished, hipStream_t s) {
    CV_CHECK(m->finalized, "llm: call cv_llm_finalize first");
    CV_CHECK(sp && sp->max_len > 0 && sp->eos >= 0 && sp->eos + sp->n_stop <= m->V, "llm_decode: bad sampling parameters");
    CV_CHECK(sp->mode == 0 || (sp->top_k > 0 && sp->top_k <= 64 && sp->win_size >= 0), "llm_decode: bad RAS parameters");
    if (!m->sp_valid || !same_sampling(m->sp, *sp)) {          // request parameters go to device memory; the graph stays valid
        m->sp = *sp; m->sp_valid = true;
        CV_HIP(hipStreamSynchronize(s));                        // host_sp may still be the source of an earlier async copy
        *m->host_sp = SampleParams{sp->mode, sp->eos, sp->n_stop, sp->min_len, sp->max_len, sp->top_p, sp->top_k, sp->win_size, sp->tau_r,
                                   sp->use_uniforms, (unsigned long long)sp->seed};
        CV_HIP(hipMemcpyAsync(m->sparams.p, m->host_sp, sizeof(SampleParams), hipMemcpyHostToDevice, s));
    }
    const int before = m->host_state->n_tokens;
    // never run past the KV cache: each step appends one position
    CV_CHECK(m->host_state->pos + n_steps < m->cfg.max_len, "llm_decode: KV cache (max_len) exhausted");
    if (m->use_graph) {
        if (!m->graph || m->graph_stream != s) {
            if (m->graph) { (void)hipGraphExecDestroy(m->graph); m->graph = nullptr; }
            std::lock_guard<std::recursive_mutex> lk(runtime_lock());
            hipGraph_t g = capture_graph(s, [&] { llm_enqueue_step(m, s); });
            CV_HIP(hipGraphInstantiate(&m->graph, g, nullptr, nullptr, 0));
            CV_HIP(hipGraphDestroy(g));
            m->graph_stream = s;
        }
        { std::lock_guard<std::recursive_mutex> lk(runtime_lock()); for (int i = 0; i < n_steps; ++i) CV_HIP(hipGraphLaunch(m->graph, s)); }
    } else {
        std::lock_guard<std::recursive_mutex> lk(runtime_lock());
        for (int i = 0; i < n_steps; ++i) llm_enqueue_step(m, s);
    }
    CV_HIP(hipMemcpyAsync(m->host_state, m->state.p, sizeof(DecodeState), hipMemcpyDeviceToHost, s));
    CV_HIP(hipMemcpyAsync(m->host_tokens, m->tokens.p, (size_t)m->cfg.max_len * sizeof(int), hipMemcpyDeviceToHost, s));
    CV_HIP(hipStreamSynchronize(s));
    CV_HIP(hipGetLastError());
    const int after = m->host_state->n_tokens;
    for (int i = before; i < after; ++i) out_tokens[i - before] = m->host_tokens[i];
    *n_out = after - before;
    *finished = m->host_state->done || m->host_state->step >= sp->max_len;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Lock-step batched decode: nb sequences, one token each per step, every weight matrix streamed once per step.
// A slot is filled by a normal (single-sequence) prefill whose KV prefix, last hidden state and loop state are then copied into it.
// ---------------------------------------------------------------------------------------------------------------------------------
static void batch_begin(cv_llm* m, int nb, hipStream_t s) {
    CV_CHECK(m->finalized, "llm: call cv_llm_finalize first");
    CV_CHECK(nb >= 1 && nb <= MAX_NB, "cv_llm_batch_begin: batch size must be 1..32");
    const auto& c = m->cfg; auto& b = m->bt;
    std::lock_guard<std::recursive_mutex> lk(runtime_lock());
    CV_HIP(hipStreamSynchronize(s));
    b.nb = nb;
    b.kcache.ensure((size_t)nb * m->slot_cache() * 4); b.vcache.ensure((size_t)nb * m->slot_cache() * 4);
    b.state.ensure((size_t)nb * sizeof(DecodeState)); b.tokens.ensure((size_t)nb * c.max_len * sizeof(int));
    b.sparams.ensure((size_t)nb * sizeof(SampleParams)); b.uniforms.ensure((size_t)nb * 2 * c.max_len * 4);
    b.h.ensure((size_t)nb * c.hidden * 4); b.qkv.ensure((size_t)nb * m->qkv_dim * 4); b.act.ensure((size_t)nb * c.inter * 4);
    b.logits.ensure((size_t)nb * m->V * 4); b.attn.ensure((size_t)nb * c.heads * 64 * 4); b.dpart.ensure((size_t)8 * nb * c.hidden * 4); b.apart.ensure((size_t)nb * c.heads * 8 * ATTN_PART * 4);   // apart: key-slice partials of the batched attention (<= 8 slices)
    b.host_state.assign(nb, DecodeState{}); b.host_tokens.assign((size_t)nb * c.max_len, 0); b.host_sp.assign(nb, SampleParams{});
    for (auto& st : b.host_state) { st.done = 1; st.stop_token = -1; }           // empty slots are "finished"
    CV_HIP(hipMemcpyAsync(b.state.p, b.host_state.data(), (size_t)nb * sizeof(DecodeState), hipMemcpyHostToDevice, s));
    CV_HIP(hipStreamSynchronize(s));
    if (b.graph_nb != nb) b.drop_graphs();
    ensure_packed(m, s);
}

static void batch_prefill(cv_llm* m, int slot, const float* lm_input, int L0, const cv_sampling* sp, hipStream_t s) {
    auto& b = m->bt; const auto& c = m->cfg;
    CV_CHECK(slot >= 0 && slot < b.nb, "cv_llm_batch_prefill: slot out of range (call cv_llm_batch_begin first)");
    CV_CHECK(sp && sp->max_len > 0 && sp->eos >= 0 && sp->eos + sp->n_stop <= m->V, "cv_llm_batch_prefill: bad sampling parameters");
    CV_CHECK(sp->mode == 0 || (sp->top_k > 0 && sp->top_k <= 64 && sp->win_size >= 0), "cv_llm_batch_prefill: bad RAS parameters");
    CV_CHECK(!sp->use_uniforms, "cv_llm_batch_prefill: injected uniforms are a single-sequence parity hook");
    llm_prefill(m, lm_input, L0, s);                              // the validated single-sequence path; leaves K/V in the handle's own cache
    const size_t row = 64 * sizeof(float);
    for (int i = 0; i < c.layers; ++i)
        for (int g = 0; g < c.kv_heads; ++g) {
            const size_t src = m->layer_cache() * i + (size_t)g * c.max_len * 64;
            const size_t dst = (size_t)slot * m->slot_cache() + src;
            CV_HIP(hipMemcpyAsync(b.kcache.as<float>() + dst, m->kcache.as<float>() + src, (size_t)L0 * row, hipMemcpyDeviceToDevice, s));
            CV_HIP(hipMemcpyAsync(b.vcache.as<float>() + dst, m->vcache.as<float>() + src, (size_t)L0 * row, hipMemcpyDeviceToDevice, s));
        }
    CV_HIP(hipMemcpyAsync(b.h.as<float>() + (size_t)slot * c.hidden, m->h.p, (size_t)c.hidden * 4, hipMemcpyDeviceToDevice, s));
    b.host_state[slot] = *m->host_state;                          // pos = L0, step = 0, done = 0
    b.host_sp[slot] = SampleParams{sp->mode, sp->eos, sp->n_stop, sp->min_len, sp->max_len, sp->top_p, sp->top_k, sp->win_size, sp->tau_r, 0,
                                   (unsigned long long)sp->seed};
    CV_HIP(hipMemcpyAsync(b.state.as<DecodeState>() + slot, &b.host_state[slot], sizeof(DecodeState), hipMemcpyHostToDevice, s));
    CV_HIP(hipMemcpyAsync(b.sparams.as<SampleParams>() + slot, &b.host_sp[slot], sizeof(SampleParams), hipMemcpyHostToDevice, s));
    CV_HIP(hipStreamSynchronize(s));
}

// Several slots filled by ONE prefill pass: the prompts are stacked row-wise (rows_in: [sum L0][hidden], slot j's rows follow slot j-1's), every
// GEMM runs once over all rows, K / V go straight into the slots' caches.  Arithmetic per row = the single-sequence prefill's.
static void batch_prefill_many(cv_llm* m, int n, const int32_t* slots, const float* rows_in, const int32_t* L0s, const cv_sampling* sps, hipStream_t s) {
    auto& b = m->bt; const auto& c = m->cfg;
    CV_CHECK(m->finalized && n >= 1 && n <= b.nb && slots && rows_in && L0s && sps, "cv_llm_batch_prefill_many: bad arguments (call cv_llm_batch_begin first)");
    std::vector<PrefillSeg> segs; int R = 0;
    for (int j = 0; j < n; ++j) {
        const cv_sampling* sp = sps + j;
        CV_CHECK(slots[j] >= 0 && slots[j] < b.nb, "cv_llm_batch_prefill_many: slot out of range");
        for (int k = 0; k < j; ++k) CV_CHECK(slots[k] != slots[j], "cv_llm_batch_prefill_many: a slot is listed twice");
        CV_CHECK(L0s[j] > 0 && L0s[j] < c.max_len, "cv_llm_batch_prefill_many: prompt length out of range (KV capacity max_len)");
        CV_CHECK(sp->max_len > 0 && sp->eos >= 0 && sp->eos + sp->n_stop <= m->V && !sp->use_uniforms, "cv_llm_batch_prefill_many: bad sampling parameters");
        CV_CHECK(sp->mode == 0 || (sp->top_k > 0 && sp->top_k <= 64 && sp->win_size >= 0), "cv_llm_batch_prefill_many: bad RAS parameters");
        segs.push_back(PrefillSeg{R, L0s[j], 0, b.kcache.as<float>() + (size_t)slots[j] * m->slot_cache(), b.vcache.as<float>() + (size_t)slots[j] * m->slot_cache()});
        R += L0s[j];
    }
    llm_prefill_rows(m, rows_in, R, segs, s);
    for (int j = 0; j < n; ++j) {
        const int slot = slots[j]; const cv_sampling* sp = sps + j;
        CV_HIP(hipMemcpyAsync(b.h.as<float>() + (size_t)slot * c.hidden, m->pf_x.as<float>() + (size_t)(segs[j].row0 + segs[j].L - 1) * c.hidden,
                              (size_t)c.hidden * 4, hipMemcpyDeviceToDevice, s));
        DecodeState st{}; st.pos = L0s[j]; st.stop_token = -1;
        b.host_state[slot] = st;
        b.host_sp[slot] = SampleParams{sp->mode, sp->eos, sp->n_stop, sp->min_len, sp->max_len, sp->top_p, sp->top_k, sp->win_size, sp->tau_r, 0,
                                       (unsigned long long)sp->seed};
        CV_HIP(hipMemcpyAsync(b.state.as<DecodeState>() + slot, &b.host_state[slot], sizeof(DecodeState), hipMemcpyHostToDevice, s));
        CV_HIP(hipMemcpyAsync(b.sparams.as<SampleParams>() + slot, &b.host_sp[slot], sizeof(SampleParams), hipMemcpyHostToDevice, s));
    }
    CV_HIP(hipStreamSynchronize(s));
}

// skinny GEMM of the batched decode (llm_batch_kernels.h): rt = row tiles (16 rows) per workgroup, ksplit = K ranges across workgroups
static void skinny(const SkinnyArgs& a, int rt, hipStream_t s, const bf16_t* wp = nullptr) {
    const int tiles = a.K / 32 / a.ksplit, row_tiles = (a.N + 15) / 16;
    if (wp && tiles >= 4 && (tiles + 3) / 4 <= 7) {          // fragment-ordered weights + wave-private activation staging (round 3); same arithmetic, same bits
        CV_CHECK(a.K % (32 * a.ksplit) == 0 && (a.mode == 0 || a.N % 4 == 0), "skinny: K range must be a multiple of 32");
        CV_CHECK(!(a.gamma && a.ksplit != 1) && (a.mode == 2) == (a.ksplit > 1), "skinny: split-K workgroups leave raw partials (mode 2), the fused norm needs the whole row");
        SkinnyArgs b = a; b.W = wp;
        const dim3 grid(((row_tiles + rt - 1) / rt) * a.ksplit);
        const bool deep = (tiles + 3) / 4 > 5;
        if (a.nb > 16) {                                       // 17 .. 32 sequences: two MFMA column tiles per weight fragment (skinny_pk2_kernel)
            if (rt == 1) hipLaunchKernelGGL((skinny_pk2_kernel<1, 7>), grid, dim3(256), 0, s, b);
            else if (rt == 4) { CV_CHECK(deep || tiles <= 28, "skinny: four row tiles per workgroup are for K ranges of up to 28 tiles"); hipLaunchKernelGGL((skinny_pk2_kernel<4, 7>), grid, dim3(256), 0, s, b); }
            else if (deep) hipLaunchKernelGGL((skinny_pk2_kernel<2, 7>), grid, dim3(256), 0, s, b);
            else hipLaunchKernelGGL((skinny_pk2_kernel<2, 5>), grid, dim3(256), 0, s, b);
            return;
        }
        if (rt == 1) hipLaunchKernelGGL((skinny_pk_kernel<1, 7>), grid, dim3(256), 0, s, b);
        else if (rt == 4) { CV_CHECK(deep || tiles <= 28, "skinny: four row tiles per workgroup are for K ranges of up to 28 tiles"); hipLaunchKernelGGL((skinny_pk_kernel<4, 7>), grid, dim3(256), 0, s, b); }
        else if (deep) hipLaunchKernelGGL((skinny_pk_kernel<2, 7>), grid, dim3(256), 0, s, b);
        else hipLaunchKernelGGL((skinny_pk_kernel<2, 5>), grid, dim3(256), 0, s, b);
        return;
    }
    CV_CHECK(a.nb <= 16, "skinny: more than 16 sequences per step need the fragment-ordered weight copies (K ranges of 4 .. 28 tiles per workgroup)");
    CV_CHECK(a.K % (32 * a.ksplit) == 0 && tiles >= 1 && (tiles + 3) / 4 <= 7 && (a.mode == 0 || a.N % 4 == 0), "skinny: K range must be a multiple of 32 and at most 28 tiles per workgroup");
    CV_CHECK(!(a.gamma && a.ksplit != 1) && (a.mode == 2) == (a.ksplit > 1), "skinny: split-K workgroups leave raw partials (mode 2), the fused norm needs the whole row");
    const dim3 grid(((row_tiles + rt - 1) / rt) * a.ksplit);
    const bool deep = (tiles + 3) / 4 > 5;
    // CV_SKINNY_X3=0: the products on the fp32 matrix pipe (8 v_mfma_f32_16x16x4_f32 per tile) instead of the exact three-term bf16 split (A/B knob)
    static const bool x3 = [] { const char* e = getenv("CV_SKINNY_X3"); return !(e && e[0] == '0'); }();
    if (x3) {
        if (rt == 1) hipLaunchKernelGGL((skinny_mfma_kernel<1, 7, true>), grid, dim3(256), 0, s, a);
        else if (deep) hipLaunchKernelGGL((skinny_mfma_kernel<2, 7, true>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((skinny_mfma_kernel<2, 5, true>), grid, dim3(256), 0, s, a);
    } else {
        if (rt == 1) hipLaunchKernelGGL((skinny_mfma_kernel<1, 7, false>), grid, dim3(256), 0, s, a);
        else if (deep) hipLaunchKernelGGL((skinny_mfma_kernel<2, 7, false>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((skinny_mfma_kernel<2, 5, false>), grid, dim3(256), 0, s, a);
    }
}

static void skinny_f8(const SkinnyF8Args& a, int rt, hipStream_t s) {
    const int tiles = a.K / 64 / a.ksplit, row_tiles = (a.N + 15) / 16;
    CV_CHECK(a.K % (64 * a.ksplit) == 0 && tiles >= 1 && (tiles + 3) / 4 <= 5 && (a.mode == 0 || a.N % 4 == 0), "skinny_f8: K range must be a multiple of 64 and at most 20 tiles per workgroup");
    CV_CHECK(!(a.gamma && a.ksplit != 1) && (a.mode == 2) == (a.ksplit > 1), "skinny_f8: split-K workgroups leave raw partials (mode 2), the fused norm needs the whole row");
    const dim3 grid(((row_tiles + rt - 1) / rt) * a.ksplit);
    const bool deep = (tiles + 3) / 4 > 4;
    CV_CHECK(!(a.gamma && deep), "skinny_f8: the fused norm covers rows of at most 16 tiles (K <= 1024)");
    if (a.gamma) {
        if (rt == 1) hipLaunchKernelGGL((skinny_fp8_kernel<1, 4, true>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((skinny_fp8_kernel<2, 4, true>), grid, dim3(256), 0, s, a);
    } else {
        if (rt == 1) hipLaunchKernelGGL((skinny_fp8_kernel<1, 4, false>), grid, dim3(256), 0, s, a);
        else if (deep) hipLaunchKernelGGL((skinny_fp8_kernel<2, 5, false>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((skinny_fp8_kernel<2, 4, false>), grid, dim3(256), 0, s, a);
    }
}
// K ranges of the fp8 down projection (64-column tiles): CosyVoice2 / 3: 76 tiles -> 4 x 19
static int down_ksplit_f8(int inter) {
    const int tiles = inter / 64;
    for (int ks = 8; ks > 1; ks >>= 1) if (tiles % ks == 0 && tiles / ks >= 2 && (tiles / ks + 3) / 4 <= 5) return ks;
    return 1;
}

// K ranges of the down projection: the largest split <= 8 that keeps >= 2 k-tiles per workgroup (CosyVoice2: 152 tiles -> 8 x 19)
static int down_ksplit(int inter) {
    const int tiles = inter / 32;
    for (int ks = 8; ks > 1; ks >>= 1) if (tiles % ks == 0 && tiles / ks >= 2) return ks;
    return 1;
}

// 4 waves = 192 keys per pass.  8 waves (384 keys: the whole context of a 10 s utterance in one load round) measured SLOWER: LM step 1013 -> 1111 us at 8
// sequences (profiles/r2_batch_decode_ab.txt) - twice the waves per (head, sequence) cost more in the merge and in CU occupancy than the second round does.
// Round 4: CV_ATTN_BATCH_GQA=1 (A/B knob, read when a step is enqueued / captured): one workgroup per (sequence, kv head) walks all heads of the group over one load of
// the K / V rows (attn_decode_batch_gqa_kernel).  Bit-identical, measured SLOWER on the MI355X and therefore off (profiles/r4_batch_serving_ab.txt: LM step 896 -> 1133 us
// at 16 slots, 1396 -> 1486 at 32; the kernel 24.1 -> 29.9 us per launch in the mixed workload): 7 x the arithmetic of a head on a seventh of the workgroups - the
// softmax's exp / DPP work per loaded key, not the L2 traffic, is what the launch is made of once the heads of a group share an XCD (the remap in the kernel above).
// Round 5 (default): attn_decode_batch_mfma_kernel - workgroup = (sequence, kv head, key slice), the group's heads as the columns of the fp32 MFMA, K / V read once
// per pair; nslice > 1 adds attn_merge_batch_kernel.  Slices: enough workgroups for one round over the 256 CUs (4-wave workgroups: 4 slices at 32 sequences x 2 kv
// heads, 8 at 16).  Which form runs is decided per decode call (batch_decode: slot count and longest context); A/B knobs read when a step is captured:
// CV_ATTN_BATCH_SLICES=1..8, CV_ATTN_BATCH_WAVES=4|8.
static void launch_attn_batch(AttnDecodeBatchArgs ad, int heads, int nb, hipStream_t s, float* part, int mode) {
    ad.nb = nb;
    const int gsz = heads / ad.kv_heads;
    if (mode >= 1 && part && gsz >= 1 && gsz <= 16 && heads % ad.kv_heads == 0) {
        const int pairs = nb * ad.kv_heads;
        int nw = [] { const char* e = getenv("CV_ATTN_BATCH_WAVES"); return (e && atoi(e) == 8) ? 8 : 4; }();
        int S = [] { const char* e = getenv("CV_ATTN_BATCH_SLICES"); return e ? atoi(e) : 0; }();
        if (S <= 0) { S = 1; while (S < 8 && pairs * S * (nw / 4) < 256) S <<= 1; }
        S = std::min(std::max(S, 1), 8);
        // mode 2 (option batch_attn = 2, "invariant"): ONE partition of a sequence's keys whatever the slot count - 8 slices of 4-wave workgroups - so that a sequence's
        // logits are the same bits in a batch of 1 and of 32, in one chain and in a cut batch (the slice count otherwise follows the slot count: 8 up to 16 slots, 4 at 32)
        if (mode == 2) { S = 8; nw = 4; }
        ad.part = part; ad.nslice = S;
        const dim3 grid((unsigned)(pairs * S));
        if (nw == 8) hipLaunchKernelGGL(attn_decode_batch_mfma_kernel<8>, grid, dim3(512), 0, s, ad);
        else hipLaunchKernelGGL(attn_decode_batch_mfma_kernel<4>, grid, dim3(256), 0, s, ad);
        if (S > 1) hipLaunchKernelGGL(attn_merge_batch_kernel, dim3((unsigned)((nb * heads * 16 + 255) / 256)), dim3(256), 0, s, ad);
        return;
    }
#ifdef CV_BUILD_EXPERIMENTS
    const bool gqa = [] { const char* e = getenv("CV_ATTN_BATCH_GQA"); return e && e[0] == '1'; }();
    if (gqa && gsz >= 2 && gsz <= 8 && heads % ad.kv_heads == 0) {
        const dim3 grid((unsigned)(ad.kv_heads * nb));
        if (gsz <= 2) hipLaunchKernelGGL(attn_decode_batch_gqa_kernel<1>, grid, dim3(512), 0, s, ad);
        else if (gsz <= 4) hipLaunchKernelGGL(attn_decode_batch_gqa_kernel<2>, grid, dim3(512), 0, s, ad);
        else hipLaunchKernelGGL(attn_decode_batch_gqa_kernel<4>, grid, dim3(512), 0, s, ad);
        return;
    }
#endif
    hipLaunchKernelGGL(attn_decode_batch_kernel<4>, dim3((unsigned)(heads * nb)), dim3(256), 0, s, ad);
}

// one token for every slot: the launch sequence of llm_enqueue_step on the multi-sequence kernels
static void batch_enqueue_step(cv_llm* m, hipStream_t s) {
    const auto& c = m->cfg; auto& b = m->bt; const int nb = b.nb;
    const long long H = c.hidden, Q = m->qkv_dim, I = c.inter, V = m->V, A = c.heads * 64;
    DecodeState* st = b.state.as<DecodeState>();
    float* h = b.h.as<float>(); float* qkv = b.qkv.as<float>(); float* act = b.act.as<float>(); float* logits = b.logits.as<float>();
    float* att = b.attn.as<float>(); float* dpart = b.dpart.as<float>();
    if (m->batch_fp8) {                                            // opt-in fp8 weights + activations (llm_batch_kernels.h, skinny_fp8_kernel)
        CV_CHECK(nb <= 16, "llm: the fp8 batched decode takes at most 16 sequences per step");
        CV_CHECK(m->have_fp8, "llm: option batch_fp8 needs the '<name>.f8' / '<name>.f8s' tensors (Qwen2LM(..., batch_fp8=True))");
        const int k8 = down_ksplit_f8(c.inter);
        skinny_f8(SkinnyF8Args{m->head_f8.w, m->head_f8.s, m->head_b, h, H, logits, V, (int)V, c.hidden, m->norm, c.rms_eps, nullptr, 0, 0, nb, 1}, 2, s);
        {
            SampleArgs sa{};
            sa.logits = logits; sa.V = (int)V; sa.sp = b.sparams.as<SampleParams>(); sa.uniforms = b.uniforms.as<float>();
            sa.st = st; sa.tokens = b.tokens.as<int>(); sa.max_tokens = c.max_len;
            sa.emb_table = m->speech_emb; sa.emb_dim = c.hidden; sa.h_out = h;
            sa.slot_logits = V; sa.slot_uniforms = 2LL * c.max_len; sa.slot_tokens = c.max_len; sa.slot_h = H;
            hipLaunchKernelGGL(sample_kernel, dim3(nb), dim3(1024), 0, s, sa);
        }
        for (int l = 0; l < c.layers; ++l) {
            const auto& L = m->layers[l]; const auto& F = m->layers_f8[l];
            skinny_f8(SkinnyF8Args{F.qkv.w, F.qkv.s, L.bqkv, h, H, qkv, Q, (int)Q, c.hidden, L.ln1, c.rms_eps, nullptr, 0, 0, nb, 1}, 1, s);
            AttnDecodeBatchArgs ad{qkv, Q, b.kcache.as<float>() + m->layer_cache() * l, b.vcache.as<float>() + m->layer_cache() * l, (long long)m->slot_cache(),
                                   m->rope_cos.as<float>(), m->rope_sin.as<float>(), c.heads, c.kv_heads, c.max_len, st, att, A};
            launch_attn_batch(ad, c.heads, nb, s, b.apart.as<float>(), b.attn_mode);
            skinny_f8(SkinnyF8Args{F.o.w, F.o.s, nullptr, att, A, h, H, c.hidden, (int)A, nullptr, 0.f, h, H, 0, nb, 1}, 1, s);
            skinny_f8(SkinnyF8Args{F.gu.w, F.gu.s, nullptr, h, H, act, I, 2 * c.inter, c.hidden, L.ln2, c.rms_eps, nullptr, 0, 1, nb, 1}, 2, s);
            if (k8 > 1) {
                skinny_f8(SkinnyF8Args{F.down.w, F.down.s, nullptr, act, I, dpart, H, c.hidden, c.inter, nullptr, 0.f, nullptr, 0, 2, nb, k8}, 2, s);
                hipLaunchKernelGGL(sum_partials_kernel, dim3((unsigned)((nb * H / 4 + 255) / 256)), dim3(256), 0, s, dpart, k8, nb, c.hidden, h, H, h, H);
            } else {
                skinny_f8(SkinnyF8Args{F.down.w, F.down.s, nullptr, act, I, h, H, c.hidden, c.inter, nullptr, 0.f, h, H, 0, nb, 1}, 2, s);
            }
        }
        hipLaunchKernelGGL(advance_pos_batch_kernel, dim3(1), dim3(64), 0, s, st, nb);
        return;
    }
    int ks = down_ksplit(c.inter);
    if (nb > 16) while (ks > 1 && c.inter / 32 / ks < 4) ks >>= 1;     // more than 16 sequences run on the fragment-ordered weights only: K ranges of >= 4 tiles (small test models)
    // CV_DOWN_DEEP=1: the down projection as ONE launch of 56 sixteen-wave workgroups over the whole K (skinny_deep_kernel) instead of 8 K ranges across
    // 448 workgroups + sum_partials_kernel.  Measured on MI355X (profiles/r3_batch_decode_ab.txt): 1086 vs 975 us per 8-sequence step - the weight
    // stream comes from HBM at ~25 GB/s per CU, so 56 CUs cannot pull 8.7 MB in the time 448 workgroups on 256 CUs do; the removed launch (4.8 us) does
    // not pay for that.  Kept, tested, off.
    const bool deep_knob = kExperiments && [] { const char* e = getenv("CV_DOWN_DEEP"); return e && e[0] == '1'; }();       // read when a step is enqueued / captured
    const bool deep_down = deep_knob && nb <= 16 && (c.inter / 32 + 15) / 16 <= 10 && c.hidden % 4 == 0;
    // 2 row tiles per workgroup for the two wide GEMMs (gate/up: 304 workgroups, head: 206); 3 (203 / 137 workgroups, at most 3 tiles per CU instead
    // of 4 on 48 CUs) measured the same step time (profiles/r2_batch_decode_ab.txt): the launch is not bound by the busiest CU's MFMA share
    const int wide_rt = 2;
    skinny(SkinnyArgs{m->head_w, m->head_b, h, H, logits, V, (int)V, c.hidden, m->norm, c.rms_eps, nullptr, 0, 0, nb, 1}, wide_rt, s, m->pk(m->head_w));
    {                                                             // every slot's sampler + embedding of the sampled token: one launch, one workgroup per slot
        SampleArgs sa{};
        sa.logits = logits; sa.V = (int)V; sa.sp = b.sparams.as<SampleParams>(); sa.uniforms = b.uniforms.as<float>();
        sa.st = st; sa.tokens = b.tokens.as<int>(); sa.max_tokens = c.max_len;
        sa.emb_table = m->speech_emb; sa.emb_dim = c.hidden; sa.h_out = h;
        sa.slot_logits = V; sa.slot_uniforms = 2LL * c.max_len; sa.slot_tokens = c.max_len; sa.slot_h = H;
        hipLaunchKernelGGL(sample_kernel, dim3(nb), dim3(1024), 0, s, sa);
    }
    for (int l = 0; l < c.layers; ++l) {
        const auto& L = m->layers[l];
        skinny(SkinnyArgs{L.wqkv, L.bqkv, h, H, qkv, Q, (int)Q, c.hidden, L.ln1, c.rms_eps, nullptr, 0, 0, nb, 1}, 1, s, m->pk(L.wqkv));
        AttnDecodeBatchArgs ad{qkv, Q, b.kcache.as<float>() + m->layer_cache() * l, b.vcache.as<float>() + m->layer_cache() * l, (long long)m->slot_cache(),
                               m->rope_cos.as<float>(), m->rope_sin.as<float>(), c.heads, c.kv_heads, c.max_len, st, att, A};
        launch_attn_batch(ad, c.heads, nb, s, b.apart.as<float>(), b.attn_mode);
        skinny(SkinnyArgs{L.wo, nullptr, att, A, h, H, c.hidden, (int)A, nullptr, 0.f, h, H, 0, nb, 1}, 1, s, m->pk(L.wo));
        // packed: four row tiles per workgroup (152 workgroups) measured 7.3 us against 8.0 for two and 8.7 for one (profiles/r3_skinny_probe.txt)
        skinny(SkinnyArgs{L.wgu, nullptr, h, H, act, I, 2 * c.inter, c.hidden, L.ln2, c.rms_eps, nullptr, 0, 1, nb, 1}, m->pk(L.wgu) ? 4 : wide_rt, s, m->pk(L.wgu));
#ifdef CV_BUILD_EXPERIMENTS
        if (deep_down) {                                          // one launch: 16-wave workgroups over the whole K (skinny_deep_kernel)
            hipLaunchKernelGGL((skinny_deep_kernel<16, 5>), dim3((unsigned)((H + 15) / 16)), dim3(1024), 0, s,
                               SkinnyArgs{L.wdown, nullptr, act, I, h, H, c.hidden, c.inter, nullptr, 0.f, h, H, 0, nb, 1});
        } else
#endif
        if (ks > 1) {
            skinny(SkinnyArgs{L.wdown, nullptr, act, I, dpart, H, c.hidden, c.inter, nullptr, 0.f, nullptr, 0, 2, nb, ks}, 2, s, m->pk(L.wdown));
            hipLaunchKernelGGL(sum_partials_kernel, dim3((unsigned)((nb * H / 4 + 255) / 256)), dim3(256), 0, s, dpart, ks, nb, c.hidden, h, H, h, H);
        } else {
            skinny(SkinnyArgs{L.wdown, nullptr, act, I, h, H, c.hidden, c.inter, nullptr, 0.f, h, H, 0, nb, 1}, 2, s, m->pk(L.wdown));
        }
    }
    hipLaunchKernelGGL(advance_pos_batch_kernel, dim3(1), dim3(64), 0, s, st, nb);
}

static void batch_decode(cv_llm* m, int n_steps, int32_t* out_tokens, int32_t* n_out, int32_t* finished, hipStream_t s) {
    auto& b = m->bt; const auto& c = m->cfg; const int nb = b.nb;
    CV_CHECK(nb > 0, "cv_llm_batch_decode: call cv_llm_batch_begin first");
    CV_CHECK(n_steps > 0 && out_tokens && n_out && finished, "cv_llm_batch_decode: bad arguments");
    std::vector<int> before(nb);
    for (int i = 0; i < nb; ++i) {
        before[i] = b.host_state[i].n_tokens;
        // a slot appends at most (its request's max_len - step) more positions: past that the sampler only marks it done
        const int left = std::max(0, std::min(n_steps, b.host_sp[i].max_len - b.host_state[i].step));
        CV_CHECK(b.host_state[i].done || b.host_state[i].pos + left < c.max_len, "cv_llm_batch_decode: KV cache (max_len) exhausted in a slot");
    }
    // Which decode attention (measured on the MI355X, profiles/r5_batch_decode_ab.txt section 5): the MFMA form (K / V once per (sequence, kv head), + a merge launch) wins
    // with many slots or long contexts, the per-head VALU form (no second launch) with few slots at short contexts - 16 slots: 885 vs 928 us per step at contexts
    // 130 - 380, 1089 vs 1061 at 443 - 693; 8 slots: 826 vs 871 and 963 vs 954; 32 slots: 1359 vs 1286 already at 130 - 380.  Decided per call from the slot count and
    // the longest live context (known on the host from the last hand-back); both forms yield the oracle's tokens.  Option "batch_attn" = 0 / 1 pins one form.
    int mode;
    {
        int longest = 0;
        for (int i = 0; i < nb; ++i) if (!b.host_state[i].done) longest = std::max(longest, b.host_state[i].pos);
        mode = (nb >= 24 || (nb >= 12 && longest >= 416) || longest >= 640) ? 1 : 0;
        if (m->batch_attn >= 0) mode = m->batch_attn;                                 // pinned (option "batch_attn" / CV_ATTN_BATCH at handle creation): 0, 1 or 2 (MFMA form, fixed slicing)
    }
    {
        std::lock_guard<std::recursive_mutex> lk(runtime_lock());
        if (m->use_graph) {
            if (b.graph_stream != s || b.graph_nb != nb) {
                b.drop_graphs();
                b.graph_stream = s; b.graph_nb = nb;
            }
            b.attn_mode = mode;                                  // (read by batch_enqueue_step during the capture)
            hipGraphExec_t& bgraph = b.graphs[mode];             // the other forms' graphs, if they were captured before, are kept
            if (!bgraph) {
                hipGraph_t g = capture_graph(s, [&] { batch_enqueue_step(m, s); });
                CV_HIP(hipGraphInstantiate(&bgraph, g, nullptr, nullptr, 0));
                CV_HIP(hipGraphDestroy(g));
            }
            for (int i = 0; i < n_steps; ++i) CV_HIP(hipGraphLaunch(bgraph, s));
        } else {
            b.attn_mode = mode;
            for (int i = 0; i < n_steps; ++i) batch_enqueue_step(m, s);
        }
    }
    CV_HIP(hipMemcpyAsync(b.host_state.data(), b.state.p, (size_t)nb * sizeof(DecodeState), hipMemcpyDeviceToHost, s));
    CV_HIP(hipMemcpyAsync(b.host_tokens.data(), b.tokens.p, (size_t)nb * c.max_len * sizeof(int), hipMemcpyDeviceToHost, s));
    CV_HIP(hipStreamSynchronize(s));
    CV_HIP(hipGetLastError());
    for (int i = 0; i < nb; ++i) {
        const int after = b.host_state[i].n_tokens;
        for (int t = before[i]; t < after; ++t) out_tokens[(size_t)i * n_steps + (t - before[i])] = b.host_tokens[(size_t)i * c.max_len + t];
        n_out[i] = after - before[i];
        finished[i] = b.host_state[i].done || b.host_state[i].step >= b.host_sp[i].max_len;
    }
}


extern "C" {

int cv_llm_create(cv_llm** out, const cv_llm_config* cfg) {
    return guarded([&] { CV_CHECK(out && cfg, "cv_llm_create: null argument"); auto* m = new cv_llm(); m->cfg = *cfg; *out = m; });
}
int cv_llm_set_tensor(cv_llm* m, const char* name, const void* dev_ptr, int32_t dtype, int64_t numel) {
    return guarded([&] { CV_CHECK(m, "null handle"); m->tm.set(name, dev_ptr, dtype, numel); });
}
int cv_llm_finalize(cv_llm* m) { return guarded([&] { CV_CHECK(m, "null handle"); llm_finalize(m); }); }
void cv_llm_destroy(cv_llm* m) {
    if (!m) return;
    // may be called from a garbage-collector finaliser on ANY thread while another thread drives a different handle: quiesce the
    // device and hold the runtime lock so stream / graph / buffer destruction never overlaps a capture, a launch burst or a realloc
    std::lock_guard<std::recursive_mutex> lk(runtime_lock());
    (void)hipDeviceSynchronize();
    delete m;
}
int cv_llm_set_option(cv_llm* m, const char* name, int32_t value) {
    return guarded([&] {
        CV_CHECK(m && name, "null argument");
        std::lock_guard<std::recursive_mutex> lk(runtime_lock());
        if (std::string(name) == "use_graph") { m->use_graph = value != 0; if (m->graph) { (void)hipGraphExecDestroy(m->graph); m->graph = nullptr; } }
        else if (std::string(name) == "head_rows") { CV_CHECK(value == 1 || value == 2, "head_rows must be 1 or 2"); m->head_rows = value; if (m->graph) { (void)hipGraphExecDestroy(m->graph); m->graph = nullptr; } }
        else if (std::string(name) == "fused_attn_oproj" || std::string(name) == "oproj_rblocks" || std::string(name) == "oproj_waves") {
            CV_CHECK(std::string(name) != "oproj_rblocks" || value == 4 || value == 8, "oproj_rblocks must be 4 or 8");
            CV_CHECK(std::string(name) != "oproj_waves" || value == 8 || value == 16, "oproj_waves must be 8 or 16");
            need_experiments(std::string(name) == "fused_attn_oproj" && value != 0, "llm option fused_attn_oproj");
            (std::string(name) == "fused_attn_oproj" ? m->fused_attn_oproj : std::string(name) == "oproj_rblocks" ? m->oproj_rblocks : m->oproj_waves) = value;
            if (m->graph) { (void)hipGraphExecDestroy(m->graph); m->graph = nullptr; }
        }
        else if (std::string(name) == "prefill_rows") m->prefill_rows = value != 0;
        else if (std::string(name) == "head_waves") { CV_CHECK(value == 4 || value == 7, "head_waves must be 4 or 7"); m->head_waves = value; if (m->graph) { (void)hipGraphExecDestroy(m->graph); m->graph = nullptr; } }
        else if (std::string(name) == "fused_qkv_attn") { need_experiments(value != 0, "llm option fused_qkv_attn"); m->fused_qkv_attn = value != 0; if (m->graph) { (void)hipGraphExecDestroy(m->graph); m->graph = nullptr; } }
        else if (std::string(name) == "prefetch" || std::string(name) == "prefetch_shift") {
            CV_CHECK(std::string(name) != "prefetch" || (value >= 0 && value <= 2), "prefetch must be 0, 1 or 2");
            need_experiments(std::string(name) == "prefetch" && value != 0, "llm option prefetch");
            (std::string(name) == "prefetch" ? m->prefetch : m->prefetch_shift) = value; if (m->graph) { (void)hipGraphExecDestroy(m->graph); m->graph = nullptr; }
        }
        else if (std::string(name) == "attn_splits") {       // key-range slices per head in the decode attention (4, 8 or 16)
            CV_CHECK(value == 4 || value == 8 || value == 16, "attn_splits must be 4, 8 or 16");
            m->attn_splits = value; if (m->graph) { (void)hipGraphExecDestroy(m->graph); m->graph = nullptr; }
        }
        else if (std::string(name) == "batch_attn") {
            CV_CHECK(value >= -1 && value <= 2, "batch_attn must be -1 (by slot count and context), 0 (VALU form), 1 (MFMA form) or 2 (MFMA form, one key partition for every slot count)");
            m->batch_attn = value;                               // (every form keeps its own captured step: no graph is dropped)
        }
        else if (std::string(name) == "batch_packed") {      // batched decode on the fragment-ordered weight copies (skinny_pk_kernel) or on the row-major tensors (round 2)
            m->batch_packed = value != 0; m->bt.drop_graphs();
        }
        else if (std::string(name) == "batch_fp8") {         // batched decode on the fp8 copies of the weights (needs the .f8 / .f8s tensors)
            CV_CHECK(value == 0 || m->have_fp8, "batch_fp8: the fp8 tensors were not registered");
            m->batch_fp8 = value != 0; m->bt.drop_graphs();
        }
        else throw Error(std::string("unknown option ") + name);
    });
}
int cv_llm_prefill(cv_llm* m, const float* lm_input, int32_t L0, void* stream) {
    return guarded([&] { CV_CHECK(m && lm_input, "null argument"); llm_prefill(m, lm_input, L0, resolve(m, stream)); });
}
int cv_llm_prefill_append(cv_llm* m, const float* rows, int32_t n_rows, void* stream) {
    return guarded([&] { CV_CHECK(m && rows, "null argument"); llm_prefill(m, rows, n_rows, resolve(m, stream), true); });
}
int cv_llm_push_token(cv_llm* m, int32_t token, void* stream) {
    return guarded([&] {
        CV_CHECK(m && m->finalized, "null argument");
        hipStream_t s = resolve(m, stream);
        const int n = m->host_state->n_tokens;
        CV_CHECK(n < m->cfg.max_len, "cv_llm_push_token: token history full");
        m->host_tokens[n] = token;
        CV_HIP(hipMemcpyAsync(m->tokens.as<int>() + n, m->host_tokens + n, sizeof(int), hipMemcpyHostToDevice, s));
        m->host_state->n_tokens = n + 1;
        CV_HIP(hipMemcpyAsync(m->state.p, m->host_state, sizeof(DecodeState), hipMemcpyHostToDevice, s));
        CV_HIP(hipStreamSynchronize(s));
    });
}
int cv_llm_last_stop_token(cv_llm* m) { return m ? m->host_state->stop_token : -1; }
int cv_llm_set_uniforms(cv_llm* m, const float* host_uniforms, int32_t n, void* stream) {
    return guarded([&] {
        CV_CHECK(m && host_uniforms && n > 0 && n <= m->cfg.max_len * 2, "cv_llm_set_uniforms: bad arguments");
        CV_HIP(hipMemcpyAsync(m->uniforms.p, host_uniforms, (size_t)n * 4, hipMemcpyHostToDevice, as_stream(stream)));
        CV_HIP(hipStreamSynchronize(as_stream(stream)));
    });
}
int cv_llm_decode(cv_llm* m, int32_t n_steps, const cv_sampling* sp, int32_t* out_tokens, int32_t* n_out, int32_t* finished, void* stream) {
    return guarded([&] { CV_CHECK(m && out_tokens && n_out && finished && n_steps > 0, "cv_llm_decode: bad arguments");
                         llm_decode(m, n_steps, sp, out_tokens, n_out, finished, resolve(m, stream)); });
}
/* Runs ONE decode step eagerly with a HIP-event pair around every kernel launch on `stream` and returns, per category
 * (0 qkv, 1 attention, 2 o_proj, 3 gate_up, 4 down, 5 head, 6 sample), the launch count and the summed duration in ms. */
int cv_llm_profile_step(cv_llm* m, const cv_sampling* sp, int32_t* counts8, float* ms8, void* stream) {
    return guarded([&] {
        CV_CHECK(m && m->finalized && sp && counts8 && ms8, "cv_llm_profile_step: bad arguments");
        hipStream_t s = resolve(m, stream);
        m->sp = *sp; m->sp_valid = true;
        CV_HIP(hipStreamSynchronize(s));
        *m->host_sp = SampleParams{sp->mode, sp->eos, sp->n_stop, sp->min_len, sp->max_len, sp->top_p, sp->top_k, sp->win_size, sp->tau_r,
                                   sp->use_uniforms, (unsigned long long)sp->seed};
        CV_HIP(hipMemcpyAsync(m->sparams.p, m->host_sp, sizeof(SampleParams), hipMemcpyHostToDevice, s));
        CV_CHECK(m->host_state->pos + 1 < m->cfg.max_len, "cv_llm_profile_step: KV cache exhausted");
        m->profiling = true; m->prof_events.clear();
        llm_enqueue_step(m, s);
        m->profiling = false;
        CV_HIP(hipMemcpyAsync(m->host_state, m->state.p, sizeof(DecodeState), hipMemcpyDeviceToHost, s));
        CV_HIP(hipStreamSynchronize(s));
        for (int i = 0; i < 8; ++i) { counts8[i] = 0; ms8[i] = 0.f; }
        for (auto& e : m->prof_events) {
            float ms = 0.f; (void)hipEventElapsedTime(&ms, e.second.first, e.second.second);
            counts8[e.first] += 1; ms8[e.first] += ms;
            (void)hipEventDestroy(e.second.first); (void)hipEventDestroy(e.second.second);
        }
        m->prof_events.clear();
    });
}
/* Duration of ONE kernel class as it runs inside the decode graph: captures a graph holding only the launches of `category` of
 * one decode step (one per layer, each on its own layer's weights and KV slice, so nothing is cache-warm that is not in the real
 * step), replays it `reps` times between a single HIP-event pair on `stream`.  A per-launch event pair costs ~3 us of its own on
 * a 3-7 us kernel; a dependent chain of the same launches does not.  Clobbers the decode activations (not pos / the KV prefix). */
int cv_llm_profile_chain(cv_llm* m, int32_t category, int32_t reps, float* total_ms, int32_t* launches, void* stream) {
    return guarded([&] {
        CV_CHECK(m && m->finalized && total_ms && launches && reps > 0 && category >= 0 && category <= 5, "cv_llm_profile_chain: bad arguments");
        hipStream_t s = resolve(m, stream);
        CV_CHECK(m->sp_valid, "cv_llm_profile_chain: run a decode / profile step first (sampling parameters)");
        std::lock_guard<std::recursive_mutex> lk(runtime_lock());
        CV_HIP(hipStreamSynchronize(s));
        hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
        struct CatScope { cv_llm* m; ~CatScope() { m->only_cat = -1; } } cat_scope{m};      // also on the exception paths below
        m->only_cat = category;
        g = capture_graph(s, [&] { llm_enqueue_step(m, s); });
        m->only_cat = -1;
        CV_HIP(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        hipEvent_t e0, e1; CV_HIP(hipEventCreate(&e0)); CV_HIP(hipEventCreate(&e1));
        for (int i = 0; i < 2; ++i) CV_HIP(hipGraphLaunch(ge, s));
        CV_HIP(hipEventRecord(e0, s));
        for (int i = 0; i < reps; ++i) CV_HIP(hipGraphLaunch(ge, s));
        CV_HIP(hipEventRecord(e1, s));
        CV_HIP(hipEventSynchronize(e1));
        CV_HIP(hipEventElapsedTime(total_ms, e0, e1));
        *launches = (category == 5 ? 1 : m->cfg.layers) * reps;
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
    });
}
int cv_llm_batch_begin(cv_llm* m, int32_t nb, void* stream) {
    return guarded([&] { CV_CHECK(m, "null handle"); batch_begin(m, nb, resolve(m, stream)); });
}
int cv_llm_batch_prefill(cv_llm* m, int32_t slot, const float* lm_input, int32_t L0, const cv_sampling* sp, void* stream) {
    return guarded([&] { CV_CHECK(m && lm_input, "null argument"); batch_prefill(m, slot, lm_input, L0, sp, resolve(m, stream)); });
}
int cv_llm_batch_prefill_many(cv_llm* m, int32_t n, const int32_t* slots, const float* rows, const int32_t* L0s, const cv_sampling* sps, void* stream) {
    return guarded([&] { CV_CHECK(m, "null handle"); batch_prefill_many(m, n, slots, rows, L0s, sps, resolve(m, stream)); });
}
/* test hook: one fp8 skinny GEMM (the kernel of the opt-in fp8 batched decode) on caller-provided operands */
int cv_skinny_fp8(const void* w8, const float* wscale, const float* bias, const float* x, int64_t ldx, float* y, int64_t ldy, int32_t N, int32_t K,
                  const float* gamma, float eps, const float* res, int64_t ldres, int32_t mode, int32_t nb, int32_t ksplit, int32_t rt, void* stream) {
    return guarded([&] {
        CV_CHECK(w8 && wscale && x && y && nb >= 1 && nb <= 16 && (rt == 1 || rt == 2), "cv_skinny_fp8: bad arguments");
        skinny_f8(SkinnyF8Args{reinterpret_cast<const unsigned char*>(w8), wscale, bias, x, ldx, y, ldy, N, K, gamma, eps, res, ldres, mode, nb, ksplit}, rt, as_stream(stream));
    });
}
int cv_llm_batch_decode(cv_llm* m, int32_t n_steps, int32_t* out_tokens, int32_t* n_out, int32_t* finished, void* stream) {
    return guarded([&] { CV_CHECK(m, "null handle"); batch_decode(m, n_steps, out_tokens, n_out, finished, resolve(m, stream)); });
}
/* logits of one slot of the batched decode as left by the last step's head GEMM (test hook) */
int cv_llm_batch_logits(cv_llm* m, int32_t slot, float* host_out, void* stream) {
    return guarded([&] {
        CV_CHECK(m && host_out && slot >= 0 && slot < m->bt.nb, "cv_llm_batch_logits: bad arguments");
        hipStream_t s = resolve(m, stream);
        CV_HIP(hipMemcpyAsync(host_out, m->bt.logits.as<float>() + (size_t)slot * m->V, (size_t)m->V * 4, hipMemcpyDeviceToHost, s));
        CV_HIP(hipStreamSynchronize(s));
    });
}
int cv_llm_last_logits(cv_llm* m, float* host_out, void* stream) {
    return guarded([&] { CV_CHECK(m && host_out, "null argument");
                         CV_HIP(hipMemcpyAsync(host_out, m->logits.p, (size_t)m->V * 4, hipMemcpyDeviceToHost, as_stream(stream)));
                         CV_HIP(hipStreamSynchronize(as_stream(stream))); });
}
int cv_llm_last_hidden(cv_llm* m, float* host_out, void* stream) {
    return guarded([&] { CV_CHECK(m && host_out, "null argument");
                         CV_HIP(hipMemcpyAsync(host_out, m->h.p, (size_t)m->cfg.hidden * 4, hipMemcpyDeviceToHost, as_stream(stream)));
                         CV_HIP(hipStreamSynchronize(as_stream(stream))); });
}

int cv_gather_rows(const void* table, int32_t dtype, int64_t table_rows, int32_t dim, const int32_t* ids_dev, int32_t n, float* out, float scale, void* stream) {
    return guarded([&] {
        CV_CHECK(table && ids_dev && out && n > 0, "cv_gather_rows: bad arguments");
        hipLaunchKernelGGL(gather_rows_kernel, dim3(n), dim3(256), 0, as_stream(stream), table, dtype == CV_BF16 ? 1 : 0, ids_dev, n, dim, out, (long long)table_rows, scale);
    });
}

}  // extern "C"
