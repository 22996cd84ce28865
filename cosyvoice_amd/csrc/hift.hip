// Stage driver: HiFT vocoder (boundary B6 of SURVEY.md §8b) — replaces HiFTGenerator.inference / decode and
// ConvRNNF0Predictor.forward (cosyvoice/hifigan/generator.py:507-569, f0_predictor.py:56-59).  fp32 throughout (the
// reference runs HiFT outside autocast, cli/model.py:312).  Activations are channel-last; every Conv1d / ConvTranspose1d
// is the exact-fp32 MFMA implicit GEMM with Snake / leaky-ReLU fused as prologue and bias / residual / (1/3) average
// fused as epilogue, so a ResBlock iteration is two launches instead of the reference's six.
#include <vector>
#include <cstdlib>
#include "ops.h"
#include "tensor_map.h"
#include "hift_kernels.h"

using namespace cv;

namespace {
struct Conv { const float* w = nullptr; const float* b = nullptr; int N = 0, K = 0, Kp = 0, taps = 1; const void* w3 = nullptr; };   // w3: the three bf16 planes of w ([3 N][taps Kp], weights.py::split3_planes)
struct ResBlockW { Conv c1[4], c2[4]; const float* a1[4]; const float* a2[4]; int k = 3; };
inline unsigned nblk(long long n) { return (unsigned)((n + 255) / 256); }
// utterances of EQUAL length that share one launch sequence (cv_hift_inference_batch): every activation buffer then holds nb dense blocks one after the other and
// every convolution runs with batch = nb (zero padding per block: the range check of gemm_conv is per batch base).  1 everywhere else - the single-utterance
// launches are exactly what they were.
thread_local int tl_nb = 1;
thread_local int tl_terms = 0;      // GemmConvArgs::w3_terms of the handle being run (option "terms")
}  // namespace

struct cv_hift {
    cv_hift_config cfg{};
    TensorMap tm;
    bool finalized = false;
    Conv f0c[5], conv_pre, conv_post, ups[4], sdown[4];
    const float* f0_cls_w = nullptr; const float* f0_cls_b = nullptr; const float* src_w = nullptr; const float* src_b = nullptr;
    ResBlockW src_rb[4]; std::vector<ResBlockW> rb;
    int scale = 480, sd_rate[4] = {1, 1, 1, 1};
    DevBuf mel_cl, fa, fb, f0, P, s, sst, x, xs, t1, r0, r1, si, y_spec, xu, sn;
    int terms = 6;                // option "terms": plane products per k of the two-sided split in the DECODER's convolutions (6 = fp32-exact class; 3: gemm_conv.h w3_terms).
                                  // The f0 predictor always runs exact: its output is integrated into a phase over the whole utterance.
    bool f0_f64 = false;          // option "f0_float64": the f0 predictor in double (the reference's mode for the causal generator, generator.py:716-717)
    DevBuf fa64, fb64;
    int cap_m = 0;
};

static Conv get_conv(const cv_hift* m, const std::string& name, int N, int K, int taps) {
    Conv c; c.N = N; c.K = K; c.Kp = round_up32(K); c.taps = taps;
    c.w = m->tm.f32(name + ".w", (long long)N * taps * c.Kp);
    c.b = m->tm.f32(name + ".b", N);
    if (m->tm.has(name + ".w3")) c.w3 = m->tm.get(name + ".w3", CV_BF16, 3LL * N * taps * c.Kp).p;
    return c;
}

static void hift_finalize(cv_hift* m) {
    const auto& c = m->cfg;
    CV_CHECK(c.n_ups >= 1 && c.n_ups <= 4 && c.n_res >= 1 && c.n_res <= 4 && c.n_dil >= 1 && c.n_dil <= 4, "hift: bad config counts");
    CV_CHECK(c.n_fft == 16 && c.hop == 4, "hift: the STFT/iSTFT kernels are specialised for n_fft 16 / hop 4 (cosyvoice2.yaml:100-102)");
    CV_CHECK(c.harmonics + 1 <= 16, "hift: at most 15 harmonics (hift_phase_kernel keeps 16 phase increments per frame in LDS)");
    m->scale = c.hop;
    for (int i = 0; i < c.n_ups; ++i) m->scale *= c.ups[i];
    int cin = c.mel;
    CV_CHECK(!c.causal || (c.look_right >= 1 && c.look_right <= 16), "hift: conv_pre_look_right out of range");
    for (int j = 0; j < 5; ++j) { m->f0c[j] = get_conv(m, "f0.conv" + std::to_string(j), c.f0_ch, cin, (c.causal && j == 0) ? 4 : 3); cin = c.f0_ch; }
    CV_CHECK(c.f0_ch % 32 == 0, "hift: f0_ch must be a multiple of 32");
    m->f0_cls_w = m->tm.f32("f0.cls.w", c.f0_ch); m->f0_cls_b = m->tm.f32("f0.cls.b", 1);
    m->src_w = m->tm.f32("source.w", c.harmonics + 1); m->src_b = m->tm.f32("source.b", 1);
    m->conv_pre = get_conv(m, "conv_pre", c.base, c.mel, c.causal ? c.look_right + 1 : 7);
    int ch = c.base;
    for (int i = 0; i < c.n_ups; ++i) {
        const int u = c.ups[i], k = c.up_k[i], q = (k + u - 1) / u;
        if (c.causal) m->ups[i] = get_conv(m, "ups." + std::to_string(i), ch / 2, ch, k);       // CausalConv1dUpsample: stride-1 conv behind a nearest upsampling
        else {
            CV_CHECK((k - u) % 2 == 0, "hift: upsample kernel - rate must be even");
            m->ups[i] = get_conv(m, "ups." + std::to_string(i), u * (ch / 2), ch, q);
        }
        ch /= 2;
    }
    // source_downs strides: cumprod([1] + ups[::-1][:-1])[::-1]   (generator.py:443-455)
    { int r = 1; for (int i = c.n_ups - 1; i >= 0; --i) { m->sd_rate[i] = r; r *= c.ups[i]; } }
    auto resblock = [&](const std::string& p, int C, int k) {
        ResBlockW w; w.k = k;
        for (int j = 0; j < c.n_dil; ++j) {
            w.c1[j] = get_conv(m, p + "convs1." + std::to_string(j), C, C, k);
            w.c2[j] = get_conv(m, p + "convs2." + std::to_string(j), C, C, k);
            w.a1[j] = m->tm.f32(p + "convs1." + std::to_string(j) + ".alpha", round_up32(C));
            w.a2[j] = m->tm.f32(p + "convs2." + std::to_string(j) + ".alpha", round_up32(C));
        }
        return w;
    };
    for (int i = 0; i < c.n_ups; ++i) {
        const int C = c.base >> (i + 1), r = m->sd_rate[i], k = r == 1 ? 1 : 2 * r;
        m->sdown[i] = get_conv(m, "source_downs." + std::to_string(i), C, k * (c.n_fft + 2), 1);
        m->src_rb[i] = resblock("source_resblocks." + std::to_string(i) + ".", C, c.src_k[i]);
        for (int j = 0; j < c.n_res; ++j) m->rb.push_back(resblock("resblocks." + std::to_string(i * c.n_res + j) + ".", C, c.res_k[j]));
    }
    m->conv_post = get_conv(m, "conv_post", c.n_fft + 2, ch, 7);
    m->finalized = true;
}

static void reserve(cv_hift* m, int frames) {
    if (frames <= m->cap_m) return;
    const auto& c = m->cfg; const size_t mm = frames, L = mm * m->scale, F = L / 4 + 1;
    size_t big = 0, T = mm, ch = c.base;
    for (int i = 0; i < c.n_ups; ++i) { T *= c.ups[i]; ch /= 2; big = std::max(big, (T + 2) * ch); }
    big = std::max(big, mm * (size_t)std::max(c.base, c.f0_ch));
    m->mel_cl.ensure(mm * c.mel * 4); m->fa.ensure(mm * c.f0_ch * 4); m->fb.ensure(mm * c.f0_ch * 4); m->f0.ensure((mm + 4) * 4);
    m->P.ensure(mm * (c.harmonics + 1) * 4); m->s.ensure(L * 4); m->sst.ensure(F * 18 * 4); m->y_spec.ensure(F * 18 * 4);
    for (DevBuf* b : {&m->x, &m->xs, &m->t1, &m->r0, &m->r1, &m->si, &m->sn}) b->ensure(big * 4);
    if (c.causal) m->xu.ensure(2 * big * 4);               // nearest-upsampled input of a CausalConv1dUpsample (twice the channels of its output)
    m->cap_m = frames;
}

// Conv1d on channel-last rows: tap j reads row (t + j*dil - pad)
static void conv(const Conv& w, const float* A, long long a_rows, long long M, int pad, int dil, float* C, hipStream_t s, int pro, float pro_p,
                 const float* alpha, int act, const float* res, float out_scale, bool accumulate, const float* act_alpha = nullptr, float* C2 = nullptr,
                 const float* c2_alpha = nullptr) {
    GemmConvArgs a{};
    a.A = A; a.a_batch = 0; a.a_len = a_rows * w.K; a.lda = w.K; a.a_off0 = -pad * w.K; a.tap_step = dil * w.K; a.taps = w.taps; a.K = w.K;
    a.pro = pro; a.pro_p = pro_p; a.pro_alpha = alpha;
    a.W = w.w; a.W3 = w.w3; a.w3_terms = tl_terms; a.Kp = w.Kp; a.ldw = 0; a.w_batch = 0; a.bias = w.b;
    a.C = C; a.c_batch = 0; a.c_len = M * w.N; a.ldc = w.N; a.c_off = 0; a.M = (int)M; a.N = w.N;
    a.act = act; a.act_p = 0.f; a.res = res; a.res_batch = 0; a.out_scale = out_scale; a.row_scale = nullptr; a.accumulate = accumulate ? 1 : 0;
    a.act_alpha = act_alpha; a.C2 = C2; a.c2_alpha = c2_alpha;
    if (tl_nb > 1) { a.a_batch = a_rows * w.K; a.c_batch = M * w.N; a.res_batch = res ? M * w.N : 0; }
    gemm_conv(a, false, tl_nb, s);
}

// ResBlock (generator.py:46-122): for each dilation: xt = conv2(snake(conv1(snake(x)))) ; x = xt + x.
// `in` is left intact; the last iteration writes dest = (x_final) * out_scale (+ dest if accumulate).
static void resblock(cv_hift* m, const ResBlockW& w, const float* in, long long T, float* dest, float out_scale, bool accumulate, hipStream_t s) {
    const auto& c = m->cfg;
    float* t1 = m->t1.as<float>(); float* r[2] = {m->r0.as<float>(), m->r1.as<float>()};
    const float* cur = in;
    // Snake is applied ONCE per value, where the value is produced (round 3): the block input by one elementwise pass, conv1's output in conv1's
    // epilogue, conv2's output (the next iteration's input) as conv2's second output.  As a PROLOGUE of the consuming convolution (rounds 1-2) every
    // element was activated once per tap and per N-tile - a k = 11 convolution over four 64-column tiles evaluated sinf 44 times per element, and the
    // HiFT convolutions were bound by that, not by the matrix pipe (profiles/r3_hift_ab.txt).  Same fp32 values either way: results are bit-identical.
    const bool once = [] { const char* e = getenv("CV_HIFT_SNAKE_ONCE"); return !(e && e[0] == '0'); }();              // A/B knob, read at every call
    if (!once) {
        for (int j = 0; j < c.n_dil; ++j) {
            const int d = c.dil[j], k = w.k;
            conv(w.c1[j], cur, T, T, c.causal ? (k - 1) * d : (k * d - d) / 2, d, t1, s, ACT_SNAKE, 0.f, w.a1[j], ACT_NONE, nullptr, 1.f, false);
            const bool last = j == c.n_dil - 1;
            float* out = last ? dest : r[j & 1];
            conv(w.c2[j], t1, T, T, c.causal ? k - 1 : (k - 1) / 2, 1, out, s, ACT_SNAKE, 0.f, w.a2[j], ACT_NONE, cur, last ? out_scale : 1.f, last && accumulate);
            cur = out;
        }
        return;
    }
    const int C = w.c1[0].K;
    CV_CHECK(C % 4 == 0, "hift: ResBlock channels must be a multiple of 4");
    float* sn = m->sn.as<float>();
    hipLaunchKernelGGL(snake_rows_kernel, dim3(nblk(tl_nb * T * C / 4)), dim3(256), 0, s, in, sn, w.a1[0], tl_nb * T * C / 4, C);
    for (int j = 0; j < c.n_dil; ++j) {
        const int d = c.dil[j], k = w.k;
        // padding: "same" for HiFTGenerator, all on the left for the causal generator (CausalConv1d 'left': (k - 1) * dilation, convolution.py:172)
        conv(w.c1[j], sn, T, T, c.causal ? (k - 1) * d : (k * d - d) / 2, d, t1, s, ACT_NONE, 0.f, nullptr, ACT_SNAKE, nullptr, 1.f, false, w.a2[j]);
        const bool last = j == c.n_dil - 1;
        float* out = last ? dest : r[j & 1];
        conv(w.c2[j], t1, T, T, c.causal ? k - 1 : (k - 1) / 2, 1, out, s, ACT_NONE, 0.f, nullptr, ACT_NONE, cur, last ? out_scale : 1.f, last && accumulate,
             nullptr, last ? nullptr : sn, last ? nullptr : w.a1[j + 1]);
        cur = out;
    }
}

// Non-causal: ConvRNNF0Predictor.  Causal (CausalConvRNNF0Predictor, f0_predictor.py:62-103): the first conv (k = 4) looks 3 frames to the RIGHT
// - zeros after the last frame when `finalize`, otherwise the last 3 frames are only context and frames - 3 values come out - the other four
// are causal to the left.  Returns the number of f0 values.  (The reference runs this predictor in float64, generator.py:716-717; here it is fp32
// on the exact-fp32 MFMA: every output row sums its taps in the same order whatever the chunk, so chunked and one-shot calls agree bit for bit.)
static void conv64(const Conv& w, const void* in, bool in_f32, int in_rows, int M, int pad, void* out, bool out_f32, int act, hipStream_t s) {
    hipLaunchKernelGGL(conv_f64_kernel, dim3((M + 15) / 16, (w.N + 63) / 64), dim3(256), 0, s, in, in_f32 ? 1 : 0, in_rows, w.K, w.w, w.Kp, w.b, out, out_f32 ? 1 : 0, M, w.N, w.taps,
                       pad, act);
}

static int hift_f0(cv_hift* m, const float* mel_cl, int frames, hipStream_t s, bool finalize = true) {
    float* a = m->fa.as<float>(); float* b = m->fb.as<float>();
    const float* cur = mel_cl;
    const bool causal = m->cfg.causal != 0;
    const int out_frames = (causal && !finalize) ? frames - 3 : frames;
    CV_CHECK(out_frames > 0, "hift: too few frames for a non-final causal chunk");
    if (m->f0_f64) {                                         // the same five convolutions + classifier, every sum in double, f0 rounded to fp32 at the end
        m->fa64.ensure((size_t)frames * m->cfg.f0_ch * 8); m->fb64.ensure((size_t)frames * m->cfg.f0_ch * 8);
        const void* c64 = mel_cl; bool f32 = true;
        for (int j = 0; j < 5; ++j) {
            void* out = (j & 1) ? m->fb64.p : m->fa64.p;
            if (causal && j == 0) conv64(m->f0c[j], c64, f32, frames, out_frames, 0, out, false, 1, s);
            else conv64(m->f0c[j], c64, f32, out_frames, out_frames, causal ? 2 : 1, out, false, 1, s);
            c64 = out; f32 = false;
        }
        Conv cls; cls.w = m->f0_cls_w; cls.b = m->f0_cls_b; cls.N = 1; cls.K = m->cfg.f0_ch; cls.Kp = m->cfg.f0_ch; cls.taps = 1;
        conv64(cls, c64, false, out_frames, out_frames, 0, m->f0.p, true, 2, s);
        return out_frames;
    }
    for (int j = 0; j < 5; ++j) {
        float* out = (j & 1) ? b : a;
        if (causal && j == 0) conv(m->f0c[j], cur, frames, out_frames, 0, 1, out, s, ACT_NONE, 0.f, nullptr, ACT_ELU, nullptr, 1.f, false);
        else conv(m->f0c[j], cur, out_frames, out_frames, causal ? 2 : 1, 1, out, s, ACT_NONE, 0.f, nullptr, ACT_ELU, nullptr, 1.f, false);
        cur = out;
    }
    frames = out_frames;
    Conv cls; cls.w = m->f0_cls_w; cls.b = m->f0_cls_b; cls.N = 1; cls.K = m->cfg.f0_ch; cls.Kp = m->cfg.f0_ch; cls.taps = 1;
    conv(cls, cur, frames, frames, 0, 1, m->f0.as<float>(), s, ACT_NONE, 0.f, nullptr, ACT_ABS, nullptr, 1.f, false);
    return frames;
}

static void hift_source(cv_hift* m, int frames, const float* noise, unsigned long long seed, hipStream_t s, const unsigned long long* seeds = nullptr) {
    const auto& c = m->cfg; const int H = c.harmonics + 1;
    const long long L = (long long)frames * m->scale;
    for (int b = 0; b < tl_nb; ++b) {                        // the phase walk and the RNG key are per utterance (seeds: one key per block of a batch)
        const float* f0 = m->f0.as<float>() + (size_t)b * frames; float* P = m->P.as<float>() + (size_t)b * frames * H;
        hipLaunchKernelGGL(hift_phase_kernel, dim3(1), dim3(256), 0, s, f0, P, frames, H, (float)c.sr, (float)m->scale);
        hipLaunchKernelGGL(hift_source_kernel, dim3(nblk(L)), dim3(256), 0, s, f0, P, noise ? noise + (size_t)b * L * H : nullptr, seeds ? seeds[b] : seed, m->src_w, m->src_b,
                           m->s.as<float>() + (size_t)b * L, frames, H, m->scale, c.nsf_alpha, c.nsf_sigma, c.voiced_thr, c.causal ? 1 : 0);
    }
}

// decode(x = mel, s = source) -> waveform   (generator.py:507-539)
static void hift_decode(cv_hift* m, const float* mel_cl, int frames, const float* src, float* speech, hipStream_t s) {
    struct TermScope { TermScope(int t) { tl_terms = t; } ~TermScope() { tl_terms = 0; } } term_scope(m->terms == 3 ? 3 : 0);
    const auto& c = m->cfg;
    const long long L = (long long)frames * m->scale, F = L / 4 + 1;
    float* sst = m->sst.as<float>();
    for (int b = 0; b < tl_nb; ++b) hipLaunchKernelGGL(hift_stft_kernel, dim3(nblk(F)), dim3(256), 0, s, src + (size_t)b * L, sst + (size_t)b * F * 18, L, F);
    float* x = m->x.as<float>(); float* xs = m->xs.as<float>(); float* si = m->si.as<float>();
    conv(m->conv_pre, mel_cl, frames, frames, 3, 1, x, s, ACT_NONE, 0.f, nullptr, ACT_NONE, nullptr, 1.f, false);
    long long T = frames; int ch = c.base;
    for (int i = 0; i < c.n_ups; ++i) {
        const int u = c.ups[i], k = c.up_k[i], p = (k - u) / 2, cout = ch / 2;
        const bool last = i == c.n_ups - 1;
        const long long Tout = (T - 1) * u - 2 * p + k, rows = Tout + (last ? 1 : 0);
        {   // leaky_relu(0.1) -> ConvTranspose1d, polyphase: row u_idx of the GEMM is output rows u_idx*u - p .. + u - 1
            const Conv& w = m->ups[i];
            GemmConvArgs a{};
            a.A = x; a.a_batch = 0; a.a_len = T * ch; a.lda = ch; a.a_off0 = 0; a.tap_step = -ch; a.taps = w.taps; a.K = ch;
            a.pro = ACT_LEAKY; a.pro_p = c.lrelu; a.pro_alpha = nullptr;
            a.W = w.w; a.Kp = w.Kp; a.ldw = 0; a.w_batch = 0; a.bias = w.b;
            a.C = xs; a.c_batch = 0; a.c_len = rows * cout; a.ldc = u * cout; a.c_off = (long long)(-p + (last ? 1 : 0)) * cout;
            a.M = (int)(T + w.taps - 1); a.N = u * cout; a.act = ACT_NONE; a.res = nullptr; a.out_scale = 1.f; a.row_scale = nullptr; a.accumulate = 0;
            if (tl_nb > 1) { a.a_batch = T * ch; a.c_batch = rows * cout; }
            gemm_conv(a, false, tl_nb, s);
            if (last) for (int b = 0; b < tl_nb; ++b) hipLaunchKernelGGL(reflect_row0_kernel, dim3(1), dim3(256), 0, s, xs + (size_t)b * rows * cout, cout);       // ReflectionPad1d((1, 0))
        }
        std::swap(x, xs);
        T = rows; ch = cout;
        {   // fusion with the source branch: si = source_resblock(source_down(s_stft)); x = x + si   (generator.py:518-521)
            const int r = m->sd_rate[i], kk = r == 1 ? 1 : 2 * r, pad = r == 1 ? 0 : r / 2;
            const long long Tsi = (F + 2 * pad - kk) / r + 1;
            CV_CHECK(Tsi == T, "hift: source branch length does not match the upsampled mel length");
            const Conv& w = m->sdown[i];
            GemmConvArgs a{};
            a.A = sst; a.a_batch = 0; a.a_len = F * 18; a.lda = r * 18; a.a_off0 = -pad * 18; a.tap_step = 0; a.taps = 1; a.K = kk * 18;
            a.pro = ACT_NONE; a.W = w.w; a.Kp = w.Kp; a.ldw = 0; a.w_batch = 0; a.bias = w.b;
            a.C = si; a.c_batch = 0; a.c_len = T * ch; a.ldc = ch; a.c_off = 0; a.M = (int)T; a.N = ch;
            a.act = ACT_NONE; a.res = nullptr; a.out_scale = 1.f; a.row_scale = nullptr; a.accumulate = 0;
            if (tl_nb > 1) { a.a_batch = F * 18; a.c_batch = T * ch; }
            gemm_conv(a, false, tl_nb, s);
            resblock(m, m->src_rb[i], si, T, x, 1.f, true, s);
        }
        for (int j = 0; j < c.n_res; ++j)      // xs = sum_j resblock_j(x) / n_res   (generator.py:523-529)
            resblock(m, m->rb[i * c.n_res + j], x, T, xs, 1.f / (float)c.n_res, j > 0, s);
        std::swap(x, xs);
    }
    CV_CHECK(T == F, "hift: frame bookkeeping mismatch");
    float* spec = m->y_spec.as<float>();
    conv(m->conv_post, x, T, T, 3, 1, spec, s, ACT_LEAKY, 0.01f, nullptr, ACT_NONE, nullptr, 1.f, false);    // F.leaky_relu default slope
    hipLaunchKernelGGL(hift_spec_kernel, dim3(nblk(tl_nb * F * 9)), dim3(256), 0, s, spec, tl_nb * F);
    for (int b = 0; b < tl_nb; ++b) hipLaunchKernelGGL(hift_istft_kernel, dim3(nblk(L)), dim3(256), 0, s, spec + (size_t)b * F * 18, speech + (size_t)b * L, F, L, c.audio_limit);
}


// CausalHiFTGenerator.decode (generator.py:684-726).  x = mel_cl [mx][mel], src [480 mx].  finalize: every frame is final.  Otherwise the last
// `look_right` frames are only the right context of conv_pre, the source STFT loses its last prod(ups) * look_right frames and the last
// prod(ups) * hop samples of the iSTFT are withheld.  Returns the number of samples written to `speech`.
static long long hift_decode_causal(cv_hift* m, const float* mel_cl, int mx, const float* src, bool finalize, float* speech, hipStream_t s) {
    struct TermScope { TermScope(int t) { tl_terms = t; } ~TermScope() { tl_terms = 0; } } term_scope(m->terms == 3 ? 3 : 0);       // option "terms" (CosyVoice3Model fp16 mode), as in hift_decode
    const auto& c = m->cfg;
    int up = 1; for (int i = 0; i < c.n_ups; ++i) up *= c.ups[i];
    const long long Ls = (long long)mx * m->scale, Fall = Ls / 4 + 1, F = finalize ? Fall : Fall - (long long)up * c.look_right;
    const int M0 = finalize ? mx : mx - c.look_right;
    CV_CHECK(M0 > 0 && F > 0 && (finalize || F > up), "hift: too few frames for a non-final causal chunk");
    float* sst = m->sst.as<float>();
    hipLaunchKernelGGL(hift_stft_kernel, dim3(nblk(Fall)), dim3(256), 0, s, src, sst, Ls, Fall);
    float* x = m->x.as<float>(); float* xs = m->xs.as<float>(); float* si = m->si.as<float>(); float* xu = m->xu.as<float>();
    conv(m->conv_pre, mel_cl, mx, M0, 0, 1, x, s, ACT_NONE, 0.f, nullptr, ACT_NONE, nullptr, 1.f, false);      // taps look right: rows >= mx read as zero
    long long T = M0; int ch = c.base;
    for (int i = 0; i < c.n_ups; ++i) {
        const int u = c.ups[i], k = c.up_k[i], cout = ch / 2;
        const bool last = i == c.n_ups - 1;
        const long long Tu = T * u, rows = Tu + (last ? 1 : 0);
        {   // leaky_relu(0.1) -> nearest x u -> left pad k - 1 -> Conv1d(k)   (CausalConv1dUpsample; leaky commutes with the upsampling)
            hipLaunchKernelGGL(repeat_rows_kernel, dim3(nblk(Tu * ch)), dim3(256), 0, s, x, xu, T, ch, u);
            const Conv& w = m->ups[i];
            GemmConvArgs a{};
            a.A = xu; a.a_batch = 0; a.a_len = Tu * ch; a.lda = ch; a.a_off0 = -(k - 1) * ch; a.tap_step = ch; a.taps = w.taps; a.K = ch;
            a.pro = ACT_LEAKY; a.pro_p = c.lrelu; a.pro_alpha = nullptr;
            a.W = w.w; a.Kp = w.Kp; a.ldw = 0; a.w_batch = 0; a.bias = w.b;
            a.C = xs; a.c_batch = 0; a.c_len = rows * cout; a.ldc = cout; a.c_off = last ? cout : 0;
            a.M = (int)Tu; a.N = cout; a.act = ACT_NONE; a.res = nullptr; a.out_scale = 1.f; a.row_scale = nullptr; a.accumulate = 0;
            gemm_conv(a, false, 1, s);
            if (last) hipLaunchKernelGGL(reflect_row0_kernel, dim3(1), dim3(256), 0, s, xs, cout);       // ReflectionPad1d((1, 0))
        }
        std::swap(x, xs);
        T = rows; ch = cout;
        {   // source branch: CausalConv1d(k = 1) or CausalConv1dDownSample(kernel 2 r, stride r, left pad r - 1), then the causal ResBlock
            const int r = m->sd_rate[i], kk = r == 1 ? 1 : 2 * r, pad = r == 1 ? 0 : r - 1;
            const long long Tsi = (F + pad - kk) / r + 1;
            CV_CHECK(Tsi == T, "hift: source branch length does not match the upsampled mel length");
            const Conv& w = m->sdown[i];
            GemmConvArgs a{};
            a.A = sst; a.a_batch = 0; a.a_len = F * 18; a.lda = r * 18; a.a_off0 = -pad * 18; a.tap_step = 0; a.taps = 1; a.K = kk * 18;
            a.pro = ACT_NONE; a.W = w.w; a.Kp = w.Kp; a.ldw = 0; a.w_batch = 0; a.bias = w.b;
            a.C = si; a.c_batch = 0; a.c_len = T * ch; a.ldc = ch; a.c_off = 0; a.M = (int)T; a.N = ch;
            a.act = ACT_NONE; a.res = nullptr; a.out_scale = 1.f; a.row_scale = nullptr; a.accumulate = 0;
            gemm_conv(a, false, 1, s);
            resblock(m, m->src_rb[i], si, T, x, 1.f, true, s);
        }
        for (int j = 0; j < c.n_res; ++j)
            resblock(m, m->rb[i * c.n_res + j], x, T, xs, 1.f / (float)c.n_res, j > 0, s);
        std::swap(x, xs);
    }
    CV_CHECK(T == F, "hift: frame bookkeeping mismatch");
    float* spec = m->y_spec.as<float>();
    conv(m->conv_post, x, T, T, 6, 1, spec, s, ACT_LEAKY, 0.01f, nullptr, ACT_NONE, nullptr, 1.f, false);    // CausalConv1d(k = 7, 'left')
    hipLaunchKernelGGL(hift_spec_kernel, dim3(nblk(F * 9)), dim3(256), 0, s, spec, F);
    const long long L = 4 * (F - 1) - (finalize ? 0 : (long long)up * c.hop);
    hipLaunchKernelGGL(hift_istft_kernel, dim3(nblk(L)), dim3(256), 0, s, spec, speech, F, L, c.audio_limit);
    return L;
}

extern "C" {

int cv_hift_create(cv_hift** out, const cv_hift_config* cfg) {
    return guarded([&] { CV_CHECK(out && cfg, "cv_hift_create: null argument"); auto* m = new cv_hift(); m->cfg = *cfg; *out = m; });
}
int cv_hift_set_tensor(cv_hift* m, const char* name, const void* dev_ptr, int32_t dtype, int64_t numel) {
    return guarded([&] { CV_CHECK(m, "null handle"); m->tm.set(name, dev_ptr, dtype, numel); });
}
int cv_hift_finalize(cv_hift* m) { return guarded([&] { CV_CHECK(m, "null handle"); hift_finalize(m); }); }
int cv_hift_set_option(cv_hift* m, const char* name, int32_t value) {
    return guarded([&] {
        CV_CHECK(m && name, "cv_hift_set_option: null argument");
        if (std::string(name) == "f0_float64") m->f0_f64 = value != 0;
        else if (std::string(name) == "terms") { CV_CHECK(value == 3 || value == 6, "terms must be 6 (fp32-exact class) or 3"); m->terms = value; }
        else throw Error(std::string("cv_hift_set_option: unknown option ") + name);
    });
}
void cv_hift_destroy(cv_hift* m) {
    if (!m) return;
    // may be called from a garbage-collector finaliser on ANY thread while another thread drives a different handle: quiesce the
    // device and hold the runtime lock so stream / graph / buffer destruction never overlaps a capture, a launch burst or a realloc
    std::lock_guard<std::recursive_mutex> lk(runtime_lock());
    (void)hipDeviceSynchronize();
    delete m;
}

int cv_hift_f0(cv_hift* m, const float* speech_feat, int32_t frames, float* f0_out, void* stream) {
    return guarded([&] {
        CV_CHECK(m && m->finalized && speech_feat && f0_out && frames > 0, "cv_hift_f0: bad arguments");
        hipStream_t s = as_stream(stream);
        reserve(m, frames);
        hipLaunchKernelGGL(to_channel_last_kernel, dim3(nblk((long long)frames * m->cfg.mel)), dim3(256), 0, s, speech_feat, m->mel_cl.as<float>(), m->cfg.mel, frames);
        hift_f0(m, m->mel_cl.as<float>(), frames, s);
        CV_HIP(hipMemcpyAsync(f0_out, m->f0.p, (size_t)frames * 4, hipMemcpyDeviceToDevice, s));
    });
}

int cv_hift_decode(cv_hift* m, const float* speech_feat, int32_t frames, const float* source, float* speech_out, void* stream) {
    return guarded([&] {
        CV_CHECK(m && m->finalized && speech_feat && source && speech_out && frames > 0, "cv_hift_decode: bad arguments");
        hipStream_t s = as_stream(stream);
        reserve(m, frames);
        hipLaunchKernelGGL(to_channel_last_kernel, dim3(nblk((long long)frames * m->cfg.mel)), dim3(256), 0, s, speech_feat, m->mel_cl.as<float>(), m->cfg.mel, frames);
        hift_decode(m, m->mel_cl.as<float>(), frames, source, speech_out, s);
    });
}

int cv_hift_inference(cv_hift* m, const float* speech_feat, int32_t frames, const float* cache_source, int32_t cache_len,
                      const float* noise, uint64_t seed, float* speech_out, float* source_out, void* stream) {
    return guarded([&] {
        CV_CHECK(m && m->finalized && speech_feat && speech_out && source_out && frames > 0, "cv_hift_inference: bad arguments");
        hipStream_t s = as_stream(stream);
        reserve(m, frames);
        const long long L = (long long)frames * m->scale;
        CV_CHECK(cache_len >= 0 && cache_len <= L, "cv_hift_inference: cache_source longer than the utterance");
        hipLaunchKernelGGL(to_channel_last_kernel, dim3(nblk((long long)frames * m->cfg.mel)), dim3(256), 0, s, speech_feat, m->mel_cl.as<float>(), m->cfg.mel, frames);
        hift_f0(m, m->mel_cl.as<float>(), frames, s);
        hift_source(m, frames, noise, seed, s);
        if (cache_len > 0) CV_HIP(hipMemcpyAsync(m->s.p, cache_source, (size_t)cache_len * 4, hipMemcpyDeviceToDevice, s));   // generator.py:566-567
        CV_HIP(hipMemcpyAsync(source_out, m->s.p, (size_t)L * 4, hipMemcpyDeviceToDevice, s));
        hift_decode(m, m->mel_cl.as<float>(), frames, m->s.as<float>(), speech_out, s);
    });
}

int cv_hift_inference_batch(cv_hift* m, int32_t n_utt, const float* speech_feat, int32_t frames, const float* noise, const uint64_t* seeds, float* speech_out,
                            float* source_out, void* stream) {
    return guarded([&] {
        CV_CHECK(m && m->finalized && !m->cfg.causal && !m->f0_f64 && speech_feat && seeds && speech_out && source_out && frames > 0 && n_utt >= 1 && n_utt <= 16,
                 "cv_hift_inference_batch: bad arguments (HiFTGenerator handle, fp32 f0 mode, 1..16 utterances)");
        hipStream_t s = as_stream(stream);
        reserve(m, (frames + 1) * n_utt);                    // every buffer is linear in the frame count (+ the reflection row of the last up-sampling stage per utterance)
        const long long L = (long long)frames * m->scale;
        for (int b = 0; b < n_utt; ++b)
            hipLaunchKernelGGL(to_channel_last_kernel, dim3(nblk((long long)frames * m->cfg.mel)), dim3(256), 0, s, speech_feat + (size_t)b * m->cfg.mel * frames,
                               m->mel_cl.as<float>() + (size_t)b * frames * m->cfg.mel, m->cfg.mel, frames);
        std::vector<unsigned long long> keys(seeds, seeds + n_utt);
        struct Scope { Scope(int n) { tl_nb = n; } ~Scope() { tl_nb = 1; } } scope(n_utt);
        hift_f0(m, m->mel_cl.as<float>(), frames, s);
        hift_source(m, frames, noise, 0, s, keys.data());
        CV_HIP(hipMemcpyAsync(source_out, m->s.p, (size_t)n_utt * L * 4, hipMemcpyDeviceToDevice, s));
        hift_decode(m, m->mel_cl.as<float>(), frames, m->s.as<float>(), speech_out, s);
    });
}

/* CausalHiFTGenerator (Fun-CosyVoice3) entry points; the handle must have been created with cfg.causal = 1. */
int cv_hift_causal_f0(cv_hift* m, const float* speech_feat, int32_t frames, int32_t finalize, float* f0_out, int32_t* n_out, void* stream) {
    return guarded([&] {
        CV_CHECK(m && m->finalized && m->cfg.causal && speech_feat && f0_out && n_out && frames > 0, "cv_hift_causal_f0: bad arguments");
        hipStream_t s = as_stream(stream);
        reserve(m, frames);
        hipLaunchKernelGGL(to_channel_last_kernel, dim3(nblk((long long)frames * m->cfg.mel)), dim3(256), 0, s, speech_feat, m->mel_cl.as<float>(), m->cfg.mel, frames);
        *n_out = hift_f0(m, m->mel_cl.as<float>(), frames, s, finalize != 0);
        CV_HIP(hipMemcpyAsync(f0_out, m->f0.p, (size_t)*n_out * 4, hipMemcpyDeviceToDevice, s));
    });
}
int cv_hift_causal_decode(cv_hift* m, const float* speech_feat, int32_t frames, const float* source, int32_t finalize, float* speech_out, int64_t* n_out,
                          void* stream) {
    return guarded([&] {
        CV_CHECK(m && m->finalized && m->cfg.causal && speech_feat && source && speech_out && n_out && frames > 0, "cv_hift_causal_decode: bad arguments");
        hipStream_t s = as_stream(stream);
        reserve(m, frames);
        hipLaunchKernelGGL(to_channel_last_kernel, dim3(nblk((long long)frames * m->cfg.mel)), dim3(256), 0, s, speech_feat, m->mel_cl.as<float>(), m->cfg.mel, frames);
        *n_out = hift_decode_causal(m, m->mel_cl.as<float>(), frames, source, finalize != 0, speech_out, s);
    });
}
int cv_hift_causal_inference(cv_hift* m, const float* speech_feat, int32_t frames, int32_t finalize, const float* noise, uint64_t seed,
                             float* speech_out, int64_t* n_speech, float* source_out, int64_t* n_source, void* stream) {
    return guarded([&] {
        CV_CHECK(m && m->finalized && m->cfg.causal && speech_feat && speech_out && source_out && n_speech && n_source && frames > 0, "cv_hift_causal_inference: bad arguments");
        hipStream_t s = as_stream(stream);
        reserve(m, frames);
        hipLaunchKernelGGL(to_channel_last_kernel, dim3(nblk((long long)frames * m->cfg.mel)), dim3(256), 0, s, speech_feat, m->mel_cl.as<float>(), m->cfg.mel, frames);
        const int nf0 = hift_f0(m, m->mel_cl.as<float>(), frames, s, finalize != 0);
        hift_source(m, nf0, noise, seed, s);
        *n_source = (long long)nf0 * m->scale;
        CV_HIP(hipMemcpyAsync(source_out, m->s.p, (size_t)*n_source * 4, hipMemcpyDeviceToDevice, s));
        *n_speech = hift_decode_causal(m, m->mel_cl.as<float>(), nf0, m->s.as<float>(), finalize != 0, speech_out, s);     // x = speech_feat[:, :, :nf0]
    });
}

}  // extern "C"
