"""Deterministic random-init weights with the REFERENCE's state-dict key names and shapes, and the synthetic utterance of
SURVEY.md §8d — the "synthetic data" of bench.py / smoke() and of every parity test (no oracle code in here).

There are no checkpoints on the build / GPU boxes, so parity and the benchmark use seeded random weights of the real
architecture (numpy PCG64 streams: reproducible across machines and torch versions).  Key names/shapes are validated by
loading these dicts into the real reference modules with strict=True (tests/golden/make_golden.py).
Configs: `cv2()` = CosyVoice2-0.5B, `tiny()` = same topology, small dims (emulator-sized tests).
"""
import numpy as np
import torch


from .configs import CV1Config, FlowConfig, HiftConfig, LLMConfig, cv1, cv2, cv3_flow, cv3_llm, tiny, tiny_cv1, tiny_cv1_k, tiny_cv3_flow, tiny_cv3_llm  # noqa: F401


class _Gen:
    def __init__(self, seed):
        self.rng = np.random.default_rng(seed)

    def normal(self, shape, std=1.0):
        return torch.from_numpy((self.rng.standard_normal(shape, dtype=np.float32) * std).astype(np.float32))

    def linear(self, out_f, in_f, gain=1.0):
        return self.normal((out_f, in_f), gain / np.sqrt(in_f))

    def conv(self, out_c, in_c, k, gain=1.0):
        return self.normal((out_c, in_c, k), gain / np.sqrt(in_c * k))

    def gamma(self, n):
        return 1.0 + self.normal((n,), 0.1)

    def beta(self, n):
        return self.normal((n,), 0.05)


def to_bf16_grid(sd):
    """Round every floating tensor to the nearest bf16 value (kept as fp32): the shared 'W16' weight policy."""
    return {k: v.to(torch.bfloat16).to(torch.float32) if v.is_floating_point() else v for k, v in sd.items()}


def make_llm(cfg: LLMConfig, seed=1986):
    """Keys of cosyvoice.llm.llm.Qwen2LM (llm/llm.py:257-297), or CosyVoice3LM (:664-706) when cfg.cv3, over transformers.Qwen2ForCausalLM."""
    g, sd, H = _Gen(seed), {}, cfg.hidden
    if not cfg.cv3:
        sd["llm_embedding.weight"] = g.normal((2, H), 0.5)
    sd["llm.model.model.embed_tokens.weight"] = g.normal((cfg.text_vocab, H), 0.5)
    for i in range(cfg.layers):
        p = "llm.model.model.layers.%d." % i
        sd[p + "self_attn.q_proj.weight"] = g.linear(cfg.heads * cfg.head_dim, H)
        sd[p + "self_attn.q_proj.bias"] = g.normal((cfg.heads * cfg.head_dim,), 0.1)
        sd[p + "self_attn.k_proj.weight"] = g.linear(cfg.kv_heads * cfg.head_dim, H)
        sd[p + "self_attn.k_proj.bias"] = g.normal((cfg.kv_heads * cfg.head_dim,), 0.1)
        sd[p + "self_attn.v_proj.weight"] = g.linear(cfg.kv_heads * cfg.head_dim, H)
        sd[p + "self_attn.v_proj.bias"] = g.normal((cfg.kv_heads * cfg.head_dim,), 0.1)
        sd[p + "self_attn.o_proj.weight"] = g.linear(H, cfg.heads * cfg.head_dim)
        sd[p + "mlp.gate_proj.weight"] = g.linear(cfg.inter, H)
        sd[p + "mlp.up_proj.weight"] = g.linear(cfg.inter, H)
        sd[p + "mlp.down_proj.weight"] = g.linear(H, cfg.inter)
        sd[p + "input_layernorm.weight"] = g.gamma(H)
        sd[p + "post_attention_layernorm.weight"] = g.gamma(H)
    sd["llm.model.model.norm.weight"] = g.gamma(H)
    sd["llm.model.lm_head.weight"] = sd["llm.model.model.embed_tokens.weight"]       # tied (unused by the hot path)
    V = cfg.speech_token_size + cfg.n_special
    sd["llm_decoder.weight"] = g.linear(V, H, gain=3.0)
    if not cfg.cv3:                                    # CosyVoice3LM: bias=False (llm/llm.py:688)
        sd["llm_decoder.bias"] = g.normal((V,), 0.1)
    else:
        # 200 of the ids are special: with uniformly random rows almost every argmax would be a stop id and every fixture would end
        # at step 0.  A trained model emits them rarely; the seeded rows of ids > eos are damped so that sequences have a body.
        sd["llm_decoder.weight"][cfg.speech_token_size + 2:] *= 0.3
    sd["speech_embedding.weight"] = g.normal((V, H), 0.5)
    return to_bf16_grid(sd)


def bistream_fixture(sd, cfg: LLMConfig, eos_bias=0.0):
    """Head of a seeded Qwen2LM state dict adjusted so that PLAIN greedy decoding can walk inference_bistream (llm/llm.py:551-661) on
    random weights: the special id speech_token_size + 1 and the fill token never win an argmax (fills are then only the forced ones; a
    sampled special id raises in the reference), and eos gets `eos_bias` so that the final phase ends.  Returns a new dict."""
    assert not cfg.cv3
    out = dict(sd)
    b = sd["llm_decoder.bias"].clone()
    b[cfg.speech_token_size + 1] = -60.0
    b[cfg.speech_token_size + 2] = -60.0
    b[cfg.speech_token_size] += eos_bias
    out["llm_decoder.bias"] = b.bfloat16().float()
    return out


def _conformer_layer(g, sd, p, d, heads, ffn):
    sd[p + "self_attn.pos_bias_u"] = g.normal((heads, d // heads), 0.1)
    sd[p + "self_attn.pos_bias_v"] = g.normal((heads, d // heads), 0.1)
    for n in ("q", "k", "v", "out"):
        sd[p + "self_attn.linear_%s.weight" % n] = g.linear(d, d)
        sd[p + "self_attn.linear_%s.bias" % n] = g.normal((d,), 0.05)
    sd[p + "self_attn.linear_pos.weight"] = g.linear(d, d)
    sd[p + "feed_forward.w_1.weight"] = g.linear(ffn, d)
    sd[p + "feed_forward.w_1.bias"] = g.normal((ffn,), 0.05)
    sd[p + "feed_forward.w_2.weight"] = g.linear(d, ffn)
    sd[p + "feed_forward.w_2.bias"] = g.normal((d,), 0.05)
    for n in ("norm_ff", "norm_mha"):
        sd[p + n + ".weight"] = g.gamma(d)
        sd[p + n + ".bias"] = g.beta(d)


def make_flow(cfg: FlowConfig, seed=1987):
    """Keys of cosyvoice.flow.flow.CausalMaskedDiffWithXvec (flow/flow.py:150-186) incl. encoder and decoder.estimator."""
    g, sd, d, C = _Gen(seed), {}, cfg.dim, cfg.est_ch
    sd["input_embedding.weight"] = g.normal((cfg.vocab, d), 1.0)
    sd["spk_embed_affine_layer.weight"] = g.linear(cfg.mel, cfg.spk_dim, gain=4.0)
    sd["spk_embed_affine_layer.bias"] = g.normal((cfg.mel,), 0.1)
    for emb in ("embed", "up_embed"):
        sd["encoder.%s.out.0.weight" % emb] = g.linear(d, d)
        sd["encoder.%s.out.0.bias" % emb] = g.normal((d,), 0.05)
        sd["encoder.%s.out.1.weight" % emb] = g.gamma(d)
        sd["encoder.%s.out.1.bias" % emb] = g.beta(d)
    sd["encoder.after_norm.weight"] = g.gamma(d)
    sd["encoder.after_norm.bias"] = g.beta(d)
    sd["encoder.pre_lookahead_layer.conv1.weight"] = g.conv(d, d, cfg.pre_lookahead + 1)
    sd["encoder.pre_lookahead_layer.conv1.bias"] = g.normal((d,), 0.05)
    sd["encoder.pre_lookahead_layer.conv2.weight"] = g.conv(d, d, 3)
    sd["encoder.pre_lookahead_layer.conv2.bias"] = g.normal((d,), 0.05)
    for i in range(cfg.enc_blocks):
        _conformer_layer(g, sd, "encoder.encoders.%d." % i, d, cfg.enc_heads, cfg.ffn)
    sd["encoder.up_layer.conv.weight"] = g.conv(d, d, 5)
    sd["encoder.up_layer.conv.bias"] = g.normal((d,), 0.05)
    for i in range(cfg.up_blocks):
        _conformer_layer(g, sd, "encoder.up_encoders.%d." % i, d, cfg.enc_heads, cfg.ffn)
    sd["encoder_proj.weight"] = g.linear(cfg.mel, d)
    sd["encoder_proj.bias"] = g.normal((cfg.mel,), 0.05)

    e = "decoder.estimator."
    cin, tdim, inner = 4 * cfg.mel, 4 * C, cfg.est_heads * 64
    sd[e + "time_mlp.linear_1.weight"] = g.linear(tdim, cin)
    sd[e + "time_mlp.linear_1.bias"] = g.normal((tdim,), 0.05)
    sd[e + "time_mlp.linear_2.weight"] = g.linear(tdim, tdim)
    sd[e + "time_mlp.linear_2.bias"] = g.normal((tdim,), 0.05)

    def resnet(p, din, dout):
        sd[p + "mlp.1.weight"] = g.linear(dout, tdim)
        sd[p + "mlp.1.bias"] = g.normal((dout,), 0.05)
        for b, di in (("block1", din), ("block2", dout)):
            sd[p + b + ".block.0.weight"] = g.conv(dout, di, 3)
            sd[p + b + ".block.0.bias"] = g.normal((dout,), 0.05)
            sd[p + b + ".block.2.weight"] = g.gamma(dout)
            sd[p + b + ".block.2.bias"] = g.beta(dout)
        sd[p + "res_conv.weight"] = g.conv(dout, din, 1)
        sd[p + "res_conv.bias"] = g.normal((dout,), 0.05)

    def tblock(p):
        sd[p + "norm1.weight"] = g.gamma(C); sd[p + "norm1.bias"] = g.beta(C)
        for n in ("to_q", "to_k", "to_v"):
            sd[p + "attn1.%s.weight" % n] = g.linear(inner, C)
        sd[p + "attn1.to_out.0.weight"] = g.linear(C, inner, gain=0.5)
        sd[p + "attn1.to_out.0.bias"] = g.normal((C,), 0.05)
        sd[p + "norm3.weight"] = g.gamma(C); sd[p + "norm3.bias"] = g.beta(C)
        sd[p + "ff.net.0.proj.weight"] = g.linear(4 * C, C)
        sd[p + "ff.net.0.proj.bias"] = g.normal((4 * C,), 0.05)
        sd[p + "ff.net.2.weight"] = g.linear(C, 4 * C, gain=0.5)
        sd[p + "ff.net.2.bias"] = g.normal((C,), 0.05)

    resnet(e + "down_blocks.0.0.", cin, C)
    for j in range(cfg.est_blocks):
        tblock(e + "down_blocks.0.1.%d." % j)
    sd[e + "down_blocks.0.2.weight"] = g.conv(C, C, 3); sd[e + "down_blocks.0.2.bias"] = g.normal((C,), 0.05)
    for i in range(cfg.est_mid):
        resnet(e + "mid_blocks.%d.0." % i, C, C)
        for j in range(cfg.est_blocks):
            tblock(e + "mid_blocks.%d.1.%d." % (i, j))
    resnet(e + "up_blocks.0.0.", 2 * C, C)
    for j in range(cfg.est_blocks):
        tblock(e + "up_blocks.0.1.%d." % j)
    sd[e + "up_blocks.0.2.weight"] = g.conv(C, C, 3); sd[e + "up_blocks.0.2.bias"] = g.normal((C,), 0.05)
    sd[e + "final_block.block.0.weight"] = g.conv(C, C, 3); sd[e + "final_block.block.0.bias"] = g.normal((C,), 0.05)
    sd[e + "final_block.block.2.weight"] = g.gamma(C); sd[e + "final_block.block.2.bias"] = g.beta(C)
    sd[e + "final_proj.weight"] = g.conv(cfg.mel, C, 1); sd[e + "final_proj.bias"] = g.normal((cfg.mel,), 0.05)
    return to_bf16_grid(sd)


def make_flow_dit(cfg: FlowConfig, seed=1990):
    """Keys of cosyvoice.flow.flow.CausalMaskedDiffWithDiT (flow/flow.py:284-318) with PreLookaheadLayer and the DiT estimator
    (flow/DiT/dit.py:104-144, flow/DiT/modules.py).  The adaLN-zero projections are NOT zero here (a trained model's are not either)."""
    assert cfg.estimator == "dit"
    g, sd, d, D, Cp = _Gen(seed), {}, cfg.dim, cfg.est_ch, cfg.ffn
    sd["input_embedding.weight"] = g.normal((cfg.vocab, d), 1.0)
    sd["spk_embed_affine_layer.weight"] = g.linear(cfg.mel, cfg.spk_dim, gain=4.0)
    sd["spk_embed_affine_layer.bias"] = g.normal((cfg.mel,), 0.1)
    sd["pre_lookahead_layer.conv1.weight"] = g.conv(Cp, d, cfg.pre_lookahead + 1)
    sd["pre_lookahead_layer.conv1.bias"] = g.normal((Cp,), 0.05)
    sd["pre_lookahead_layer.conv2.weight"] = g.conv(d, Cp, 3)
    sd["pre_lookahead_layer.conv2.bias"] = g.normal((d,), 0.05)
    e = "decoder.estimator."
    sd[e + "time_embed.time_mlp.0.weight"] = g.linear(D, 256); sd[e + "time_embed.time_mlp.0.bias"] = g.normal((D,), 0.05)
    sd[e + "time_embed.time_mlp.2.weight"] = g.linear(D, D); sd[e + "time_embed.time_mlp.2.bias"] = g.normal((D,), 0.05)
    sd[e + "input_embed.proj.weight"] = g.linear(D, 4 * cfg.mel); sd[e + "input_embed.proj.bias"] = g.normal((D,), 0.05)
    for c in ("conv1", "conv2"):
        sd[e + "input_embed.conv_pos_embed.%s.0.weight" % c] = g.conv(D, D // 16, 31)
        sd[e + "input_embed.conv_pos_embed.%s.0.bias" % c] = g.normal((D,), 0.05)
    inner, ffi = cfg.est_heads * 64, cfg.est_mid * D
    for i in range(cfg.est_blocks):
        p = e + "transformer_blocks.%d." % i
        sd[p + "attn_norm.linear.weight"] = g.linear(6 * D, D, gain=0.5); sd[p + "attn_norm.linear.bias"] = g.normal((6 * D,), 0.1)
        for n in ("to_q", "to_k", "to_v"):
            sd[p + "attn.%s.weight" % n] = g.linear(inner, D); sd[p + "attn.%s.bias" % n] = g.normal((inner,), 0.05)
        sd[p + "attn.to_out.0.weight"] = g.linear(D, inner, gain=0.5); sd[p + "attn.to_out.0.bias"] = g.normal((D,), 0.05)
        sd[p + "ff.ff.0.0.weight"] = g.linear(ffi, D); sd[p + "ff.ff.0.0.bias"] = g.normal((ffi,), 0.05)
        sd[p + "ff.ff.2.weight"] = g.linear(D, ffi, gain=0.5); sd[p + "ff.ff.2.bias"] = g.normal((D,), 0.05)
    sd[e + "norm_out.linear.weight"] = g.linear(2 * D, D, gain=0.5); sd[e + "norm_out.linear.bias"] = g.normal((2 * D,), 0.1)
    sd[e + "proj_out.weight"] = g.linear(cfg.mel, D); sd[e + "proj_out.bias"] = g.normal((cfg.mel,), 0.05)
    return to_bf16_grid(sd)


def make_hift(cfg: HiftConfig, seed=1988):
    """Keys of cosyvoice.hifigan.generator.HiFTGenerator (generator.py:378-476) or, for cfg.causal, CausalHiFTGenerator (:572-682) incl.
    f0_predictor; weight-norm parametrisation keys `parametrizations.weight.original0/1` (g, v) as torch >= 2.1 stores them."""
    g, sd = _Gen(seed), {}

    def wn(p, v, gain_dim0=True):
        # weight = g * v / ||v||, norm over all dims but 0 (ConvTranspose1d: dim 0 is C_in)  — SURVEY Appendix C.11
        norm = v.reshape(v.shape[0], -1).norm(dim=1).reshape(-1, 1, 1)
        sd[p + "parametrizations.weight.original0"] = norm * (1.0 + g.normal((v.shape[0], 1, 1), 0.1))
        sd[p + "parametrizations.weight.original1"] = v

    sd["m_source.l_linear.weight"] = g.normal((1, cfg.harmonics + 1), 1.0)
    sd["m_source.l_linear.bias"] = g.normal((1,), 0.1)
    sd["conv_pre.bias"] = g.normal((cfg.base,), 0.05)
    wn("conv_pre.", g.conv(cfg.base, cfg.mel, cfg.look_right + 1 if cfg.causal else 7))       # causal: CausalConv1d(k = look_right + 1, 'right')
    ch = cfg.base
    for i, (u, k) in enumerate(zip(cfg.ups, cfg.up_k)):
        sd["ups.%d.bias" % i] = g.normal((ch // 2,), 0.05)
        if cfg.causal:                                        # CausalConv1dUpsample: a Conv1d [C_out, C_in, k] behind a nearest upsampling
            wn("ups.%d." % i, g.conv(ch // 2, ch, k))
        else:
            wn("ups.%d." % i, g.normal((ch, ch // 2, k), 1.0 / np.sqrt(ch * k / u)))
        ch //= 2
    # source_downs: strides = cumprod([1] + ups[::-1][:-1])[::-1]   (generator.py:443-455)
    rates = np.cumprod([1] + cfg.ups[::-1][:-1])[::-1]
    for i, u in enumerate(rates):
        c = cfg.base // (2 ** (i + 1))
        k = 1 if u == 1 else int(u) * 2
        sd["source_downs.%d.weight" % i] = g.conv(c, cfg.n_fft + 2, k, gain=0.5)
        sd["source_downs.%d.bias" % i] = g.normal((c,), 0.05)

    def resblock(p, c, k):
        for grp in ("convs1", "convs2"):
            for j in range(len(cfg.res_d)):
                sd[p + "%s.%d.bias" % (grp, j)] = g.normal((c,), 0.05)
                wn(p + "%s.%d." % (grp, j), g.conv(c, c, k, gain=0.7))
        for grp in ("activations1", "activations2"):
            for j in range(len(cfg.res_d)):
                sd[p + "%s.%d.alpha" % (grp, j)] = 1.0 + g.normal((c,), 0.2).abs()

    for i in range(len(cfg.ups)):
        resblock("source_resblocks.%d." % i, cfg.base // (2 ** (i + 1)), cfg.src_k[i])
    for i in range(len(cfg.ups)):
        for j, k in enumerate(cfg.res_k):
            resblock("resblocks.%d." % (i * len(cfg.res_k) + j), cfg.base // (2 ** (i + 1)), k)
    sd["conv_post.bias"] = g.normal((cfg.n_fft + 2,), 0.05)
    wn("conv_post.", g.conv(cfg.n_fft + 2, ch, 7, gain=0.3))
    cin = cfg.mel
    for j in range(5):
        sd["f0_predictor.condnet.%d.bias" % (2 * j)] = g.normal((cfg.f0_ch,), 0.05)
        wn("f0_predictor.condnet.%d." % (2 * j), g.conv(cfg.f0_ch, cin, 4 if (cfg.causal and j == 0) else 3, gain=1.4))
        cin = cfg.f0_ch
    sd["f0_predictor.classifier.weight"] = g.linear(1, cfg.f0_ch, gain=30.0)
    sd["f0_predictor.classifier.bias"] = g.normal((1,), 1.0)
    return sd


def synthetic_utterance(llm: LLMConfig, flow: FlowConfig, n_prompt_tok=87, n_prompt_text=12, n_text=30, seed=1986):
    """SURVEY.md §8d synthetic inputs (U10 by default): ids uniform, prompt mel randn*2-5, x-vector randn."""
    g = torch.Generator().manual_seed(seed)
    return dict(
        text=torch.randint(0, llm.text_vocab, (1, n_text), generator=g, dtype=torch.int32),
        prompt_text=torch.randint(0, llm.text_vocab, (1, n_prompt_text), generator=g, dtype=torch.int32),
        llm_prompt_speech_token=torch.randint(0, llm.speech_token_size, (1, n_prompt_tok), generator=g, dtype=torch.int32),
        flow_prompt_speech_token=torch.randint(0, flow.vocab, (1, n_prompt_tok), generator=g, dtype=torch.int32),
        prompt_speech_feat=torch.randn(1, 2 * n_prompt_tok, flow.mel, generator=g) * 2.0 - 5.0,
        llm_embedding=torch.randn(1, flow.spk_dim, generator=g),
        flow_embedding=torch.randn(1, flow.spk_dim, generator=g),
    )


def ref_small_flow():
    """Flow config used for golden vectors from the real reference: the reference hard-codes 512 channels in
    PreLookaheadLayer / Upsample1D (upsample_encoder.py:203,217), so dim stays 512; everything else is shrunk."""
    return FlowConfig(vocab=60, dim=512, enc_heads=8, ffn=256, enc_blocks=1, up_blocks=1, spk_dim=192, est_ch=64, est_heads=1,
                      est_blocks=1, est_mid=1)


# ------------------------------------------------------------------------------------------------------------------------------------
# CosyVoice-300M (cosyvoice_amd/cosyvoice1.py): reference key names of TransformerLM / MaskedDiffWithXvec; the 22.05 kHz HiFTGenerator comes
# from make_hift(configs.cv1()[1]).  Validated by strict=True loads into the real classes (tests/golden/make_golden_cv1.py).
# ------------------------------------------------------------------------------------------------------------------------------------
def _espnet_encoder(g, sd, p, kind, d_in, d, heads, ffn, blocks):
    """ConformerEncoder (no CNN module, no macaron) / TransformerEncoder with rel_pos_espnet + rel_selfattn (transformer/encoder.py:330-474)."""
    sd[p + "embed.out.0.weight"], sd[p + "embed.out.0.bias"] = g.linear(d, d_in), g.beta(d)
    sd[p + "embed.out.1.weight"], sd[p + "embed.out.1.bias"] = g.gamma(d), g.beta(d)
    n_att, n_ff = ("norm_mha", "norm_ff") if kind == "conformer" else ("norm1", "norm2")
    for i in range(blocks):
        q = p + "encoders.%d." % i
        for n in ("linear_q", "linear_k", "linear_v", "linear_out"):
            sd[q + "self_attn.%s.weight" % n], sd[q + "self_attn.%s.bias" % n] = g.linear(d, d), g.beta(d)
        sd[q + "self_attn.linear_pos.weight"] = g.linear(d, d)
        sd[q + "self_attn.pos_bias_u"], sd[q + "self_attn.pos_bias_v"] = g.normal((heads, d // heads), 0.3), g.normal((heads, d // heads), 0.3)
        sd[q + "feed_forward.w_1.weight"], sd[q + "feed_forward.w_1.bias"] = g.linear(ffn, d), g.beta(ffn)
        sd[q + "feed_forward.w_2.weight"], sd[q + "feed_forward.w_2.bias"] = g.linear(d, ffn, gain=0.5), g.beta(d)
        for n in (n_ff, n_att):
            sd[q + n + ".weight"], sd[q + n + ".bias"] = g.gamma(d), g.beta(d)
    sd[p + "after_norm.weight"], sd[p + "after_norm.bias"] = g.gamma(d), g.beta(d)


def make_cv1_llm(cfg: CV1Config, seed=2001):
    """cosyvoice.llm.llm.TransformerLM (llm/llm.py:35-82)."""
    g, sd = _Gen(seed), {}
    D = cfg.llm_dim
    sd["text_embedding.weight"] = g.normal((cfg.text_vocab, cfg.text_enc_in), 1.0)
    _espnet_encoder(g, sd, "text_encoder.", "conformer", cfg.text_enc_in, D, cfg.text_heads, cfg.text_ffn, cfg.text_blocks)
    sd["text_encoder_affine_layer.weight"], sd["text_encoder_affine_layer.bias"] = g.linear(D, D), g.beta(D)
    sd["llm_embedding.weight"] = g.normal((2, D), 1.0)
    _espnet_encoder(g, sd, "llm.", "transformer", D, D, cfg.llm_heads, cfg.llm_ffn, cfg.llm_blocks)
    sd["llm_decoder.weight"], sd["llm_decoder.bias"] = g.linear(cfg.speech_token_size + 1, D, gain=6.0), g.normal((cfg.speech_token_size + 1,), 0.1)
    sd["speech_embedding.weight"] = g.normal((cfg.speech_token_size, D), 1.0)
    sd["spk_embed_affine_layer.weight"], sd["spk_embed_affine_layer.bias"] = g.linear(D, cfg.spk_dim), g.beta(D)
    return sd


def make_cv1_flow(cfg: CV1Config, seed=2002):
    """cosyvoice.flow.flow.MaskedDiffWithXvec (flow/flow.py:25-61) with InterpolateRegulator, ConditionalCFM and the U-Net ConditionalDecoder
    (flow/decoder.py:88-205; Matcha ResnetBlock1D / Block1D / Downsample1D / Upsample1D / BasicTransformerBlock keys)."""
    g, sd = _Gen(seed), {}
    mel = cfg.mel
    sd["input_embedding.weight"] = g.normal((cfg.speech_token_size, cfg.flow_dim), 1.0)
    sd["spk_embed_affine_layer.weight"], sd["spk_embed_affine_layer.bias"] = g.linear(mel, cfg.spk_dim), g.beta(mel)
    _espnet_encoder(g, sd, "encoder.", "conformer", cfg.flow_dim, cfg.flow_dim, cfg.flow_heads, cfg.flow_ffn, cfg.flow_blocks)
    sd["encoder_proj.weight"], sd["encoder_proj.bias"] = g.linear(mel, cfg.flow_dim), g.beta(mel)
    r = "length_regulator.model."
    for i in range(cfg.regulator_layers):
        sd[r + "%d.weight" % (3 * i)], sd[r + "%d.bias" % (3 * i)] = g.conv(mel, mel, 3), g.beta(mel)
        sd[r + "%d.weight" % (3 * i + 1)], sd[r + "%d.bias" % (3 * i + 1)] = g.gamma(mel), g.beta(mel)
    sd[r + "%d.weight" % (3 * cfg.regulator_layers)], sd[r + "%d.bias" % (3 * cfg.regulator_layers)] = g.conv(mel, mel, 1), g.beta(mel)
    e = "decoder.estimator."
    temb = cfg.est_ch[0] * 4
    sd[e + "time_mlp.linear_1.weight"], sd[e + "time_mlp.linear_1.bias"] = g.linear(temb, 4 * mel), g.beta(temb)
    sd[e + "time_mlp.linear_2.weight"], sd[e + "time_mlp.linear_2.bias"] = g.linear(temb, temb), g.beta(temb)
    inner = cfg.est_heads * cfg.est_head_dim

    def block1d(p, cin, cout):
        sd[p + "block.0.weight"], sd[p + "block.0.bias"] = g.conv(cout, cin, 3), g.beta(cout)
        sd[p + "block.1.weight"], sd[p + "block.1.bias"] = g.gamma(cout), g.beta(cout)

    def stage(p, cin, cout):
        block1d(p + "0.block1.", cin, cout)
        block1d(p + "0.block2.", cout, cout)
        sd[p + "0.mlp.1.weight"], sd[p + "0.mlp.1.bias"] = g.linear(cout, temb), g.beta(cout)
        sd[p + "0.res_conv.weight"], sd[p + "0.res_conv.bias"] = g.conv(cout, cin, 1), g.beta(cout)
        for j in range(cfg.est_blocks):
            t = p + "1.%d." % j
            sd[t + "norm1.weight"], sd[t + "norm1.bias"] = g.gamma(cout), g.beta(cout)
            for n in ("to_q", "to_k", "to_v"):
                sd[t + "attn1.%s.weight" % n] = g.linear(inner, cout)
            sd[t + "attn1.to_out.0.weight"], sd[t + "attn1.to_out.0.bias"] = g.linear(cout, inner, gain=0.5), g.beta(cout)
            sd[t + "norm3.weight"], sd[t + "norm3.bias"] = g.gamma(cout), g.beta(cout)
            sd[t + "ff.net.0.proj.weight"], sd[t + "ff.net.0.proj.bias"] = g.linear(4 * cout, cout), g.beta(4 * cout)
            sd[t + "ff.net.2.weight"], sd[t + "ff.net.2.bias"] = g.linear(cout, 4 * cout, gain=0.5), g.beta(cout)

    ch, cout = cfg.est_ch, 4 * mel
    for i, c in enumerate(ch):
        cin, cout = cout, c
        stage(e + "down_blocks.%d." % i, cin, cout)
        if i < len(ch) - 1:                                   # Downsample1D: Conv1d(c, c, 3, stride 2)
            sd[e + "down_blocks.%d.2.conv.weight" % i], sd[e + "down_blocks.%d.2.conv.bias" % i] = g.conv(c, c, 3), g.beta(c)
        else:
            sd[e + "down_blocks.%d.2.weight" % i], sd[e + "down_blocks.%d.2.bias" % i] = g.conv(c, c, 3), g.beta(c)
    for i in range(cfg.est_mid):
        stage(e + "mid_blocks.%d." % i, ch[-1], ch[-1])
    up = ch[::-1] + [ch[0]]
    for i in range(len(up) - 1):
        stage(e + "up_blocks.%d." % i, 2 * up[i], up[i + 1])
        c = up[i + 1]
        if i < len(up) - 2:                                   # Upsample1D: ConvTranspose1d(c, c, 4, 2, 1), weight [in, out, k]
            sd[e + "up_blocks.%d.2.conv.weight" % i], sd[e + "up_blocks.%d.2.conv.bias" % i] = g.normal((c, c, 4), 1.0 / np.sqrt(2 * c)), g.beta(c)
        else:
            sd[e + "up_blocks.%d.2.weight" % i], sd[e + "up_blocks.%d.2.bias" % i] = g.conv(c, c, 3), g.beta(c)
    block1d(e + "final_block.", up[-1], up[-1])
    sd[e + "final_proj.weight"], sd[e + "final_proj.bias"] = g.conv(mel, up[-1], 1), g.beta(mel)
    return sd
