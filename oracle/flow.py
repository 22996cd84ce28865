"""Oracle: token -> mel flow-matching decoder (TEST INFRASTRUCTURE — see oracle/__init__.py).

Restates, in plain torch fp32 over a reference-named state dict:
  cosyvoice/flow/flow.py:235-281                 CausalMaskedDiffWithXvec.inference
  cosyvoice/transformer/upsample_encoder.py:37-103,244-321   Upsample1D, PreLookaheadLayer, UpsampleConformerEncoder.forward
  cosyvoice/transformer/subsampling.py:83-113    LinearNoSubsampling
  cosyvoice/transformer/embedding.py:201-302     EspnetRelPositionalEncoding
  cosyvoice/transformer/encoder_layer.py:160-236 ConformerEncoderLayer (no macaron, no conv module)
  cosyvoice/transformer/attention.py:200-330     RelPositionMultiHeadedAttention (+ rel_shift)
  cosyvoice/flow/flow_matching.py:71-124,196-227 solve_euler / CausalConditionalCFM.forward
  cosyvoice/flow/decoder.py:36-85,405-494        CausalConv1d, CausalBlock1D, CausalResnetBlock1D, CausalConditionalDecoder.forward
  cosyvoice/utils/mask.py:127-158,161-236        subsequent_chunk_mask / add_optional_chunk_mask (static-chunk branch)
  cosyvoice/utils/common.py:188-196              mask_to_bias
and the third-party Matcha-TTS / diffusers pieces listed in SURVEY.md Appendix B (un-pinned: PARITY UNPINNED there).
"""
import math

import torch
import torch.nn.functional as F


# ------------------------------------------------------------------------------------------ encoder
# Operand precision of the Linear / Conv1d products.  False (default) is the fp32 restatement pinned against the reference
# goldens.  True mirrors the product's "bf16" mode: the activation operand (after any fused input activation) is rounded to
# bfloat16 (round-to-nearest-even) before the product, weights are the same bf16-valued tensors, accumulation is fp32.
_BF16_ACT = False


class bf16_act:
    """with oracle.flow.bf16_act(): ...   -> products take bf16-rounded activations (cosyvoice_amd/csrc/gemm_conv.h ABF16)."""
    def __enter__(self):
        global _BF16_ACT
        self.prev, _BF16_ACT = _BF16_ACT, True
    def __exit__(self, *a):
        global _BF16_ACT
        _BF16_ACT = self.prev


def _r(x):
    return x.bfloat16().float() if _BF16_ACT else x


def _attn_pv(attn, v):
    """softmax(S) @ V.  bf16 mode mirrors the flash kernel: the numerator uses bf16-rounded un-normalised probabilities and bf16 V,
    the denominator the unrounded ones (cosyvoice_amd/csrc/attention.h attention_bf16_kernel)."""
    if not _BF16_ACT:
        return torch.matmul(attn, v)
    # attn = e / sum(e) with e = exp(s - max): recover e up to the row scale (any positive row scale cancels in num / den)
    e = attn / attn.amax(dim=-1, keepdim=True).clamp_min(1e-30)
    return torch.matmul(_r(e), _r(v)) / e.sum(dim=-1, keepdim=True)


def _linear(x, w, b=None):
    return F.linear(_r(x), w, b)


def _conv1d(x, w, b=None, **kw):
    return F.conv1d(_r(x), w, b, **kw)


def rel_pos_emb(size, d_model):
    """pos_emb[:, m] encodes relative position (size-1-m), m in [0, 2*size-1)   (embedding.py:225-302, offset=0)."""
    pos = torch.arange(size - 1, -size, -1, dtype=torch.float32).unsqueeze(1)
    div = torch.exp(torch.arange(0, d_model, 2, dtype=torch.float32) * -(math.log(10000.0) / d_model))
    pe = torch.zeros(2 * size - 1, d_model)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe.unsqueeze(0)


def rel_shift(x):
    """attention.py:222-244."""
    b, h, t, n = x.shape
    zero_pad = torch.zeros((b, h, t, 1), dtype=x.dtype)
    x_padded = torch.cat([zero_pad, x], dim=-1).view(b, h, n + 1, t)
    return x_padded[:, :, 1:].view_as(x)[:, :, :, : n // 2 + 1]


def subsequent_chunk_mask(size, chunk):
    pos = torch.arange(size)
    block_value = (torch.div(pos, chunk, rounding_mode="trunc") + 1) * chunk
    return pos.unsqueeze(0) < block_value.unsqueeze(1)


def embed(sd, p, x, d):
    """LinearNoSubsampling: Linear -> LayerNorm(1e-5) -> * sqrt(d)   (subsampling.py:83-113, embedding.py:256-270)."""
    x = _linear(x, sd[p + "out.0.weight"], sd[p + "out.0.bias"])
    x = F.layer_norm(x, (d,), sd[p + "out.1.weight"], sd[p + "out.1.bias"], 1e-5)
    return x * math.sqrt(d)


def conformer_layer(sd, p, x, mask, pos_emb, heads):
    """Pre-norm MHA(rel-pos) + FFN(SiLU)   (encoder_layer.py:201-236 with macaron/conv disabled)."""
    b, t, d = x.shape
    dk = d // heads
    r = x
    n = F.layer_norm(x, (d,), sd[p + "norm_mha.weight"], sd[p + "norm_mha.bias"], 1e-12)
    q = _linear(n, sd[p + "self_attn.linear_q.weight"], sd[p + "self_attn.linear_q.bias"]).view(b, t, heads, dk)
    k = _linear(n, sd[p + "self_attn.linear_k.weight"], sd[p + "self_attn.linear_k.bias"]).view(b, t, heads, dk).transpose(1, 2)
    v = _linear(n, sd[p + "self_attn.linear_v.weight"], sd[p + "self_attn.linear_v.bias"]).view(b, t, heads, dk).transpose(1, 2)
    pp = _linear(pos_emb, sd[p + "self_attn.linear_pos.weight"]).view(1, -1, heads, dk).transpose(1, 2)
    qu = (q + sd[p + "self_attn.pos_bias_u"]).transpose(1, 2)
    qv = (q + sd[p + "self_attn.pos_bias_v"]).transpose(1, 2)
    ac = torch.matmul(_r(qu), _r(k).transpose(-2, -1))          # bf16 mode: q, k, v and the probabilities are MFMA operands too
    bd = rel_shift(torch.matmul(qv, pp.transpose(-2, -1)))
    scores = (ac + bd) / math.sqrt(dk)
    m = mask.unsqueeze(1).eq(0)                                   # attention.py:108-114
    scores = scores.masked_fill(m, -float("inf"))
    attn = torch.softmax(scores, dim=-1).masked_fill(m, 0.0)
    a = _attn_pv(attn, v).transpose(1, 2).contiguous().view(b, t, d)
    x = r + _linear(a, sd[p + "self_attn.linear_out.weight"], sd[p + "self_attn.linear_out.bias"])
    r = x
    n = F.layer_norm(x, (d,), sd[p + "norm_ff.weight"], sd[p + "norm_ff.bias"], 1e-12)
    f = _linear(F.silu(_linear(n, sd[p + "feed_forward.w_1.weight"], sd[p + "feed_forward.w_1.bias"])),
                 sd[p + "feed_forward.w_2.weight"], sd[p + "feed_forward.w_2.bias"])
    return r + f


def encoder(sd, cfg, xs, context=None, streaming=False):
    """UpsampleConformerEncoder.forward for batch 1, full-length mask (upsample_encoder.py:244-307).
    xs [1,T,d] masked token embeddings; context [1,3,d] or None.  Returns h [1,2T,d]."""
    p, d, heads = "encoder.", cfg.dim, cfg.enc_heads
    T = xs.shape[1]
    x = embed(sd, p + "embed.", xs, d)
    pos = rel_pos_emb(T, d)
    if context is not None and context.shape[1] != 0:
        ctx = embed(sd, p + "embed.", context, d)
    else:
        ctx = None
    mask = torch.ones(1, T, T, dtype=torch.bool)
    if streaming:
        mask = mask & subsequent_chunk_mask(T, cfg.chunk).unsqueeze(0)
    # PreLookaheadLayer (upsample_encoder.py:82-103)
    o = x.transpose(1, 2)
    if ctx is None:
        o = F.pad(o, (0, cfg.pre_lookahead))
    else:
        o = torch.cat([o, ctx.transpose(1, 2)], dim=2)
    o = F.leaky_relu(_conv1d(o, sd[p + "pre_lookahead_layer.conv1.weight"], sd[p + "pre_lookahead_layer.conv1.bias"]))
    o = F.pad(o, (2, 0))
    o = _conv1d(o, sd[p + "pre_lookahead_layer.conv2.weight"], sd[p + "pre_lookahead_layer.conv2.bias"])
    x = o.transpose(1, 2) + x
    for i in range(cfg.enc_blocks):
        x = conformer_layer(sd, p + "encoders.%d." % i, x, mask, pos, heads)
    # Upsample1D: nearest x2, left pad 4, Conv1d k5 (upsample_encoder.py:59-63)
    o = F.interpolate(x.transpose(1, 2), scale_factor=2.0, mode="nearest")
    o = F.pad(o, (4, 0))
    o = _conv1d(o, sd[p + "up_layer.conv.weight"], sd[p + "up_layer.conv.bias"])
    x = o.transpose(1, 2)
    T2 = x.shape[1]
    x = embed(sd, p + "up_embed.", x, d)
    pos = rel_pos_emb(T2, d)
    mask = torch.ones(1, T2, T2, dtype=torch.bool)
    if streaming:
        mask = mask & subsequent_chunk_mask(T2, cfg.chunk * 2).unsqueeze(0)
    for i in range(cfg.up_blocks):
        x = conformer_layer(sd, p + "up_encoders.%d." % i, x, mask, pos, heads)
    return F.layer_norm(x, (d,), sd[p + "after_norm.weight"], sd[p + "after_norm.bias"], 1e-5)


# ------------------------------------------------------------------------------------------ estimator
def sinusoidal_pos_emb(t, dim, scale=1000):
    """matcha SinusoidalPosEmb (Appendix B)."""
    half = dim // 2
    emb = math.log(10000) / (half - 1)
    emb = torch.exp(torch.arange(half).float() * -emb)
    emb = scale * t.unsqueeze(1) * emb.unsqueeze(0)
    return torch.cat((emb.sin(), emb.cos()), dim=-1)


def causal_block(sd, p, x, mask):
    """CausalBlock1D: (x*mask) -> causal conv k3 -> LayerNorm over channels -> Mish, * mask   (flow/decoder.py:65-78)."""
    y = _conv1d(F.pad(x * mask, (2, 0)), sd[p + "block.0.weight"], sd[p + "block.0.bias"])
    y = F.layer_norm(y.transpose(1, 2), (y.shape[1],), sd[p + "block.2.weight"], sd[p + "block.2.bias"], 1e-5).transpose(1, 2)
    return F.mish(y) * mask


def resnet_block(sd, p, x, mask, temb):
    """matcha ResnetBlock1D.forward with CausalBlock1D blocks (Appendix B; flow/decoder.py:81-85)."""
    h = causal_block(sd, p + "block1.", x, mask)
    h = h + _linear(F.mish(temb), sd[p + "mlp.1.weight"], sd[p + "mlp.1.bias"]).unsqueeze(-1)
    h = causal_block(sd, p + "block2.", h, mask)
    return h + _conv1d(x * mask, sd[p + "res_conv.weight"], sd[p + "res_conv.bias"])


def transformer_block(sd, p, x, bias, heads):
    """matcha BasicTransformerBlock (self-attention only) + diffusers Attention/GELU (Appendix B)."""
    b, t, c = x.shape
    n = F.layer_norm(x, (c,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-5)
    q = _linear(n, sd[p + "attn1.to_q.weight"]).view(b, t, heads, 64).transpose(1, 2)
    k = _linear(n, sd[p + "attn1.to_k.weight"]).view(b, t, heads, 64).transpose(1, 2)
    v = _linear(n, sd[p + "attn1.to_v.weight"]).view(b, t, heads, 64).transpose(1, 2)
    s = torch.matmul(_r(q), _r(k).transpose(-2, -1)) / 8.0 + bias.unsqueeze(1)
    a = _attn_pv(torch.softmax(s, dim=-1), v).transpose(1, 2).reshape(b, t, heads * 64)
    x = _linear(a, sd[p + "attn1.to_out.0.weight"], sd[p + "attn1.to_out.0.bias"]) + x
    n = F.layer_norm(x, (c,), sd[p + "norm3.weight"], sd[p + "norm3.bias"], 1e-5)
    f = _linear(F.gelu(_linear(n, sd[p + "ff.net.0.proj.weight"], sd[p + "ff.net.0.proj.bias"])),
                 sd[p + "ff.net.2.weight"], sd[p + "ff.net.2.bias"])
    return f + x


def estimator(sd, cfg, x, mask, mu, t, spks, cond, streaming=False):
    """CausalConditionalDecoder.forward for channels=[256] (no temporal down/up-sampling)   (flow/decoder.py:405-494).
    x, mu, cond [B,80,T]; mask [B,1,T]; t [B]; spks [B,80]  ->  [B,80,T]."""
    e, heads = "decoder.estimator.", cfg.est_heads
    temb = sinusoidal_pos_emb(t, 4 * cfg.mel)
    temb = _linear(F.silu(_linear(temb, sd[e + "time_mlp.linear_1.weight"], sd[e + "time_mlp.linear_1.bias"])),
                    sd[e + "time_mlp.linear_2.weight"], sd[e + "time_mlp.linear_2.bias"])
    T = x.shape[-1]
    h = torch.cat([x, mu, spks.unsqueeze(-1).expand(-1, -1, T), cond], dim=1)

    def attn_bias():
        m = mask.bool()
        if streaming:
            am = m & subsequent_chunk_mask(T, 2 * cfg.chunk).unsqueeze(0)       # add_optional_chunk_mask static branch
        else:
            am = m.repeat(1, T, 1)
        return (1.0 - am.float()) * -1.0e10                                      # mask_to_bias

    def stage(p, h, nblk):
        h = resnet_block(sd, p + "0.", h, mask, temb)
        y = h.transpose(1, 2).contiguous()
        b = attn_bias()
        for j in range(nblk):
            y = transformer_block(sd, p + "1.%d." % j, y, b, heads)
        return y.transpose(1, 2).contiguous()

    h = stage(e + "down_blocks.0.", h, cfg.est_blocks)
    skip = h
    h = _conv1d(F.pad(h * mask, (2, 0)), sd[e + "down_blocks.0.2.weight"], sd[e + "down_blocks.0.2.bias"])
    for i in range(cfg.est_mid):
        h = stage(e + "mid_blocks.%d." % i, h, cfg.est_blocks)
    h = torch.cat([h[:, :, : skip.shape[-1]], skip], dim=1)
    h = stage(e + "up_blocks.0.", h, cfg.est_blocks)
    h = _conv1d(F.pad(h * mask, (2, 0)), sd[e + "up_blocks.0.2.weight"], sd[e + "up_blocks.0.2.bias"])
    h = causal_block(sd, e + "final_block.", h, mask)
    out = _conv1d(h * mask, sd[e + "final_proj.weight"], sd[e + "final_proj.bias"])
    return out * mask


def cfm_noise(T, seed=0):
    """CausalConditionalCFM.__init__: set_all_random_seed(0); randn([1,80,50*300])   (flow_matching.py:199-200)."""
    g = torch.Generator().manual_seed(seed)
    return torch.randn([1, 80, 50 * 300], generator=g)[:, :, :T]


def solve_euler(sd, cfg, mu, mask, spks, cond, n_timesteps=10, streaming=False, noise=None, temperature=1.0):
    """CausalConditionalCFM.forward + ConditionalCFM.solve_euler with classifier-free guidance (flow_matching.py:71-124,203-227)."""
    T = mu.shape[2]
    x = (cfm_noise(T) if noise is None else noise[:, :, :T]) * temperature
    t_span = torch.linspace(0, 1, n_timesteps + 1)
    t_span = 1 - torch.cos(t_span * 0.5 * torch.pi)
    t, dt = t_span[0].unsqueeze(0), t_span[1] - t_span[0]
    zeros = torch.zeros_like(mu)
    for step in range(1, len(t_span)):
        x_in = torch.cat([x, x], 0)
        mask_in = torch.cat([mask, mask], 0)
        mu_in = torch.cat([mu, zeros], 0)
        t_in = torch.cat([t, t], 0)
        spks_in = torch.cat([spks, torch.zeros_like(spks)], 0)
        cond_in = torch.cat([cond, torch.zeros_like(cond)], 0)
        d = estimator(sd, cfg, x_in, mask_in, mu_in, t_in, spks_in, cond_in, streaming)
        d, d_cfg = d[:1], d[1:]
        d = (1.0 + cfg.cfg_rate) * d - cfg.cfg_rate * d_cfg
        x = x + dt * d
        t = t + dt
        if step < len(t_span) - 1:
            dt = t_span[step + 1] - t
    return x.float()


# ------------------------------------------------------------------------------------------ flow.inference
def inference(sd, cfg, token, prompt_token, prompt_feat, embedding, streaming=False, finalize=True, noise=None,
              n_timesteps=None, return_all=False):
    """CausalMaskedDiffWithXvec.inference (flow/flow.py:235-281). token/prompt_token [1,n] ints, prompt_feat [1,2p,80],
    embedding [1,192]  ->  mel [1,80,2*n_new]."""
    emb = _linear(F.normalize(embedding, dim=1), sd["spk_embed_affine_layer.weight"], sd["spk_embed_affine_layer.bias"])
    tok = torch.cat([prompt_token, token], dim=1).long().clamp(min=0)
    x = sd["input_embedding.weight"][tok[0]].unsqueeze(0)
    if finalize:
        h = encoder(sd, cfg, x, None, streaming)
    else:
        h = encoder(sd, cfg, x[:, : -cfg.pre_lookahead], x[:, -cfg.pre_lookahead:], streaming)
    mel_len1 = prompt_feat.shape[1]
    mel_len2 = h.shape[1] - mel_len1
    mu = _linear(h, sd["encoder_proj.weight"], sd["encoder_proj.bias"]).transpose(1, 2).contiguous()
    conds = torch.zeros(1, mel_len1 + mel_len2, cfg.mel)
    conds[:, :mel_len1] = prompt_feat
    conds = conds.transpose(1, 2)
    mask = torch.ones(1, 1, mel_len1 + mel_len2)
    feat = solve_euler(sd, cfg, mu, mask, emb, conds, n_timesteps or cfg.n_timesteps, streaming, noise)
    out = feat[:, :, mel_len1:]
    if return_all:
        return out, dict(h=h, mu=mu, spk=emb, conds=conds, feat=feat)
    return out
