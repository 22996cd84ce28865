"""CosyVoice-300M (first-generation CosyVoice) synthesis path - SURVEY.md section 8 row a18, BASELINE.json configs[0]:
"CosyVoice-300M-SFT inference_sft, one short utterance, PyTorch CPU eager (plumbing, no GPU)".

This module is the host-side plumbing of that configuration and nothing more: plain torch fp32 eager, on whatever device its tensors
live (the CPU in configs[0]).  It is NOT the accelerated path - the MI355X kernels serve CosyVoice2 / CosyVoice3 (model.py) - and it
never touches the HIP library.  What it mirrors, with the reference's call surface, state-dict key names (so `load()` takes the
published llm.pt / flow.pt / hift.pt) and per-request state:

  CosyVoiceModel.tts / token2wav / llm_job        cosyvoice/cli/model.py:27-242   (mel-overlap fade, flow cache, HiFT cache)
  TransformerLM.inference                         cosyvoice/llm/llm.py:162-223    (ConformerEncoder text encoder, causal; 14-block
                                                                                   TransformerEncoder stepped with forward_chunk + KV cache)
  MaskedDiffWithXvec.inference                    cosyvoice/flow/flow.py:102-146  (+ InterpolateRegulator.inference length_regulator.py:52-70,
                                                                                   ConditionalCFM.forward with flow_cache flow_matching.py:36-69,
                                                                                   ConditionalDecoder decoder.py:88-291, Matcha-TTS blocks)
  HiFTGenerator.inference at 22.05 kHz            cosyvoice/hifigan/generator.py:378-569 (upsample 8 x 8, SineGen type 1 :125-186)

Layout: the modules are functions over a flat state dict (reference key names), activations channel-last [T, C] inside the encoders and
channel-first [C, T] inside the conv stacks, batch 1 throughout (the reference asserts it, flow/flow.py:112).  Random draws (RAS
multinomial, CFM noise, SineGen phases / noise) use the global torch RNG in the reference's order, so a seeded run reproduces the
reference (tests/test_cosyvoice1.py compares against goldens made by the real classes).
"""
import math
import threading
import uuid as uuid_mod

import numpy as np
import torch
import torch.nn.functional as F


class _P:
    """A view of a flat state dict under a key prefix."""

    def __init__(self, sd, prefix=""):
        self.sd, self.prefix = sd, prefix

    def __call__(self, name):
        return self.sd[self.prefix + name]

    def get(self, name):
        return self.sd.get(self.prefix + name)

    def sub(self, name):
        return _P(self.sd, self.prefix + name)

    def count(self, stem):
        """number of consecutive `stem{i}.` groups present"""
        n = 0
        while any(k.startswith("%s%s%d." % (self.prefix, stem, n)) for k in self.sd):
            n += 1
        return n


def _linear(p, x):
    return F.linear(x, p("weight"), p.get("bias"))


def _ln(p, x, eps):
    return F.layer_norm(x, (x.shape[-1],), p("weight"), p("bias"), eps)


def _wn(p):
    """Conv weight, folding torch weight-norm if the checkpoint stores it (parametrizations.weight.original0/1 or legacy weight_g/weight_v)."""
    w = p.get("weight")
    if w is not None:
        return w
    g, v = p.get("parametrizations.weight.original0"), p.get("parametrizations.weight.original1")
    if g is None:
        g, v = p("weight_g"), p("weight_v")
    return v * (g / v.flatten(1).norm(dim=1).view(-1, *([1] * (v.dim() - 1))))


# ------------------------------------------------------------------------------------------------------------------------------------
# espnet-style encoders with relative positions (transformer/encoder.py, encoder_layer.py, attention.py:249-330, embedding.py:201-302)
# ------------------------------------------------------------------------------------------------------------------------------------
class EspnetEncoder:
    """ConformerEncoder as configured by cosyvoice.yaml (input 'linear', no CNN module, no macaron, swish FFN) or TransformerEncoder (input
    'linear_legacy', ReLU FFN); both rel_pos_espnet + rel_selfattn, pre-norm, final after_norm.  `causal`: static_chunk_size = 1."""

    def __init__(self, sd, prefix, heads, kind, causal=False):
        assert kind in ("conformer", "transformer")
        self.p, self.heads, self.kind, self.causal = _P(sd, prefix), heads, kind, causal
        self.n_layers = self.p.count("encoders.")
        self.d = self.p("after_norm.weight").shape[0]

    def output_size(self):
        return self.d

    def _embed(self, xs):
        x = _ln(self.p.sub("embed.out.1."), _linear(self.p.sub("embed.out.0."), xs), 1e-5)
        if self.kind == "transformer":
            x = F.relu(x)                                    # LegacyLinearNoSubsampling (subsampling.py:338-383)
        return x * math.sqrt(self.d)

    def _pos(self, n_keys, ref):
        """EspnetRelPositionalEncoding.position_encoding(offset=0, size=n_keys): rows for relative positions n_keys-1 ... -(n_keys-1)."""
        rel = torch.arange(n_keys - 1, -n_keys, -1, dtype=torch.float32).unsqueeze(1)
        div = torch.exp(torch.arange(0, self.d, 2, dtype=torch.float32) * -(math.log(10000.0) / self.d))
        pe = torch.zeros(2 * n_keys - 1, self.d)
        pe[:, 0::2], pe[:, 1::2] = torch.sin(rel * div), torch.cos(rel * div)
        return pe.to(ref)

    def _attention(self, a, x, pos, cache, causal):
        """RelPositionMultiHeadedAttention for batch 1: x [t1, d]; cache (k, v) [h, t0, dk] or None; returns (out [t1, d], new cache)."""
        h, t1 = self.heads, x.shape[0]
        split = lambda y: y.view(-1, h, self.d // h).transpose(0, 1)          # [h, t, dk]
        q, k, v = split(_linear(a.sub("linear_q."), x)), split(_linear(a.sub("linear_k."), x)), split(_linear(a.sub("linear_v."), x))
        if cache is not None:
            k, v = torch.cat([cache[0], k], 1), torch.cat([cache[1], v], 1)
        n_keys = k.shape[1]
        pe = split(F.linear(pos, a("linear_pos.weight")))                     # [h, 2 * n_keys - 1, dk]
        ac = torch.matmul(q + a("pos_bias_u").unsqueeze(1), k.transpose(1, 2))
        bd = torch.matmul(q + a("pos_bias_v").unsqueeze(1), pe.transpose(1, 2))
        # rel_shift (attention.py:232-247) as an index: query i (absolute position n_keys - t1 + i) against key j reads column t1 - 1 - i + j
        idx = (t1 - 1 - torch.arange(t1).unsqueeze(1) + torch.arange(n_keys).unsqueeze(0)).to(x.device)
        scores = (ac + bd.gather(2, idx.unsqueeze(0).expand(h, -1, -1))) / math.sqrt(self.d // h)
        if causal and t1 > 1:
            keep = torch.ones(t1, n_keys, dtype=torch.bool, device=x.device).tril(n_keys - t1)
            scores = scores.masked_fill(~keep, -float("inf"))
        o = torch.matmul(torch.softmax(scores, -1), v).transpose(0, 1).reshape(t1, self.d)
        return _linear(a.sub("linear_out."), o), (k, v)

    def _layer(self, i, x, pos, cache, causal):
        L = self.p.sub("encoders.%d." % i)
        n_att, n_ff = ("norm_mha.", "norm_ff.") if self.kind == "conformer" else ("norm1.", "norm2.")
        att, new_cache = self._attention(L.sub("self_attn."), _ln(L.sub(n_att), x, 1e-12), pos, cache, causal)
        x = x + att
        y = _linear(L.sub("feed_forward.w_1."), _ln(L.sub(n_ff), x, 1e-12))
        y = F.silu(y) if self.kind == "conformer" else F.relu(y)
        return x + _linear(L.sub("feed_forward.w_2."), y), new_cache

    def forward(self, xs):
        """xs [T, d_in] -> [T, d]   (BaseEncoder.forward for one unpadded sequence)."""
        x = self._embed(xs)
        pos = self._pos(x.shape[0], x)
        for i in range(self.n_layers):
            x, _ = self._layer(i, x, pos, None, self.causal)
        return _ln(self.p.sub("after_norm."), x, 1e-5)

    def forward_chunk(self, xs, caches):
        """BaseEncoder.forward_chunk with required_cache_size = -1 (all history kept): xs [t1, d_in], caches = per-layer (k, v) or None.
        A multi-row chunk is attended causally (the lower-triangular att_mask TransformerLM passes, llm/llm.py:208-210)."""
        x = self._embed(xs)
        t0 = 0 if caches is None else caches[0][0].shape[1]
        pos = self._pos(t0 + x.shape[0], x)
        new = []
        for i in range(self.n_layers):
            x, c = self._layer(i, x, pos, None if caches is None else caches[i], True)
            new.append(c)
        return _ln(self.p.sub("after_norm."), x, 1e-5), new


# ------------------------------------------------------------------------------------------------------------------------------------
# sampling (utils/common.py:138-167): same decisions and the same draws from the global torch RNG as the reference
# ------------------------------------------------------------------------------------------------------------------------------------
def ras_sampling(weighted_scores, decoded_tokens, sampling, top_p=0.8, top_k=25, win_size=10, tau_r=0.1):
    prob, order = weighted_scores.softmax(dim=0).sort(descending=True, stable=True)
    cum = torch.cumsum(prob, 0)                              # element i is taken while the running sum BEFORE it is < top_p, at most top_k
    n = int(min(top_k, 1 + int((cum[:-1] < top_p).sum().item()))) if prob.numel() > 1 else 1
    # (the reference accumulates in python floats of the same fp32 values in the same order: the prefix sums agree)
    top = int(order[:n][prob[:n].clone().multinomial(1, replacement=True)].item())
    recent = decoded_tokens[-win_size:] if win_size > 0 else []
    if sum(1 for t in recent if int(t) == top) >= win_size * tau_r:
        weighted_scores[top] = -float("inf")                 # in place, like the reference (:142)
        top = int(weighted_scores.softmax(dim=0).multinomial(1, replacement=True).item())
    return top


class TransformerLM:
    """cosyvoice.llm.llm.TransformerLM for inference.  `sd`: its state dict (text_embedding, text_encoder.*, text_encoder_affine_layer,
    llm_embedding, llm.*, llm_decoder, speech_embedding, spk_embed_affine_layer)."""

    def __init__(self, sd, text_heads=16, llm_heads=16, sampling=ras_sampling):
        self.sd = sd
        self.text_encoder = EspnetEncoder(sd, "text_encoder.", text_heads, "conformer", causal=True)    # static_chunk_size: 1 (cosyvoice.yaml)
        self.llm = EspnetEncoder(sd, "llm.", llm_heads, "transformer")
        self.speech_token_size = sd["llm_decoder.weight"].shape[0] - 1
        self.llm_input_size = sd["llm_embedding.weight"].shape[1]
        self.sos, self.task_id, self.eos_token = 0, 1, self.speech_token_size
        self.sampling = sampling

    @torch.inference_mode()
    def inference(self, text, text_len, prompt_text, prompt_text_len, prompt_speech_token, prompt_speech_token_len, embedding,
                  sampling=25, max_token_text_ratio=20, min_token_text_ratio=2, uuid=""):
        sd = self.sd
        dev = sd["llm_embedding.weight"].device
        ids = torch.cat([prompt_text, text], dim=1).reshape(-1).long().to(dev)
        embedding, prompt_speech_token = embedding.to(dev), prompt_speech_token.to(dev)
        n_text = int(text.shape[1])                          # (text_len + prompt_text_len) - prompt_text_len, llm.py:196-197
        x = self.text_encoder.forward(F.embedding(ids, sd["text_embedding.weight"]))
        x = _linear(_P(sd, "text_encoder_affine_layer."), x)
        rows = [sd["llm_embedding.weight"][self.sos].unsqueeze(0)]
        if embedding.shape[0] != 0:
            rows.append(_linear(_P(sd, "spk_embed_affine_layer."), F.normalize(embedding.to(x), dim=1)))
        rows += [x, sd["llm_embedding.weight"][self.task_id].unsqueeze(0)]
        if int(prompt_speech_token.shape[1]) != 0:
            rows.append(F.embedding(prompt_speech_token.reshape(-1).long(), sd["speech_embedding.weight"]))
        lm_input = torch.cat(rows, 0)
        min_len, max_len = int(n_text * min_token_text_ratio), int(n_text * max_token_text_ratio)
        out_tokens, caches = [], None
        for i in range(max_len):
            y, caches = self.llm.forward_chunk(lm_input, caches)
            logp = _linear(_P(sd, "llm_decoder."), y[-1]).log_softmax(dim=-1).cpu()    # the sampler draws from the host RNG (same tokens on every device)
            if i < min_len:
                logp[self.speech_token_size] = -float("inf")
            top = self.sampling(logp, out_tokens, sampling)
            if top == self.eos_token:
                break
            yield top
            out_tokens.append(top)
            lm_input = sd["speech_embedding.weight"][top].unsqueeze(0)


# ------------------------------------------------------------------------------------------------------------------------------------
# flow: MaskedDiffWithXvec (flow/flow.py:25-146)
# ------------------------------------------------------------------------------------------------------------------------------------
def _conv1d(p, x, stride=1, padding=0, dilation=1):
    return F.conv1d(x, _wn(p), p.get("bias"), stride, padding, dilation)


def _block1d(p, x):
    """Matcha Block1D: Conv1d(3) -> GroupNorm(8) -> Mish on [1, C, T]."""
    y = _conv1d(p.sub("block.0."), x, padding=1)
    return F.mish(F.group_norm(y, 8, p("block.1.weight"), p("block.1.bias")))


def _resnet1d(p, x, temb):
    h = _block1d(p.sub("block1."), x) + _linear(p.sub("mlp.1."), F.mish(temb)).unsqueeze(-1)
    return _block1d(p.sub("block2."), h) + _conv1d(p.sub("res_conv."), x)


def _transformer_block(p, x, heads):
    """Matcha BasicTransformerBlock (diffusers Attention without bias on q/k/v, GELU feed-forward) on [1, T, C], full attention."""
    y = _ln(p.sub("norm1."), x, 1e-5)
    t, inner = y.shape[1], p("attn1.to_q.weight").shape[0]
    split = lambda z: z.view(1, t, heads, inner // heads).transpose(1, 2)
    o = F.scaled_dot_product_attention(split(F.linear(y, p("attn1.to_q.weight"))), split(F.linear(y, p("attn1.to_k.weight"))),
                                       split(F.linear(y, p("attn1.to_v.weight"))))
    x = _linear(p.sub("attn1.to_out.0."), o.transpose(1, 2).reshape(1, t, inner)) + x
    y = _linear(p.sub("ff.net.2."), F.gelu(_linear(p.sub("ff.net.0.proj."), _ln(p.sub("norm3."), x, 1e-5))))
    return y + x


class ConditionalDecoder:
    """flow/decoder.py:88-291 with channels = [256, 256]: a real 1-D U-Net (stride-2 Downsample1D, ConvTranspose Upsample1D), each stage a
    ResnetBlock1D + n_blocks transformer blocks.  Masks are all ones for an unpadded batch-1 sequence and are therefore dropped."""

    def __init__(self, sd, prefix, heads=8):
        self.p, self.heads = _P(sd, prefix), heads
        self.n_down, self.n_mid, self.n_up = self.p.count("down_blocks."), self.p.count("mid_blocks."), self.p.count("up_blocks.")
        self.in_channels = self.p("time_mlp.linear_1.weight").shape[1]

    def _stage(self, p, x, temb):
        x = _resnet1d(p.sub("0."), x, temb).transpose(1, 2)
        for j in range(p.count("1.")):
            x = _transformer_block(p.sub("1.%d." % j), x, self.heads)
        return x.transpose(1, 2)

    def __call__(self, x, mask, mu, t, spks=None, cond=None, streaming=False):
        p = self.p
        half = self.in_channels // 2                           # SinusoidalPosEmb(in_channels), scale 1000
        freq = torch.exp(torch.arange(half, device=t.device).float() * -(math.log(10000) / (half - 1)))
        emb = 1000 * t.reshape(-1, 1) * freq.unsqueeze(0)
        temb = torch.cat([emb.sin(), emb.cos()], -1).to(t.dtype)
        temb = _linear(p.sub("time_mlp.linear_2."), F.silu(_linear(p.sub("time_mlp.linear_1."), temb)))
        x = torch.cat([x, mu], 1)
        if spks is not None:
            x = torch.cat([x, spks.unsqueeze(-1).expand(-1, -1, x.shape[-1])], 1)
        if cond is not None:
            x = torch.cat([x, cond], 1)
        hiddens = []
        for i in range(self.n_down):
            d = p.sub("down_blocks.%d." % i)
            x = self._stage(d, x, temb)
            hiddens.append(x)
            x = _conv1d(d.sub("2.conv."), x, stride=2, padding=1) if d.get("2.conv.weight") is not None else _conv1d(d.sub("2."), x, padding=1)
        for i in range(self.n_mid):
            x = self._stage(p.sub("mid_blocks.%d." % i), x, temb)
        for i in range(self.n_up):
            u = p.sub("up_blocks.%d." % i)
            skip = hiddens.pop()
            x = self._stage(u, torch.cat([x[:, :, :skip.shape[-1]], skip], 1), temb)
            if u.get("2.conv.weight") is not None:
                x = F.conv_transpose1d(x, u("2.conv.weight"), u("2.conv.bias"), stride=2, padding=1)
            else:
                x = _conv1d(u.sub("2."), x, padding=1)
        x = _block1d(p.sub("final_block."), x)
        return _conv1d(p.sub("final_proj."), x) * mask


class ConditionalCFM:
    """flow/flow_matching.py:21-124: Euler solver with classifier-free guidance and the prompt / overlap `flow_cache` of CosyVoice-300M."""

    def __init__(self, estimator, inference_cfg_rate=0.7, t_scheduler="cosine"):
        self.estimator, self.inference_cfg_rate, self.t_scheduler = estimator, inference_cfg_rate, t_scheduler

    @torch.inference_mode()
    def __call__(self, mu, mask, n_timesteps, temperature=1.0, spks=None, cond=None, prompt_len=0, cache=None):
        z = torch.randn(mu.shape, dtype=mu.dtype).to(mu.device) * temperature     # drawn on the host: the same stream on every device
        cache = (torch.zeros(1, mu.shape[1], 0, 2) if cache is None else cache).to(mu.device)
        n_cache = cache.shape[2]
        if n_cache != 0:                                      # keep the prompt and the overlap region on their previous trajectory (:53-56)
            z[:, :, :n_cache] = cache[:, :, :, 0]
            mu[:, :, :n_cache] = cache[:, :, :, 1]
        keep = lambda a: torch.cat([a[:, :, :prompt_len], a[:, :, -34:]], dim=2)
        new_cache = torch.stack([keep(z), keep(mu)], dim=-1)
        t_span = torch.linspace(0, 1, n_timesteps + 1, device=mu.device, dtype=mu.dtype)
        if self.t_scheduler == "cosine":
            t_span = 1 - torch.cos(t_span * 0.5 * torch.pi)
        return self.solve_euler(z, t_span, mu, mask, spks, cond), new_cache

    def solve_euler(self, x, t_span, mu, mask, spks, cond):
        t, dt = t_span[0].unsqueeze(0), t_span[1] - t_span[0]
        zeros = torch.zeros_like
        for step in range(1, len(t_span)):
            d = self.estimator(torch.cat([x, x]), torch.cat([mask, mask]), torch.cat([mu, zeros(mu)]), torch.cat([t, t]),
                               torch.cat([spks, zeros(spks)]), torch.cat([cond, zeros(cond)]))
            d = (1.0 + self.inference_cfg_rate) * d[:1] - self.inference_cfg_rate * d[1:]
            x = x + dt * d
            t = t + dt
            if step < len(t_span) - 1:
                dt = t_span[step + 1] - t
        return x.float()


class _ConditionalDecoderB2(ConditionalDecoder):
    """the estimator called on the CFG pair (batch 2): the blocks above are written for batch 1, so the two rows run one after the other"""

    def __call__(self, x, mask, mu, t, spks=None, cond=None, streaming=False):
        run = ConditionalDecoder.__call__
        return torch.cat([run(self, x[i:i + 1], mask[i:i + 1], mu[i:i + 1], t[i:i + 1], spks[i:i + 1], cond[i:i + 1]) for i in range(x.shape[0])])


class MaskedDiffWithXvec:
    def __init__(self, sd, enc_heads=8, est_heads=8, input_frame_rate=50, n_timesteps=10):
        self.sd, self.input_frame_rate, self.n_timesteps = sd, input_frame_rate, n_timesteps
        self.encoder = EspnetEncoder(sd, "encoder.", enc_heads, "conformer")
        self.decoder = ConditionalCFM(_ConditionalDecoderB2(sd, "decoder.estimator.", est_heads))
        self.output_size = sd["encoder_proj.weight"].shape[0]

    def _regulate(self, x):
        """InterpolateRegulator.model: (Conv1d(3) -> GroupNorm(1) -> Mish) x len(sampling_ratios), then Conv1d(1)."""
        p = _P(self.sd, "length_regulator.model.")
        last = max(int(k[len(p.prefix):].split(".")[0]) for k in self.sd if k.startswith(p.prefix))
        i = 0
        while i < last:
            x = F.mish(F.group_norm(_conv1d(p.sub("%d." % i), x, padding=1), 1, p("%d.weight" % (i + 1)), p("%d.bias" % (i + 1))))
            i += 3
        return _conv1d(p.sub("%d." % last), x)

    @torch.inference_mode()
    def inference(self, token, token_len, prompt_token, prompt_token_len, prompt_feat, prompt_feat_len, embedding, flow_cache):
        assert token.shape[0] == 1
        sd = self.sd
        dev = sd["input_embedding.weight"].device
        token, prompt_token, prompt_feat, embedding = token.to(dev), prompt_token.to(dev), prompt_feat.to(dev), embedding.to(dev)
        spk = _linear(_P(sd, "spk_embed_affine_layer."), F.normalize(embedding.float(), dim=1))
        n1, n2 = int(prompt_token.shape[1]), int(token.shape[1])
        ids = torch.cat([prompt_token.reshape(-1), token.reshape(-1)]).long().clamp(min=0)
        h = _linear(_P(sd, "encoder_proj."), self.encoder.forward(F.embedding(ids, sd["input_embedding.weight"])))
        mel_len1, mel_len2 = int(prompt_feat.shape[1]), int(n2 / self.input_frame_rate * 22050 / 256)
        # InterpolateRegulator.inference (length_regulator.py:52-70): prompt and head / middle / tail of the new tokens are stretched separately
        # so that the 20-token overlap of consecutive chunks always maps to the same mel frames
        interp = lambda a, size: F.interpolate(a.t().unsqueeze(0).contiguous(), size=size, mode="linear")
        x1, x2 = h[:n1], h[n1:]
        edge = int(20 / self.input_frame_rate * 22050 / 256)
        if n2 > 40:
            x2 = torch.cat([interp(x2[:20], edge), interp(x2[20:-20], mel_len2 - 2 * edge), interp(x2[-20:], edge)], dim=2)
        else:
            x2 = interp(x2, mel_len2)
        x = torch.cat([interp(x1, mel_len1), x2], dim=2) if n1 != 0 else x2
        mu = self._regulate(x)                                 # [1, 80, mel_len1 + mel_len2]
        conds = torch.zeros(1, self.output_size, mel_len1 + mel_len2, dtype=mu.dtype, device=mu.device)
        conds[:, :, :mel_len1] = prompt_feat.transpose(1, 2)
        mask = torch.ones(1, 1, mel_len1 + mel_len2, dtype=mu.dtype, device=mu.device)
        feat, flow_cache = self.decoder(mu=mu.contiguous(), mask=mask, spks=spk, cond=conds, n_timesteps=self.n_timesteps, prompt_len=mel_len1, cache=flow_cache)
        feat = feat[:, :, mel_len1:]
        assert feat.shape[2] == mel_len2
        return feat.float(), flow_cache


# ------------------------------------------------------------------------------------------------------------------------------------
# HiFTGenerator at 22.05 kHz (hifigan/generator.py:378-569, f0_predictor.py:23-59)
# ------------------------------------------------------------------------------------------------------------------------------------
class HiFTGenerator:
    def __init__(self, sd, sampling_rate=22050, upsample_rates=(8, 8), upsample_kernel_sizes=(16, 16), n_fft=16, hop_len=4,
                 resblock_kernel_sizes=(3, 7, 11), resblock_dilation_sizes=((1, 3, 5),) * 3, source_resblock_kernel_sizes=(7, 11),
                 source_resblock_dilation_sizes=((1, 3, 5),) * 2, nb_harmonics=8, nsf_alpha=0.1, nsf_sigma=0.003, nsf_voiced_threshold=10,
                 lrelu_slope=0.1, audio_limit=0.99):
        self.p = _P(sd)
        self.sampling_rate, self.ups, self.up_k, self.n_fft, self.hop = sampling_rate, tuple(upsample_rates), tuple(upsample_kernel_sizes), n_fft, hop_len
        self.res_k, self.res_d = tuple(resblock_kernel_sizes), tuple(tuple(d) for d in resblock_dilation_sizes)
        self.src_k, self.src_d = tuple(source_resblock_kernel_sizes), tuple(tuple(d) for d in source_resblock_dilation_sizes)
        self.harmonics, self.sine_amp, self.noise_std, self.voiced_thr = nb_harmonics, nsf_alpha, nsf_sigma, nsf_voiced_threshold
        self.lrelu, self.audio_limit = lrelu_slope, audio_limit
        self.scale = int(np.prod(self.ups)) * hop_len
        self.sinegen_type = 1 if sampling_rate == 22050 else 2
        n = torch.arange(n_fft, dtype=torch.float64)
        self.window = (0.5 - 0.5 * torch.cos(2 * math.pi * n / n_fft)).float()      # scipy get_window("hann", n_fft, fftbins=True)

    def f0_predictor(self, x):
        p = self.p.sub("f0_predictor.")
        for i in range(0, 10, 2):
            x = F.elu(_conv1d(p.sub("condnet.%d." % i), x, padding=1))
        return torch.abs(_linear(p.sub("classifier."), x.transpose(1, 2)).squeeze(-1))

    def _source(self, f0):
        """SourceModuleHnNSF over SineGen type 1 (generator.py:125-186, 318-375): f0 [1, L, 1] -> merged source [1, L, 1]."""
        assert self.sinegen_type == 1, "the 24 kHz generators (SineGen2) are served by cosyvoice_amd.hift"
        f0 = f0.transpose(1, 2)
        mult = torch.arange(1, self.harmonics + 2, dtype=f0.dtype, device=f0.device).view(1, -1, 1)
        theta = 2 * np.pi * (torch.cumsum(f0 * mult / self.sampling_rate, dim=-1) % 1)
        phase = -np.pi + torch.rand(1, self.harmonics + 1, 1) * (2 * np.pi)        # Uniform(-pi, pi).sample()
        phase[:, 0, :] = 0
        sine = self.sine_amp * torch.sin(theta + phase.to(f0.device))
        uv = (f0 > self.voiced_thr).float()
        sine = sine * uv + (uv * self.noise_std + (1 - uv) * self.sine_amp / 3) * torch.randn(sine.shape).to(sine.device)
        merged = torch.tanh(_linear(self.p.sub("m_source.l_linear."), sine.transpose(1, 2)))
        torch.randn(uv.shape)                                  # the reference draws (and discards) the noise branch here: keep the RNG in step
        return merged

    def _resblock(self, p, x, k, dils):
        snake = lambda a, y: y + (1.0 / (a.view(1, -1, 1) + 1e-9)) * torch.sin(y * a.view(1, -1, 1)) ** 2
        for j, d in enumerate(dils):
            y = _conv1d(p.sub("convs1.%d." % j), snake(p("activations1.%d.alpha" % j), x), padding=(k * d - d) // 2, dilation=d)
            x = _conv1d(p.sub("convs2.%d." % j), snake(p("activations2.%d.alpha" % j), y), padding=(k - 1) // 2) + x
        return x

    @torch.inference_mode()
    def decode(self, x, s):
        p = self.p
        spec = torch.stft(s.squeeze(1), self.n_fft, self.hop, self.n_fft, window=self.window.to(s.device), return_complex=True)
        s_stft = torch.cat([spec.real, spec.imag], dim=1)
        x = _conv1d(p.sub("conv_pre."), x, padding=3)
        down = [1] + list(self.ups[::-1][:-1])
        cum = np.cumprod(down)[::-1]
        for i, (u, k) in enumerate(zip(self.ups, self.up_k)):
            x = F.conv_transpose1d(F.leaky_relu(x, self.lrelu), _wn(p.sub("ups.%d." % i)), p.get("ups.%d.bias" % i), stride=u, padding=(k - u) // 2)
            if i == len(self.ups) - 1:
                x = F.pad(x, (1, 0), mode="reflect")
            r = int(cum[i])
            si = _conv1d(p.sub("source_downs.%d." % i), s_stft) if r == 1 else _conv1d(p.sub("source_downs.%d." % i), s_stft, stride=r, padding=r // 2)
            x = x + self._resblock(p.sub("source_resblocks.%d." % i), si, self.src_k[i], self.src_d[i])
            xs = None
            for j, (rk, rd) in enumerate(zip(self.res_k, self.res_d)):
                y = self._resblock(p.sub("resblocks.%d." % (i * len(self.res_k) + j)), x, rk, rd)
                xs = y if xs is None else xs + y
            x = xs / len(self.res_k)
        x = _conv1d(p.sub("conv_post."), F.leaky_relu(x), padding=3)
        half = self.n_fft // 2 + 1
        mag, ph = torch.clip(torch.exp(x[:, :half]), max=1e2), torch.sin(x[:, half:])
        wav = torch.istft(torch.complex(mag * torch.cos(ph), mag * torch.sin(ph)), self.n_fft, self.hop, self.n_fft, window=self.window.to(x.device))
        return torch.clamp(wav, -self.audio_limit, self.audio_limit)

    @torch.inference_mode()
    def inference(self, speech_feat, cache_source=torch.zeros(1, 1, 0)):
        speech_feat = speech_feat.to(self.p("conv_pre.bias").device)
        f0 = self.f0_predictor(speech_feat)
        s = self._source(F.interpolate(f0[:, None], scale_factor=float(self.scale), mode="nearest").transpose(1, 2)).transpose(1, 2)
        if cache_source.shape[2] != 0:
            s[:, :, :cache_source.shape[2]] = cache_source.to(s.device)
        return self.decode(speech_feat, s), s


# ------------------------------------------------------------------------------------------------------------------------------------
# CosyVoiceModel (cli/model.py:27-242)
# ------------------------------------------------------------------------------------------------------------------------------------
def fade_in_out(fade_in, fade_out, window):
    """utils/common.py:170-178."""
    n = window.shape[0] // 2
    out = fade_in.clone()
    w = torch.as_tensor(window, dtype=torch.float64, device=out.device)      # numpy float64 window: the reference's product is formed in float64
    out[..., :n] = (out[..., :n].double() * w[:n] + fade_out[..., -n:].to(out.device).double() * w[n:]).to(out.dtype)
    return out


class CosyVoiceModel:
    """The reference's CosyVoiceModel over the modules above: same `load`, `tts`, `token2wav`, `llm_job`, `vc_job`, same per-uuid dicts.
    Deliberate differences (as for CosyVoice2Model): the streaming loop waits on a condition variable instead of
    `time.sleep(0.1)`, and the per-request state is dropped in a `finally`."""

    def __init__(self, llm=None, flow=None, hift=None, fp16=False):
        assert not fp16, "configs[0] is fp32 eager"
        self.llm, self.flow, self.hift, self.fp16 = llm, flow, hift, fp16
        rate = flow.input_frame_rate if flow is not None else 50
        self.token_min_hop_len, self.token_max_hop_len, self.token_overlap_len = 2 * rate, 4 * rate, 20
        self.mel_overlap_len = int(self.token_overlap_len / rate * 22050 / 256)
        self.mel_window = np.hamming(2 * self.mel_overlap_len)
        self.mel_cache_len = 20
        self.source_cache_len = int(self.mel_cache_len * 256)
        self.speech_window = np.hamming(2 * self.source_cache_len)
        self.stream_scale_factor = 1
        self.lock = threading.Lock()
        self.tts_speech_token_dict, self.llm_end_dict, self.mel_overlap_dict, self.flow_cache_dict, self.hift_cache_dict = {}, {}, {}, {}, {}
        self._cond = {}
        self._llm_error = {}
        self.silent_tokens = []

    def load(self, llm_model, flow_model, hift_model, device="cpu", **kw):
        """cli/model.py:65-73: the three state-dict files of a CosyVoice-300M model directory.  `device`: where the weights (and the compute)
        live - "cpu" is BASELINE.json configs[0]; "cuda" runs the same torch ops through PyTorch-ROCm."""
        ld = lambda f: {k: v.float().to(device) for k, v in torch.load(f, map_location="cpu", weights_only=True).items()}
        self.llm = TransformerLM(ld(llm_model), **{k: kw[k] for k in ("text_heads", "llm_heads") if k in kw})
        self.flow = MaskedDiffWithXvec(ld(flow_model), **{k: kw[k] for k in ("enc_heads", "est_heads", "input_frame_rate") if k in kw})
        self.hift = HiFTGenerator({k.replace("generator.", ""): v for k, v in ld(hift_model).items()}, **kw.get("hift", {}))

    def llm_job(self, text, prompt_text, llm_prompt_speech_token, llm_embedding, uuid):
        cond = self._cond[uuid]
        t = lambda n: torch.tensor([n], dtype=torch.int32)
        silent, max_silent = 0, 5
        try:
            for i in self.llm.inference(text=text, text_len=t(text.shape[1]), prompt_text=prompt_text, prompt_text_len=t(prompt_text.shape[1]),
                                        prompt_speech_token=llm_prompt_speech_token, prompt_speech_token_len=t(llm_prompt_speech_token.shape[1]),
                                        embedding=llm_embedding, uuid=uuid):
                if i in self.silent_tokens:
                    silent += 1
                    if silent > max_silent:
                        continue
                else:
                    silent = 0
                with cond:
                    self.tts_speech_token_dict[uuid].append(i)
                    cond.notify_all()
        except BaseException as e:
            self._llm_error[uuid] = e
        finally:
            with cond:
                self.llm_end_dict[uuid] = True
                cond.notify_all()

    def vc_job(self, source_speech_token, uuid):
        with self._cond[uuid]:
            self.tts_speech_token_dict[uuid] = source_speech_token.flatten().tolist()
            self.llm_end_dict[uuid] = True
            self._cond[uuid].notify_all()

    @torch.inference_mode()
    def token2wav(self, token, prompt_token, prompt_feat, embedding, uuid, finalize=False, speed=1.0):
        """cli/model.py:135-173."""
        t = lambda n: torch.tensor([n], dtype=torch.int32)
        tts_mel, self.flow_cache_dict[uuid] = self.flow.inference(token=token.to(torch.int32), token_len=t(token.shape[1]), prompt_token=prompt_token,
                                                                  prompt_token_len=t(prompt_token.shape[1]), prompt_feat=prompt_feat,
                                                                  prompt_feat_len=t(prompt_feat.shape[1]), embedding=embedding,
                                                                  flow_cache=self.flow_cache_dict[uuid])
        if self.mel_overlap_dict[uuid].shape[2] != 0:
            tts_mel = fade_in_out(tts_mel, self.mel_overlap_dict[uuid], self.mel_window)
        cache = self.hift_cache_dict[uuid]
        if cache is not None:
            tts_mel = torch.concat([cache["mel"], tts_mel], dim=2)
            cache_source = cache["source"]
        else:
            cache_source = torch.zeros(1, 1, 0)
        if finalize is False:
            self.mel_overlap_dict[uuid] = tts_mel[:, :, -self.mel_overlap_len:]
            tts_mel = tts_mel[:, :, :-self.mel_overlap_len]
            tts_speech, tts_source = self.hift.inference(speech_feat=tts_mel, cache_source=cache_source)
            if cache is not None:
                tts_speech = fade_in_out(tts_speech, cache["speech"], self.speech_window)
            self.hift_cache_dict[uuid] = {"mel": tts_mel[:, :, -self.mel_cache_len:], "source": tts_source[:, :, -self.source_cache_len:],
                                          "speech": tts_speech[:, -self.source_cache_len:]}
            tts_speech = tts_speech[:, :-self.source_cache_len]
        else:
            if speed != 1.0:
                assert cache is None, "speed change only support non-stream inference mode"
                tts_mel = F.interpolate(tts_mel, size=int(tts_mel.shape[2] / speed), mode="linear")
            tts_speech, tts_source = self.hift.inference(speech_feat=tts_mel, cache_source=cache_source)
            if cache is not None:
                tts_speech = fade_in_out(tts_speech, cache["speech"], self.speech_window)
        return tts_speech

    def tts(self, text=torch.zeros(1, 0, dtype=torch.int32), flow_embedding=torch.zeros(0, 192), llm_embedding=torch.zeros(0, 192),
            prompt_text=torch.zeros(1, 0, dtype=torch.int32), llm_prompt_speech_token=torch.zeros(1, 0, dtype=torch.int32),
            flow_prompt_speech_token=torch.zeros(1, 0, dtype=torch.int32), prompt_speech_feat=torch.zeros(1, 0, 80),
            source_speech_token=torch.zeros(1, 0, dtype=torch.int32), stream=False, speed=1.0, **kwargs):
        """cli/model.py:175-242: generator of {'tts_speech': [1, S] fp32 cpu}."""
        this_uuid = str(uuid_mod.uuid1())
        with self.lock:
            self.tts_speech_token_dict[this_uuid], self.llm_end_dict[this_uuid] = [], False
            self.hift_cache_dict[this_uuid] = None
            self.mel_overlap_dict[this_uuid] = torch.zeros(1, 80, 0)
            self.flow_cache_dict[this_uuid] = torch.zeros(1, 80, 0, 2)
            self._cond[this_uuid] = threading.Condition()
        cond = self._cond[this_uuid]
        if source_speech_token.shape[1] == 0:
            p = threading.Thread(target=self.llm_job, args=(text, prompt_text, llm_prompt_speech_token, llm_embedding, this_uuid))
        else:
            p = threading.Thread(target=self.vc_job, args=(source_speech_token, this_uuid))
        p.start()
        voc = dict(prompt_token=flow_prompt_speech_token, prompt_feat=prompt_speech_feat, embedding=flow_embedding, uuid=this_uuid)
        try:
            if stream is True:
                hop = self.token_min_hop_len
                while True:
                    need = hop + self.token_overlap_len
                    with cond:
                        cond.wait_for(lambda: len(self.tts_speech_token_dict[this_uuid]) >= need or self.llm_end_dict[this_uuid])
                        have = len(self.tts_speech_token_dict[this_uuid])
                        toks = list(self.tts_speech_token_dict[this_uuid][:need])
                    if have >= need:
                        yield {"tts_speech": self.token2wav(token=torch.tensor(toks).unsqueeze(0), finalize=False, **voc).cpu()}
                        with cond:
                            self.tts_speech_token_dict[this_uuid] = self.tts_speech_token_dict[this_uuid][hop:]
                        hop = min(self.token_max_hop_len, int(hop * self.stream_scale_factor))
                    else:
                        break                                  # the LM has ended and fewer than hop + overlap tokens remain
                p.join()
                self._raise_llm_error(this_uuid)
                rest = torch.tensor(self.tts_speech_token_dict[this_uuid]).unsqueeze(0)
                yield {"tts_speech": self.token2wav(token=rest, finalize=True, **voc).cpu()}
            else:
                p.join()
                self._raise_llm_error(this_uuid)
                allt = torch.tensor(self.tts_speech_token_dict[this_uuid]).unsqueeze(0)
                yield {"tts_speech": self.token2wav(token=allt, finalize=True, speed=speed, **voc).cpu()}
        finally:
            p.join()
            with self.lock:
                for d in (self.tts_speech_token_dict, self.llm_end_dict, self.mel_overlap_dict, self.hift_cache_dict, self.flow_cache_dict, self._cond, self._llm_error):
                    d.pop(this_uuid, None)

    def _raise_llm_error(self, uuid):
        err = self._llm_error.pop(uuid, None)
        if err is not None:
            raise err
