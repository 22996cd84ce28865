"""Oracle: Fun-CosyVoice3 flow = CausalMaskedDiffWithDiT around the DiT estimator (TEST INFRASTRUCTURE - see oracle/__init__.py).

Restates cosyvoice/flow/flow.py:369-414 (CausalMaskedDiffWithDiT.inference), cosyvoice/transformer/upsample_encoder.py:66-103 (PreLookaheadLayer),
cosyvoice/flow/DiT/dit.py:76-98,145-176 (InputEmbedding, DiT.forward) and cosyvoice/flow/DiT/modules.py:71-83 (SinusPositionEmbedding), :115-144
(CausalConvPositionEmbedding), :230-282 (AdaLayerNormZero / _Final), :289-301 (FeedForward, GELU tanh), :363-407 (AttnProcessor incl. the rotary
quirk: rotation before the head split with 64-dim freqs, i.e. head 0 only), :500-530 (DiTBlock), :606-616 (TimestepEmbedding).  The CFM solver is the
shared one (oracle.flow.solve_euler restated from flow_matching.py:71-124,203-227).  Third-party boundary: x_transformers 2.11.24
(RotaryEmbedding.forward_from_seq_len, apply_rotary_pos_emb) - restated from the published source, PARITY UNPINNED there.
Pinned by tests/golden/dit_tiny.npz, generated from the REAL reference classes (tests/golden/make_golden.py::golden_dit)."""
import math

import torch
import torch.nn.functional as F

from . import flow as OF

_r = OF._r          # bf16 operand rounding when oracle.flow.bf16_act is active (mirrors the device's bf16 mode), identity otherwise


def _lin(x, w, b=None):
    return F.linear(_r(x), _r(w), b)


def rotary_freqs(T, dim=64, base=10000.0):
    inv = 1.0 / (base ** (torch.arange(0, dim, 2).float() / dim))
    f = torch.arange(T).float()[:, None] * inv[None, :]
    return torch.stack((f, f), -1).reshape(T, dim)                       # interleaved duplicates


def apply_rotary_head0(t, freqs):
    """apply_rotary_pos_emb on the un-split [B, T, H*64] projection with 64-dim freqs: channels 0..63 (= head 0) turn, the rest pass."""
    left, right = t[..., :64], t[..., 64:]
    x = left.reshape(*left.shape[:-1], 32, 2)
    rot = torch.stack((-x[..., 1], x[..., 0]), -1).reshape(left.shape)
    return torch.cat((left * freqs.cos() + rot * freqs.sin(), right), -1)


def time_embedding(sd, p, t):
    half = 128
    emb = torch.exp(torch.arange(half).float() * -(math.log(10000) / (half - 1)))
    e = 1000 * t[:, None] * emb[None, :]
    e = torch.cat((e.sin(), e.cos()), -1)
    h = F.silu(_lin(e, sd[p + "time_mlp.0.weight"], sd[p + "time_mlp.0.bias"]))
    return _lin(h, sd[p + "time_mlp.2.weight"], sd[p + "time_mlp.2.bias"])


def causal_conv_pos(sd, p, x):
    """x [B, T, D] -> two causal grouped convs (k 31, 16 groups) + Mish (modules.py:115-144)."""
    y = x.transpose(1, 2)
    for c in ("conv1", "conv2"):
        w = sd[p + c + ".0.weight"]
        y = F.mish(F.conv1d(_r(F.pad(y, (w.shape[2] - 1, 0))), _r(w), sd[p + c + ".0.bias"], groups=16))
    return y.transpose(1, 2)


def chunk_mask(T, chunk):
    i = torch.arange(T)
    return (i[None, :] // chunk) <= (i[:, None] // chunk) if chunk > 0 else torch.ones(T, T, dtype=torch.bool)


def estimator(sd, cfg, x, mask, mu, t, spks, cond, streaming=False):
    """DiT.forward (dit.py:145-176): x, mu, cond [B, 80, T], t [B], spks [B, 80] -> [B, 80, T]."""
    e = "decoder.estimator."
    D, H = cfg.est_ch, cfg.est_heads
    x, mu, cond = x.transpose(1, 2), mu.transpose(1, 2), cond.transpose(1, 2)
    B, T = x.shape[0], x.shape[1]
    temb = time_embedding(sd, e + "time_embed.", t)
    h = _lin(torch.cat([x, cond, mu, spks[:, None, :].expand(B, T, -1)], -1), sd[e + "input_embed.proj.weight"], sd[e + "input_embed.proj.bias"])
    h = causal_conv_pos(sd, e + "input_embed.conv_pos_embed.", h) + h
    freqs = rotary_freqs(T)
    allow = chunk_mask(T, 2 * cfg.chunk if streaming else 0)
    ln = lambda v: F.layer_norm(v, (D,), eps=1e-6)
    for i in range(cfg.est_blocks):
        p = e + "transformer_blocks.%d." % i
        m6 = _lin(F.silu(temb), sd[p + "attn_norm.linear.weight"], sd[p + "attn_norm.linear.bias"])
        shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = m6.chunk(6, dim=1)
        n = ln(h) * (1 + scale_msa[:, None]) + shift_msa[:, None]
        q = apply_rotary_head0(_lin(n, sd[p + "attn.to_q.weight"], sd[p + "attn.to_q.bias"]), freqs)
        k = apply_rotary_head0(_lin(n, sd[p + "attn.to_k.weight"], sd[p + "attn.to_k.bias"]), freqs)
        v = _lin(n, sd[p + "attn.to_v.weight"], sd[p + "attn.to_v.bias"])
        q, k, v = (z.view(B, T, H, 64).transpose(1, 2) for z in (q, k, v))
        s = torch.matmul(_r(q), _r(k).transpose(-1, -2)) / 8.0
        a = torch.softmax(s.masked_fill(~allow, float("-inf")), -1)
        o = OF._attn_pv(a, v).transpose(1, 2).reshape(B, T, H * 64)
        o = _lin(o, sd[p + "attn.to_out.0.weight"], sd[p + "attn.to_out.0.bias"])
        h = h + gate_msa[:, None] * o
        n = ln(h) * (1 + scale_mlp[:, None]) + shift_mlp[:, None]
        f = F.gelu(_lin(n, sd[p + "ff.ff.0.0.weight"], sd[p + "ff.ff.0.0.bias"]), approximate="tanh")
        h = h + gate_mlp[:, None] * _lin(f, sd[p + "ff.ff.2.weight"], sd[p + "ff.ff.2.bias"])
    m2 = _lin(F.silu(temb), sd[e + "norm_out.linear.weight"], sd[e + "norm_out.linear.bias"])
    scale, shift = m2.chunk(2, dim=1)
    h = ln(h) * (1 + scale)[:, None, :] + shift[:, None, :]
    return _lin(h, sd[e + "proj_out.weight"], sd[e + "proj_out.bias"]).transpose(1, 2)


def pre_lookahead(sd, cfg, x, context=None):
    """PreLookaheadLayer.forward (upsample_encoder.py:82-103): x [1, n, 80], context [1, 3, 80] or None."""
    la = cfg.pre_lookahead
    y = x.transpose(1, 2)
    y = torch.cat([y, context.transpose(1, 2)], 2) if context is not None and context.numel() else F.pad(y, (0, la))
    y = F.leaky_relu(F.conv1d(_r(y), _r(sd["pre_lookahead_layer.conv1.weight"]), sd["pre_lookahead_layer.conv1.bias"]))
    y = F.conv1d(_r(F.pad(y, (2, 0))), _r(sd["pre_lookahead_layer.conv2.weight"]), sd["pre_lookahead_layer.conv2.bias"])
    return y.transpose(1, 2) + x


def solve_euler(sd, cfg, mu, mask, spks, cond, n_timesteps=10, streaming=False):
    """flow_matching.py:71-124 with the DiT as the estimator (same control flow as oracle.flow.solve_euler)."""
    T = mu.shape[2]
    x = OF.cfm_noise(T)
    t_span = 1 - torch.cos(torch.linspace(0, 1, n_timesteps + 1) * 0.5 * torch.pi)
    t, dt = t_span[0].unsqueeze(0), t_span[1] - t_span[0]
    for step in range(1, len(t_span)):
        d = estimator(sd, cfg, torch.cat([x, x], 0), torch.cat([mask, mask], 0), torch.cat([mu, torch.zeros_like(mu)], 0), torch.cat([t, t], 0),
                      torch.cat([spks, torch.zeros_like(spks)], 0), torch.cat([cond, torch.zeros_like(cond)], 0), streaming)
        x = x + dt * ((1.0 + cfg.cfg_rate) * d[:1] - cfg.cfg_rate * d[1:])
        t = t + dt
        if step < len(t_span) - 1:
            dt = t_span[step + 1] - t
    return x.float()


def inference(sd, cfg, token, prompt_token, prompt_feat, embedding, streaming=False, finalize=True, n_timesteps=10):
    """CausalMaskedDiffWithDiT.inference (flow/flow.py:369-414) -> mel [1, 80, 2 * n_new]."""
    emb = F.linear(_r(F.normalize(embedding, dim=1)), _r(sd["spk_embed_affine_layer.weight"]), sd["spk_embed_affine_layer.bias"])
    tok = torch.cat([prompt_token, token], 1).long().clamp(min=0)
    x = sd["input_embedding.weight"][tok[0]].unsqueeze(0)
    la = cfg.pre_lookahead
    h = pre_lookahead(sd, cfg, x) if finalize else pre_lookahead(sd, cfg, x[:, :-la], x[:, -la:])
    h = h.repeat_interleave(2, dim=1)
    mel_len1, T = prompt_feat.shape[1], h.shape[1]
    conds = torch.zeros(1, T, cfg.mel)
    conds[:, :mel_len1] = prompt_feat
    feat = solve_euler(sd, cfg, h.transpose(1, 2).contiguous(), torch.ones(1, 1, T), emb, conds.transpose(1, 2), n_timesteps, streaming)
    return feat[:, :, mel_len1:]
