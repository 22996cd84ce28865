"""Restatement of the Matcha-TTS classes the reference imports (third_party/Matcha-TTS is an EMPTY submodule in the
reference snapshot, pin unrecoverable: SURVEY.md §2 row 19, Appendix B).  Used ONLY to let the real
cosyvoice/flow/{decoder,flow_matching}.py run when generating golden vectors.

Sources restated (published algorithms):
  matcha/models/components/decoder.py      SinusoidalPosEmb, Block1D, ResnetBlock1D, Downsample1D, TimestepEmbedding, Upsample1D
  matcha/models/components/transformer.py  BasicTransformerBlock (self-attention only) + FeedForward/GELU
  diffusers==0.29.0 models/attention_processor.py  Attention (to_q/to_k/to_v no bias, to_out.0 with bias, SDPA, float mask added)
  matcha/models/components/flow_matching.py BASECFM.__init__
State-dict key names follow upstream so that real flow.pt checkpoints load with strict=True.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


class SinusoidalPosEmb(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.dim = dim
        assert dim % 2 == 0

    def forward(self, x, scale=1000):
        if x.ndim < 1:
            x = x.unsqueeze(0)
        half_dim = self.dim // 2
        emb = math.log(10000) / (half_dim - 1)
        emb = torch.exp(torch.arange(half_dim, device=x.device).float() * -emb)
        emb = scale * x.unsqueeze(1) * emb.unsqueeze(0)
        return torch.cat((emb.sin(), emb.cos()), dim=-1)


class Block1D(nn.Module):
    def __init__(self, dim, dim_out, groups=8):
        super().__init__()
        self.block = nn.Sequential(nn.Conv1d(dim, dim_out, 3, padding=1), nn.GroupNorm(groups, dim_out), nn.Mish())

    def forward(self, x, mask):
        return self.block(x * mask) * mask


class ResnetBlock1D(nn.Module):
    def __init__(self, dim, dim_out, time_emb_dim, groups=8):
        super().__init__()
        self.mlp = nn.Sequential(nn.Mish(), nn.Linear(time_emb_dim, dim_out))
        self.block1 = Block1D(dim, dim_out, groups=groups)
        self.block2 = Block1D(dim_out, dim_out, groups=groups)
        self.res_conv = nn.Conv1d(dim, dim_out, 1)

    def forward(self, x, mask, time_emb):
        h = self.block1(x, mask)
        h += self.mlp(time_emb).unsqueeze(-1)
        h = self.block2(h, mask)
        return h + self.res_conv(x * mask)


class Downsample1D(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.conv = nn.Conv1d(dim, dim, 3, 2, 1)

    def forward(self, x):
        return self.conv(x)


class Upsample1D(nn.Module):
    def __init__(self, channels, use_conv=False, use_conv_transpose=True, out_channels=None, name="conv"):
        super().__init__()
        self.channels = channels
        self.out_channels = out_channels or channels
        self.conv = nn.ConvTranspose1d(channels, self.out_channels, 4, 2, 1)

    def forward(self, inputs):
        return self.conv(inputs)


class TimestepEmbedding(nn.Module):
    def __init__(self, in_channels, time_embed_dim, act_fn="silu", out_dim=None, post_act_fn=None, cond_proj_dim=None):
        super().__init__()
        assert act_fn == "silu"
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(time_embed_dim, out_dim or time_embed_dim)

    def forward(self, sample, condition=None):
        return self.linear_2(self.act(self.linear_1(sample)))


class _Attention(nn.Module):
    """diffusers Attention(query_dim, heads, dim_head, bias=False) self-attention with AttnProcessor2_0."""

    def __init__(self, query_dim, heads, dim_head, dropout=0.0):
        super().__init__()
        inner = heads * dim_head
        self.heads = heads
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(query_dim, inner, bias=False)
        self.to_v = nn.Linear(query_dim, inner, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim), nn.Dropout(dropout)])

    def forward(self, x, attention_mask=None):
        b, t, _ = x.shape
        q = self.to_q(x).view(b, t, self.heads, -1).transpose(1, 2)
        k = self.to_k(x).view(b, t, self.heads, -1).transpose(1, 2)
        v = self.to_v(x).view(b, t, self.heads, -1).transpose(1, 2)
        m = None if attention_mask is None else attention_mask.unsqueeze(1)      # [B,1,T,T] float bias
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=m)
        o = o.transpose(1, 2).reshape(b, t, -1)
        return self.to_out[1](self.to_out[0](o))


class _GELU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out)

    def forward(self, x):
        return F.gelu(self.proj(x))


class _FeedForward(nn.Module):
    def __init__(self, dim, mult=4, dropout=0.0, activation_fn="gelu"):
        super().__init__()
        assert activation_fn == "gelu"
        self.net = nn.ModuleList([_GELU(dim, dim * mult), nn.Dropout(dropout), nn.Linear(dim * mult, dim)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, num_attention_heads, attention_head_dim, dropout=0.0, activation_fn="gelu", **kw):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn1 = _Attention(dim, num_attention_heads, attention_head_dim, dropout)
        self.norm3 = nn.LayerNorm(dim)
        self.ff = _FeedForward(dim, dropout=dropout, activation_fn=activation_fn)

    def forward(self, hidden_states, attention_mask=None, timestep=None, **kw):
        hidden_states = self.attn1(self.norm1(hidden_states), attention_mask=attention_mask) + hidden_states
        return self.ff(self.norm3(hidden_states)) + hidden_states


class BASECFM(nn.Module):
    def __init__(self, n_feats, cfm_params, n_spks=1, spk_emb_dim=128):
        super().__init__()
        self.n_feats = n_feats
        self.n_spks = n_spks
        self.spk_emb_dim = spk_emb_dim
        self.solver = cfm_params.solver
        self.sigma_min = getattr(cfm_params, "sigma_min", 1e-4) if hasattr(cfm_params, "sigma_min") else 1e-4
        self.estimator = None
