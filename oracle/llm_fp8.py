"""Oracle mirror of the opt-in fp8 batched decode (TEST INFRASTRUCTURE - see oracle/__init__.py).  PARITY UNPINNED BY DEFINITION: the reference has no
fp8 path ("fp8 has no reference - tolerance vs bf16/fp32 oracle only").  This file restates what the product's fp8 mode is
DEFINED to compute, so that the kernel can be held to it exactly:

  weights      OCP e4m3, one fp32 scale per output row:  sw[n] = max_k |W[n][k]| / 448,  Wq = e4m3(W / sw)           (round to nearest even)
  activations  per sequence and per K range of a workgroup, after the RMSNorm gamma:  xg = x * gamma,  sx = max |xg| / 448,  q = e4m3(xg * (1 / sx))
  product      exact fp8 x fp8 products, fp32 accumulation:  y[n] = ((sum_k Wq[n][k] q[k]) * (sx * rstd)) * sw[n]   (+ bias, + residual)
  down         the K = inter contraction is cut into `ksplit` ranges, each with its own sx; the ranges are summed in order, then the residual

Prefill stays fp32 (the product prefills on the exact-fp32 MFMA in every mode); only decode steps (one new position) use the fp8 linears."""
import math

import torch
import torch.nn.functional as F

from . import llm as OL

FP8_MAX = 448.0


def quant_rows(w):
    wf = w.float()
    s = wf.abs().amax(dim=1).clamp_min(1e-30) / FP8_MAX
    return (wf / s[:, None]).clamp(-FP8_MAX, FP8_MAX).to(torch.float8_e4m3fn).float(), s


def quant_act(xg):
    amax = xg.abs().amax()
    sx = amax / FP8_MAX if float(amax) > 0 else torch.tensor(1.0)
    inv = 1.0 / sx
    return (xg * inv).to(torch.float8_e4m3fn).float(), sx


def down_ksplit(inter):
    tiles = inter // 64
    for ks in (8, 4, 2):
        if tiles % ks == 0 and tiles // ks >= 2 and (tiles // ks + 3) // 4 <= 5:
            return ks
    return 1


class Fp8Linear:
    def __init__(self, w, bias=None):
        self.wq, self.sw = quant_rows(w)
        self.bias = bias

    def __call__(self, x, gamma=None, eps=0.0, ksplit=1):
        """x [K] fp32 -> [N]"""
        K = x.shape[0]
        sc_extra = torch.rsqrt(x.pow(2).sum() / K + eps) if gamma is not None else None
        xg = x * gamma if gamma is not None else x
        y = None
        step = K // ksplit
        for s in range(ksplit):
            q, sx = quant_act(xg[s * step:(s + 1) * step])
            acc = self.wq[:, s * step:(s + 1) * step] @ q
            part = (acc * (sx * sc_extra if sc_extra is not None else sx)) * self.sw
            y = part if y is None else y + part
        return y if self.bias is None else y + self.bias


class Qwen2OracleFp8(OL.Qwen2Oracle):
    """Prefill: the fp32 oracle.  Decode step (one row): every Linear of the backbone through Fp8Linear; `head(h)` is the fp8 head on the pre-norm
    hidden state (the product fuses the final RMSNorm into the head GEMM)."""

    def __init__(self, sd, cfg):
        super().__init__(sd, cfg)
        self.lin = {}
        for i in range(cfg.layers):
            p = "llm.model.model.layers.%d." % i
            qkv_w = torch.cat([sd[p + "self_attn.q_proj.weight"], sd[p + "self_attn.k_proj.weight"], sd[p + "self_attn.v_proj.weight"]], 0)
            qkv_b = torch.cat([sd[p + "self_attn.q_proj.bias"], sd[p + "self_attn.k_proj.bias"], sd[p + "self_attn.v_proj.bias"]], 0)
            self.lin[i] = dict(qkv=Fp8Linear(qkv_w, qkv_b), o=Fp8Linear(sd[p + "self_attn.o_proj.weight"]), gate=Fp8Linear(sd[p + "mlp.gate_proj.weight"]),
                               up=Fp8Linear(sd[p + "mlp.up_proj.weight"]), down=Fp8Linear(sd[p + "mlp.down_proj.weight"]))
        self.head_lin = Fp8Linear(sd["llm_decoder.weight"], sd.get("llm_decoder.bias"))
        self.ks = down_ksplit(cfg.inter)

    def step(self, x_row):
        """x_row [hidden]: one decode step on the fp8 linears; returns the pre-norm hidden state [hidden]."""
        c, sd = self.cfg, self.sd
        pos = torch.tensor([self.pos])
        cos, sin = OL.rope_cos_sin(pos, c.head_dim, c.rope_theta)
        h = x_row.float()
        for i in range(c.layers):
            p = "llm.model.model.layers.%d." % i
            L = self.lin[i]
            qkv = L["qkv"](h, sd[p + "input_layernorm.weight"], c.rms_eps)
            nq, nk = c.heads * c.head_dim, c.kv_heads * c.head_dim
            q = qkv[:nq].view(1, c.heads, c.head_dim); k = qkv[nq:nq + nk].view(1, c.kv_heads, c.head_dim); v = qkv[nq + nk:].view(1, c.kv_heads, c.head_dim)
            q = q * cos[:, None, :] + OL.rotate_half(q) * sin[:, None, :]
            k = k * cos[:, None, :] + OL.rotate_half(k) * sin[:, None, :]
            self.k[i] = torch.cat([self.k[i], k], 0); self.v[i] = torch.cat([self.v[i], v], 0)
            g = c.heads // c.kv_heads
            kk, vv = self.k[i].repeat_interleave(g, dim=1), self.v[i].repeat_interleave(g, dim=1)
            s = torch.einsum("qhd,khd->hqk", q, kk) / math.sqrt(c.head_dim)
            a = torch.einsum("hqk,khd->qhd", torch.softmax(s, -1), vv).reshape(c.heads * c.head_dim)
            h = h + L["o"](a)
            gamma = sd[p + "post_attention_layernorm.weight"]
            gt, up = L["gate"](h, gamma, c.rms_eps), L["up"](h, gamma, c.rms_eps)
            act = (gt / (1.0 + torch.exp(-gt))) * up
            h = h + L["down"](act, ksplit=self.ks)
        self.pos += 1
        return h

    def head(self, h_prenorm):
        return self.head_lin(h_prenorm, self.sd["llm.model.model.norm.weight"], self.cfg.rms_eps)


def inference(sd, cfg, text, prompt_text, prompt_speech_token, max_token_text_ratio=20, min_token_text_ratio=2, trace=None):
    """Greedy decode as the product's fp8 batched path runs it: fp32 prefill, then head -> sample -> embed -> backbone step, all on the fp8 linears."""
    lm_input = OL.build_lm_input(sd, cfg, text, prompt_text, prompt_speech_token)
    n_text = text.shape[1]
    min_len, max_len = int(n_text * min_token_text_ratio), int(n_text * max_token_text_ratio)
    stop = [cfg.speech_token_size + i for i in range(cfg.n_special)]
    m = Qwen2OracleFp8(sd, cfg)
    # pre-norm hidden state of the last prompt position: run the fp32 prefill on all but the last row, then the last row alone
    h = _prefill_prenorm(m, lm_input)
    out = []
    for i in range(max_len):
        logits = m.head(h)
        logp = logits.log_softmax(dim=-1)
        if trace is not None:
            trace.setdefault("logp", []).append(logp.clone())
        if i < min_len:
            logp[cfg.speech_token_size] = -float("inf")
        top = int(logp.argmax().item())
        if top in stop:
            break
        out.append(top)
        h = m.step(sd["speech_embedding.weight"][top])
    return out


def _prefill_prenorm(m, lm_input):
    """fp32 forward of the prompt that also returns the hidden state BEFORE the final RMSNorm (Qwen2Oracle.forward returns it after)."""
    c, sd = m.cfg, m.sd
    q_len = lm_input.shape[0]
    pos = torch.arange(0, q_len)
    cos, sin = OL.rope_cos_sin(pos, c.head_dim, c.rope_theta)
    h = lm_input.float()
    for i in range(c.layers):
        p = "llm.model.model.layers.%d." % i
        r = h
        n = OL.rms_norm(h, sd[p + "input_layernorm.weight"], c.rms_eps)
        q = F.linear(n, sd[p + "self_attn.q_proj.weight"], sd[p + "self_attn.q_proj.bias"]).view(q_len, c.heads, c.head_dim)
        k = F.linear(n, sd[p + "self_attn.k_proj.weight"], sd[p + "self_attn.k_proj.bias"]).view(q_len, c.kv_heads, c.head_dim)
        v = F.linear(n, sd[p + "self_attn.v_proj.weight"], sd[p + "self_attn.v_proj.bias"]).view(q_len, c.kv_heads, c.head_dim)
        q = q * cos[:, None, :] + OL.rotate_half(q) * sin[:, None, :]
        k = k * cos[:, None, :] + OL.rotate_half(k) * sin[:, None, :]
        m.k[i], m.v[i] = k, v
        g = c.heads // c.kv_heads
        kk, vv = k.repeat_interleave(g, dim=1), v.repeat_interleave(g, dim=1)
        s = torch.einsum("qhd,khd->hqk", q, kk) / math.sqrt(c.head_dim)
        causal = torch.arange(q_len)[None, :] <= pos[:, None]
        s = s.masked_fill(~causal[None], float("-inf"))
        a = torch.einsum("hqk,khd->qhd", torch.softmax(s, -1), vv).reshape(q_len, c.heads * c.head_dim)
        h = r + F.linear(a, sd[p + "self_attn.o_proj.weight"])
        r = h
        n = OL.rms_norm(h, sd[p + "post_attention_layernorm.weight"], c.rms_eps)
        h = r + F.linear(F.silu(F.linear(n, sd[p + "mlp.gate_proj.weight"])) * F.linear(n, sd[p + "mlp.up_proj.weight"]), sd[p + "mlp.down_proj.weight"])
    m.pos = q_len
    return h[-1]
