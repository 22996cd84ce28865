"""Oracle: Qwen2LM speech-token language model (TEST INFRASTRUCTURE — see oracle/__init__.py).

Restates  cosyvoice/llm/llm.py:226-254 (Qwen2Encoder), :458-549 (Qwen2LM.inference / inference_wrapper),
:150-160 (sampling_ids)  and the decoder-only transformer they call: transformers.Qwen2ForCausalLM
(third-party; reference pin transformers==4.51.3, requirements.txt:38).  Published Qwen2 algorithm: pre-norm
RMSNorm(eps) -> q/k/v Linear (with bias) -> rotate-half RoPE(theta) -> GQA causal SDPA (scale 1/sqrt(64)) ->
o_proj (no bias) -> residual -> RMSNorm -> SiLU-gated MLP -> residual; final RMSNorm.

Decode semantics follow the INTENDED behaviour of forward_one_step (plain causal attention over the whole KV cache);
the reference's `[1,1]` decode mask is mis-handled by transformers 5.x (SURVEY.md §0 "oracle trap"), so the golden
generator calls HF with attention_mask=None for q_len == 1.
"""
import math

import torch
import torch.nn.functional as F

from .sampling import ras_sampling


def rms_norm(x, w, eps):
    v = x.float()
    return w * (v * torch.rsqrt(v.pow(2).mean(-1, keepdim=True) + eps))


def rope_cos_sin(positions, head_dim, theta):
    inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
    freqs = positions.float()[:, None] * inv_freq[None, :]
    emb = torch.cat([freqs, freqs], dim=-1)
    return emb.cos(), emb.sin()


def rotate_half(x):
    x1, x2 = x[..., : x.shape[-1] // 2], x[..., x.shape[-1] // 2:]
    return torch.cat([-x2, x1], dim=-1)


class Qwen2Oracle:
    """Functional Qwen2 backbone over a reference-named state dict, with an explicit fp32 KV cache."""

    def __init__(self, sd, cfg):
        self.sd, self.cfg = sd, cfg
        self.reset()

    def reset(self):
        self.k = [None] * self.cfg.layers
        self.v = [None] * self.cfg.layers
        self.pos = 0

    def forward(self, x, return_all=False):
        """x [q, hidden] new positions appended to the cache (llm/llm.py:242-254 forward_one_step). Returns the final-norm
        hidden states [q, hidden] (== outs.hidden_states[-1])."""
        c, sd = self.cfg, self.sd
        q_len = x.shape[0]
        pos = torch.arange(self.pos, self.pos + q_len)
        cos, sin = rope_cos_sin(pos, c.head_dim, c.rope_theta)
        h = x.float()
        for i in range(c.layers):
            p = "llm.model.model.layers.%d." % i
            r = h
            n = rms_norm(h, sd[p + "input_layernorm.weight"], c.rms_eps)
            q = F.linear(n, sd[p + "self_attn.q_proj.weight"], sd[p + "self_attn.q_proj.bias"]).view(q_len, c.heads, c.head_dim)
            k = F.linear(n, sd[p + "self_attn.k_proj.weight"], sd[p + "self_attn.k_proj.bias"]).view(q_len, c.kv_heads, c.head_dim)
            v = F.linear(n, sd[p + "self_attn.v_proj.weight"], sd[p + "self_attn.v_proj.bias"]).view(q_len, c.kv_heads, c.head_dim)
            q = q * cos[:, None, :] + rotate_half(q) * sin[:, None, :]
            k = k * cos[:, None, :] + rotate_half(k) * sin[:, None, :]
            self.k[i] = k if self.k[i] is None else torch.cat([self.k[i], k], 0)
            self.v[i] = v if self.v[i] is None else torch.cat([self.v[i], v], 0)
            L = self.k[i].shape[0]
            g = c.heads // c.kv_heads
            kk = self.k[i].repeat_interleave(g, dim=1)
            vv = self.v[i].repeat_interleave(g, dim=1)
            s = torch.einsum("qhd,khd->hqk", q, kk) / math.sqrt(c.head_dim)
            causal = torch.arange(L)[None, :] <= (pos[:, None])
            s = s.masked_fill(~causal[None], float("-inf"))
            a = torch.einsum("hqk,khd->qhd", torch.softmax(s, -1), vv).reshape(q_len, c.heads * c.head_dim)
            h = r + F.linear(a, sd[p + "self_attn.o_proj.weight"])
            r = h
            n = rms_norm(h, sd[p + "post_attention_layernorm.weight"], c.rms_eps)
            m = F.silu(F.linear(n, sd[p + "mlp.gate_proj.weight"])) * F.linear(n, sd[p + "mlp.up_proj.weight"])
            h = r + F.linear(m, sd[p + "mlp.down_proj.weight"])
        self.pos += q_len
        return rms_norm(h, sd["llm.model.model.norm.weight"], c.rms_eps)


def build_lm_input(sd, cfg, text, prompt_text, prompt_speech_token):
    """[sos | embed_tokens(prompt_text ++ text) | task_id | speech_embedding(prompt)]   (llm/llm.py:472-494)."""
    text = torch.cat([prompt_text, text], dim=1).long()
    text_emb = sd["llm.model.model.embed_tokens.weight"][text[0]]
    if cfg.cv3:                                              # CosyVoice3LM: rows sts+0 / sts+2 of speech_embedding (llm/llm.py:483-485, 677-681)
        sos = sd["speech_embedding.weight"][cfg.speech_token_size:cfg.speech_token_size + 1]
        task = sd["speech_embedding.weight"][cfg.speech_token_size + 2:cfg.speech_token_size + 3]
    else:
        sos = sd["llm_embedding.weight"][0:1]
        task = sd["llm_embedding.weight"][1:2]
    sp = sd["speech_embedding.weight"][prompt_speech_token[0].long()] if prompt_speech_token.shape[1] else torch.zeros(0, cfg.hidden)
    return torch.cat([sos, text_emb, task, sp], dim=0)


def greedy_sampling(weighted_scores, decoded_tokens, sampling):
    """The yaml-injectable greedy sampler north_star parity is defined on (SURVEY.md §5 'Config / flags')."""
    return int(weighted_scores.argmax().item())


def inference(sd, cfg, text, prompt_text, prompt_speech_token, sampling_fn=greedy_sampling, sampling=25,
              max_token_text_ratio=20, min_token_text_ratio=2, trace=None):
    """Qwen2LM.inference + inference_wrapper non-vLLM branch (llm/llm.py:458-502, 535-549). Returns the token list.
    `trace`, if a dict, receives per-step logp tensors (for teacher-forced comparisons)."""
    lm_input = build_lm_input(sd, cfg, text, prompt_text, prompt_speech_token)
    text_len = text.shape[1]
    min_len = int(text_len * min_token_text_ratio)           # (text_len + prompt_len - prompt_len) * ratio  (:497-498)
    max_len = int(text_len * max_token_text_ratio)
    stop = [cfg.speech_token_size + i for i in range(cfg.n_special)]      # 3 for Qwen2LM (:297), 200 for CosyVoice3LM (:703)
    if cfg.cv3:
        assert bool((torch.cat([prompt_text, text], dim=1) == cfg.endofprompt_id).any()), "<|endofprompt|> not detected"       # :478-480
    model = Qwen2Oracle(sd, cfg)
    out = []
    x = lm_input
    for i in range(max_len):
        y = model.forward(x)
        logp = F.linear(y[-1], sd["llm_decoder.weight"], sd.get("llm_decoder.bias")).log_softmax(dim=-1)
        if trace is not None:
            trace.setdefault("logp", []).append(logp.clone())
        if i < min_len:                                      # sampling_ids ignore_eos (:150-160)
            logp[cfg.speech_token_size] = -float("inf")
        top = sampling_fn(logp, out, sampling)
        if top in stop:
            break
        out.append(top)
        x = sd["speech_embedding.weight"][top].reshape(1, -1)
    return out


def inference_bistream(sd, cfg, text_chunks, prompt_text, prompt_speech_token, sampling_fn=greedy_sampling, sampling=25, mix_ratio=(5, 15)):
    """Qwen2LM / CosyVoice3LM .inference_bistream (llm/llm.py:551-661) statement by statement: `text_chunks` is an iterable of [1, n]
    id tensors (the generator of the reference).  Returns the yielded tokens.  Kept quirks: `lm_input` stays visible after it was
    forwarded and is forwarded again in front of the final text (:642); ignore_eos masks index `speech_token_size` (:150-160)."""
    sts = cfg.speech_token_size
    if cfg.cv3:
        sos_id, eos, task_id, fill = sts + 0, sts + 1, sts + 2, sts + 3
        sos, task = sd["speech_embedding.weight"][sos_id:sos_id + 1], sd["speech_embedding.weight"][task_id:task_id + 1]
    else:
        eos, fill = sts, sts + 2
        sos, task = sd["llm_embedding.weight"][0:1], sd["llm_embedding.weight"][1:2]
    emb_t = sd["llm.model.model.embed_tokens.weight"]
    emb_s = sd["speech_embedding.weight"]
    prompt_sp = emb_s[prompt_speech_token[0].long()] if prompt_speech_token.shape[1] else torch.zeros(0, cfg.hidden)
    lm_input = sos
    out_tokens, yielded = [], []
    model = Qwen2Oracle(sd, cfg)
    prompt_ids = prompt_text[0].long()
    if cfg.cv3:                                                                              # :583-588
        pl = prompt_ids.tolist()
        assert cfg.endofprompt_id in pl, "<|endofprompt|> not detected in CosyVoice3 prompt_text"
        k = pl.index(cfg.endofprompt_id)
        lm_input = torch.cat([lm_input, emb_t[prompt_ids[: k + 1]]], 0)
        prompt_ids = prompt_ids[k + 1:]
    text_cache = emb_t[prompt_ids]
    n_prompt = prompt_speech_token.shape[1]
    next_fill_index = (int(n_prompt / mix_ratio[1]) + 1) * mix_ratio[1] - n_prompt

    def head(y):
        return F.linear(y[-1], sd["llm_decoder.weight"], sd.get("llm_decoder.bias")).log_softmax(dim=-1)

    for this_text in text_chunks:
        text_cache = torch.cat([text_cache, emb_t[this_text[0].long()]], 0)
        while prompt_sp.shape[0] != 0:
            if text_cache.shape[0] >= mix_ratio[0]:
                lm_input = torch.cat([lm_input, text_cache[: mix_ratio[0]], prompt_sp[: mix_ratio[1]]], 0)
                text_cache, prompt_sp = text_cache[mix_ratio[0]:], prompt_sp[mix_ratio[1]:]
            else:
                break
        if prompt_sp.shape[0] == 0:
            if (len(out_tokens) != 0 and out_tokens[-1] == fill) or (len(out_tokens) == 0 and lm_input.shape[0] == 1):
                if text_cache.shape[0] >= mix_ratio[0]:
                    lm_input_text = text_cache[: mix_ratio[0]]
                    if len(out_tokens) != 0 and out_tokens[-1] == fill:
                        lm_input = lm_input_text
                    else:
                        lm_input = torch.cat([lm_input, lm_input_text], 0)
                    text_cache = text_cache[mix_ratio[0]:]
                else:
                    continue
            while True:
                logp = head(model.forward(lm_input))
                if next_fill_index != -1 and len(out_tokens) == next_fill_index:
                    top = fill
                    next_fill_index += mix_ratio[1] + 1
                else:
                    logp[sts] = -float("inf")                                                # sampling_ids(ignore_eos=True)
                    top = sampling_fn(logp, out_tokens, sampling)
                if top == fill:
                    next_fill_index = len(out_tokens) + mix_ratio[1] + 1
                out_tokens.append(top)
                if top >= sts:
                    if top == fill:
                        break
                    raise ValueError("should not get token {}".format(top))
                yielded.append(top)
                lm_input = emb_s[top].reshape(1, -1)
    lm_input = torch.cat([lm_input, text_cache, task], 0)                                    # :642 (lm_input may already have been forwarded)
    while True:
        logp = head(model.forward(lm_input))
        top = sampling_fn(logp, out_tokens, sampling)                                        # ignore_eos=False
        out_tokens.append(top)
        if top >= sts:
            if top == eos:
                break
            raise ValueError("should not get token {}".format(top))
        yielded.append(top)
        lm_input = emb_s[top].reshape(1, -1)
    return yielded
