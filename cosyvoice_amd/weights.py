"""Weight repacker: reference state dicts (llm.pt / flow.pt / hift.pt key names, SURVEY.md §5 'Checkpoint / resume')
-> the device layouts the HIP kernels stream.

  * Linear / Conv weights -> row-major [N][taps * Kp] with Kp = round_up(K, 32), zero padded (gemm_conv.h);
    LLM and flow matrices are stored as bf16 ("W16A32": weights bf16, activations / accumulation fp32),
    HiFT stays fp32 (the reference always runs HiFT in fp32, cli/model.py:312).
  * q/k/v projections are fused into one matrix, gate/up rows are interleaved, ConvTranspose1d is re-expressed in
    polyphase form, weight-norm (g * v / ||v||) is folded (the reference's remove_weight_norm is broken).
Everything returned is a dict name -> contiguous torch tensor on `device`; cosyvoice_amd/{llm,flow,hift}.py register
them with cv_*_set_tensor and keep them alive.
"""
import torch

from .ops import pack_weight, round_up


def _bf16(t, device):
    return t.to(device=device, dtype=torch.bfloat16).contiguous()


def _f32(t, device):
    return t.to(device=device, dtype=torch.float32).contiguous()


def pack_llm(sd, cfg, device):
    """sd: Qwen2LM state dict (cosyvoice/llm/llm.py:257-297 over transformers Qwen2ForCausalLM)."""
    out = {}
    for i in range(cfg.layers):
        p = "llm.model.model.layers.%d." % i
        q = "layers.%d." % i
        out[q + "ln1"] = _f32(sd[p + "input_layernorm.weight"], device)
        out[q + "wqkv"] = _bf16(torch.cat([sd[p + "self_attn.q_proj.weight"], sd[p + "self_attn.k_proj.weight"], sd[p + "self_attn.v_proj.weight"]], 0), device)
        out[q + "bqkv"] = _f32(torch.cat([sd[p + "self_attn.q_proj.bias"], sd[p + "self_attn.k_proj.bias"], sd[p + "self_attn.v_proj.bias"]], 0), device)
        out[q + "wo"] = _bf16(sd[p + "self_attn.o_proj.weight"], device)
        out[q + "ln2"] = _f32(sd[p + "post_attention_layernorm.weight"], device)
        gu = torch.stack([sd[p + "mlp.gate_proj.weight"], sd[p + "mlp.up_proj.weight"]], 1).reshape(2 * cfg.inter, cfg.hidden)
        out[q + "wgu"] = _bf16(gu, device)
        out[q + "wdown"] = _bf16(sd[p + "mlp.down_proj.weight"], device)
    out["norm"] = _f32(sd["llm.model.model.norm.weight"], device)
    out["head.w"] = _bf16(sd["llm_decoder.weight"], device)
    # CosyVoice3LM has no head bias (llm/llm.py:688): the fused head GEMV takes a zero vector
    bias = sd["llm_decoder.bias"] if "llm_decoder.bias" in sd else torch.zeros(sd["llm_decoder.weight"].shape[0])
    out["head.b"] = _f32(bias, device)
    out["embed.speech"] = _bf16(sd["speech_embedding.weight"], device)
    host_only = {"embed.text": _bf16(sd["llm.model.model.embed_tokens.weight"], device)}      # gathered by cv_gather_rows
    if "llm_embedding.weight" in sd:                    # Qwen2LM: sos / task_id rows; CosyVoice3LM keeps them in speech_embedding
        host_only["embed.llm"] = _bf16(sd["llm_embedding.weight"], device)
    return out, host_only


FP8_MAX = 448.0                        # largest finite OCP e4m3 value


def quantize_fp8_rows(w):
    """w [N, K] (any float dtype) -> (uint8 [N, K] of OCP e4m3 bit patterns, fp32 [N] scales): w ~= fp8 * scale[:, None], scale = rowwise absmax / 448,
    round to nearest even (torch.float8_e4m3fn - the format gfx950's v_mfma_*_fp8_fp8 reads)."""
    wf = w.float()
    scale = wf.abs().amax(dim=1).clamp_min(1e-30) / FP8_MAX
    q = (wf / scale[:, None]).clamp(-FP8_MAX, FP8_MAX).to(torch.float8_e4m3fn)
    return q.view(torch.uint8).contiguous(), scale.contiguous()


def quantize_llm_fp8(packed, cfg):
    """fp8 copies of the decode matrices of pack_llm()'s output ("<name>.f8" / "<name>.f8s"), for Qwen2LM(batch_fp8=True)."""
    out = {}
    names = ["head.w"] + ["layers.%d.%s" % (i, n) for i in range(cfg.layers) for n in ("wqkv", "wo", "wgu", "wdown")]
    for n in names:
        out[n + ".f8"], out[n + ".f8s"] = quantize_fp8_rows(packed[n])
    return out


def _lin(out, name, w, b, device, dtype):
    wp, _ = pack_weight(w.to(device).float(), dtype)
    out[name + ".w"] = wp
    if b is not None:
        out[name + ".b"] = _f32(b, device)


def _conv(out, name, w, b, device, dtype):
    """Conv1d weight [Cout, Cin, k] -> [Cout][k][Cin_p]."""
    wp, _ = pack_weight(w.to(device).float().permute(0, 2, 1).contiguous(), dtype)
    out[name + ".w"] = wp
    if b is not None:
        out[name + ".b"] = _f32(b, device)


def _mfma_fragments(w):
    """bf16 matrix [N][K] (N % 16 == 0, K % 32 == 0) -> [N/16][K/32][64 lanes][8]: the operand fragment of v_mfma_f32_16x16x32_bf16 for the 16-row
    tile nt and k-step ks - lane l holds row 16 nt + l % 16, columns 32 ks + 8 (l / 16) .. + 7 - as one contiguous 1 KB block."""
    N, K = w.shape
    return w.reshape(N // 16, 16, K // 32, 4, 8).permute(0, 2, 3, 1, 4).reshape(N // 16, K // 32, 64, 8)


def pack_flow_tail(w_out, w_ff1, w_ff2, w_qkv_next=None, waves=4):
    """The weight stream of flow_tail_kernel (csrc/flow_tail.h) for one transformer block: every matrix cut into MFMA fragments and laid out in the
    exact order each of the 4 waves consumes them - out-projection, FF1 (passes of <= 8 tiles), FF2, then Q, K and V of the NEXT block - k-step
    major, tile minor inside a pass, wave w owning tiles w, w + 4, w + 8, ...  Returns bf16 [4][fragments per wave][64][8]."""
    def wave_part(fr, w, lo, hi, per_pass):
        mine = fr[lo:hi][w::waves]                                 # [tiles of this wave][KS][64][8]
        parts = []
        for p0 in range(0, mine.shape[0], per_pass):
            parts.append(mine[p0:p0 + per_pass].permute(1, 0, 2, 3).reshape(-1, 64, 8))       # k-step major, tile minor
        return torch.cat(parts, 0)

    fo, f1, f2 = _mfma_fragments(w_out), _mfma_fragments(w_ff1), _mfma_fragments(w_ff2)
    fq = _mfma_fragments(w_qkv_next) if w_qkv_next is not None else None
    tc = f1.shape[0] // waves
    pc = 8 if tc > 8 else tc
    out = []
    for w in range(waves):
        parts = [wave_part(fo, w, 0, fo.shape[0], fo.shape[0] // waves), wave_part(f1, w, 0, f1.shape[0], pc), wave_part(f2, w, 0, f2.shape[0], f2.shape[0] // waves)]
        if fq is not None:
            ni = fq.shape[0] // 3
            for which in range(3):
                parts.append(wave_part(fq, w, which * ni, (which + 1) * ni, ni // waves))
        out.append(torch.cat(parts, 0))
    return torch.stack(out, 0).contiguous()


def _band_qkv_parts(w_qkv, C, waves):
    """The QKV GEMM of a block as band passes (flow_band.h phase F / flow_lnqkv_kernel): per wave, the fragments of its tiles for every pass of C output rows, k-step
    major.  With two tiles per wave the Q | K rows are paired so that a lane's 4 + 4 accumulator columns are adjacent in memory (see pack_flow_band)."""
    tiles = C // 16
    wq = w_qkv
    if tiles == 2 * waves:
        inner2 = 2 * (wq.shape[0] // 3)
        idx = torch.arange(wq.shape[0])
        i = torch.arange(16)
        for c in range(inner2 // C):
            for wv in range(waves):
                for t in range(2):
                    j = wv + waves * t
                    idx[c * C + 16 * j:c * C + 16 * j + 16] = c * C + 32 * wv + 8 * (i >> 2) + 4 * t + (i & 3)
        wq = wq[idx.to(wq.device)]
    fq = _mfma_fragments(wq)
    assert w_qkv.shape[0] % C == 0 and w_qkv.shape[1] == C
    return [[fq[c * tiles:(c + 1) * tiles][w::waves].permute(1, 0, 2, 3).reshape(-1, 64, 8) for c in range(w_qkv.shape[0] // C)] for w in range(waves)]


def pack_flow_band_qkv(w_qkv, waves):
    """The weight stream of flow_lnqkv_kernel (csrc/flow_band.h, round 6): LayerNorm + QKV of a stage's FIRST transformer block as one launch per row band - the
    QKV part of a `bandq` stream on its own.  w_qkv: [3 INNER][C] fused q | k | v rows.  Returns bf16 [waves][fragments per wave][64][8]."""
    return torch.stack([torch.cat(parts, 0) for parts in _band_qkv_parts(w_qkv, w_qkv.shape[1], waves)], 0).contiguous()


def pack_flow_band(w_out, w_ff1, w_ff2, waves, w_qkv_next=None):
    """The weight stream of flow_band_kernel (csrc/flow_band.h) for one transformer block: out-projection [C][INNER], FF1 [FF][C] and FF2 [C][FF] cut into MFMA
    fragments in the order each wave consumes them.  Wave w owns the 16-column tiles w, w + waves, ... of every C-wide output; a PASS is (its tiles) x (<= 8
    k-steps), k-step major, tile minor: the out-projection in passes of 8 k-steps, then per chunk j of C hidden columns FF1 (rows j C .. j C + C of w_ff1, all of
    K = C) followed by FF2 (all rows, k-steps of hidden columns j C .. j C + C).  With `w_qkv_next` ([3 INNER][C], the NEXT block's fused q | k | v rows): behind them
    its QKV GEMM in passes of C output rows, each laid out like an FF1 chunk (the HAS_QKV form of the kernel).  Returns bf16 [waves][fragments per wave][64][8]."""
    C = w_out.shape[0]
    fo, f1, f2 = _mfma_fragments(w_out), _mfma_fragments(w_ff1), _mfma_fragments(w_ff2)
    ka, kc, nch, tiles = fo.shape[1], C // 32, w_ff1.shape[0] // C, C // 16
    qparts = _band_qkv_parts(w_qkv_next, C, waves) if w_qkv_next is not None else None
    out = []
    for w in range(waves):
        parts = []
        mine = fo[w::waves]                                          # [tiles of this wave][ka][64][8]
        for p0 in range(0, ka, 8):
            parts.append(mine[:, p0:p0 + 8].permute(1, 0, 2, 3).reshape(-1, 64, 8))
        for j in range(nch):
            parts.append(f1[j * tiles:(j + 1) * tiles][w::waves].permute(1, 0, 2, 3).reshape(-1, 64, 8))
            parts.append(f2[w::waves][:, j * kc:(j + 1) * kc].permute(1, 0, 2, 3).reshape(-1, 64, 8))
        if qparts is not None:
            parts.extend(qparts[w])
        out.append(torch.cat(parts, 0))
    return torch.stack(out, 0).contiguous()


def pack_flow(sd, cfg, device, dtype=torch.bfloat16, experiments=False):
    """sd: CausalMaskedDiffWithXvec state dict (cosyvoice/flow/flow.py:150-186)."""
    out = {}
    out["input_embedding"] = _bf16(sd["input_embedding.weight"], device) if dtype == torch.bfloat16 else _f32(sd["input_embedding.weight"], device)
    _lin(out, "spk_affine", sd["spk_embed_affine_layer.weight"], sd["spk_embed_affine_layer.bias"], device, dtype)
    e = "encoder."
    for name in ("embed", "up_embed"):
        _lin(out, "enc.%s.lin" % name, sd[e + name + ".out.0.weight"], sd[e + name + ".out.0.bias"], device, dtype)
        out["enc.%s.ln.g" % name] = _f32(sd[e + name + ".out.1.weight"], device)
        out["enc.%s.ln.b" % name] = _f32(sd[e + name + ".out.1.bias"], device)
    out["enc.after_norm.g"] = _f32(sd[e + "after_norm.weight"], device)
    out["enc.after_norm.b"] = _f32(sd[e + "after_norm.bias"], device)
    _conv(out, "enc.pre.conv1", sd[e + "pre_lookahead_layer.conv1.weight"], sd[e + "pre_lookahead_layer.conv1.bias"], device, dtype)
    _conv(out, "enc.pre.conv2", sd[e + "pre_lookahead_layer.conv2.weight"], sd[e + "pre_lookahead_layer.conv2.bias"], device, dtype)
    _conv(out, "enc.up.conv", sd[e + "up_layer.conv.weight"], sd[e + "up_layer.conv.bias"], device, dtype)

    def conformer(src, dst):
        a = src + "self_attn."
        _lin(out, dst + "qkv", torch.cat([sd[a + "linear_q.weight"], sd[a + "linear_k.weight"], sd[a + "linear_v.weight"]], 0),
             torch.cat([sd[a + "linear_q.bias"], sd[a + "linear_k.bias"], sd[a + "linear_v.bias"]], 0), device, dtype)
        _lin(out, dst + "pos", sd[a + "linear_pos.weight"], None, device, dtype)
        _lin(out, dst + "out", sd[a + "linear_out.weight"], sd[a + "linear_out.bias"], device, dtype)
        out[dst + "bias_u"] = _f32(sd[a + "pos_bias_u"].reshape(-1), device)
        out[dst + "bias_v"] = _f32(sd[a + "pos_bias_v"].reshape(-1), device)
        _lin(out, dst + "ff1", sd[src + "feed_forward.w_1.weight"], sd[src + "feed_forward.w_1.bias"], device, dtype)
        _lin(out, dst + "ff2", sd[src + "feed_forward.w_2.weight"], sd[src + "feed_forward.w_2.bias"], device, dtype)
        for n in ("norm_mha", "norm_ff"):
            out[dst + n + ".g"] = _f32(sd[src + n + ".weight"], device)
            out[dst + n + ".b"] = _f32(sd[src + n + ".bias"], device)

    for i in range(cfg.enc_blocks):
        conformer(e + "encoders.%d." % i, "enc.layers.%d." % i)
    for i in range(cfg.up_blocks):
        conformer(e + "up_encoders.%d." % i, "enc.up_layers.%d." % i)
    _lin(out, "encoder_proj", sd["encoder_proj.weight"], sd["encoder_proj.bias"], device, dtype)

    s = "decoder.estimator."
    _lin(out, "est.time1", sd[s + "time_mlp.linear_1.weight"], sd[s + "time_mlp.linear_1.bias"], device, dtype)
    _lin(out, "est.time2", sd[s + "time_mlp.linear_2.weight"], sd[s + "time_mlp.linear_2.bias"], device, dtype)

    def resnet(src, dst):
        _lin(out, dst + "mlp", sd[src + "mlp.1.weight"], sd[src + "mlp.1.bias"], device, dtype)
        for b in ("block1", "block2"):
            _conv(out, dst + b + ".conv", sd[src + b + ".block.0.weight"], sd[src + b + ".block.0.bias"], device, dtype)
            out[dst + b + ".ln.g"] = _f32(sd[src + b + ".block.2.weight"], device)
            out[dst + b + ".ln.b"] = _f32(sd[src + b + ".block.2.bias"], device)
        _conv(out, dst + "res", sd[src + "res_conv.weight"], sd[src + "res_conv.bias"], device, dtype)

    def tblock(src, dst):
        out[dst + "norm1.g"] = _f32(sd[src + "norm1.weight"], device); out[dst + "norm1.b"] = _f32(sd[src + "norm1.bias"], device)
        _lin(out, dst + "qkv", torch.cat([sd[src + "attn1.to_q.weight"], sd[src + "attn1.to_k.weight"], sd[src + "attn1.to_v.weight"]], 0), None, device, dtype)
        _lin(out, dst + "out", sd[src + "attn1.to_out.0.weight"], sd[src + "attn1.to_out.0.bias"], device, dtype)
        out[dst + "norm3.g"] = _f32(sd[src + "norm3.weight"], device); out[dst + "norm3.b"] = _f32(sd[src + "norm3.bias"], device)
        _lin(out, dst + "ff1", sd[src + "ff.net.0.proj.weight"], sd[src + "ff.net.0.proj.bias"], device, dtype)
        _lin(out, dst + "ff2", sd[src + "ff.net.2.weight"], sd[src + "ff.net.2.bias"], device, dtype)

    stages = [(s + "down_blocks.0.", "est.stage.0.")]
    stages += [(s + "mid_blocks.%d." % i, "est.stage.%d." % (i + 1)) for i in range(cfg.est_mid)]
    stages += [(s + "up_blocks.0.", "est.stage.%d." % (cfg.est_mid + 1))]
    inner = cfg.est_heads * 64
    tail_ok = dtype == torch.bfloat16 and (cfg.est_ch, inner) in ((256, 512), (64, 64))      # the instantiations of flow_tail_kernel (csrc/flow.hip)
    for src, dst in stages:
        resnet(src + "0.", dst + "res.")
        for j in range(cfg.est_blocks):
            tblock(src + "1.%d." % j, dst + "tf.%d." % j)
        if tail_ok:                                                 # the fused row-local tail of every block (csrc/flow_tail.h): a second, fragment-ordered copy
            for j in range(cfg.est_blocks):
                q = dst + "tf.%d." % j
                nxt = out[dst + "tf.%d.qkv.w" % (j + 1)].reshape(3 * inner, cfg.est_ch) if j + 1 < cfg.est_blocks else None
                z = torch.zeros(cfg.est_ch, device=device)
                out[q + "tail_prm"] = torch.cat([out[q + "out.b"], out[q + "norm3.g"], out[q + "norm3.b"], out[q + "ff1.b"], out[q + "ff2.b"],
                                                 out[dst + "tf.%d.norm1.g" % (j + 1)] if nxt is not None else z, out[dst + "tf.%d.norm1.b" % (j + 1)] if nxt is not None else z]).contiguous()
                if experiments:                                     # the 16-row tail of round 3 (csrc/experiments/flow_tail.h): only a CV_BUILD_EXPERIMENTS library can run it
                    out[q + "tail"] = pack_flow_tail(out[q + "out.w"].reshape(cfg.est_ch, inner), out[q + "ff1.w"].reshape(4 * cfg.est_ch, cfg.est_ch),
                                                     out[q + "ff2.w"].reshape(cfg.est_ch, 4 * cfg.est_ch), nxt)
                # the 64-row band form for large passes (csrc/flow_band.h): 8 waves at the real width, 4 at the test width
                out[q + "band"] = pack_flow_band(out[q + "out.w"].reshape(cfg.est_ch, inner), out[q + "ff1.w"].reshape(4 * cfg.est_ch, cfg.est_ch),
                                                 out[q + "ff2.w"].reshape(cfg.est_ch, 4 * cfg.est_ch), 8 if cfg.est_ch == 256 else 4)
                if j == 0:                                          # LayerNorm + QKV of the stage's first block as one band launch (flow_lnqkv_kernel, round 6)
                    out[q + "lnqkv"] = pack_flow_band_qkv(out[q + "qkv.w"].reshape(3 * inner, cfg.est_ch), 8 if cfg.est_ch == 256 else 4)
                if nxt is not None:                                 # the same stream with the next block's QKV GEMM behind it (flow_band_kernel<.., HAS_QKV>)
                    out[q + "bandq"] = pack_flow_band(out[q + "out.w"].reshape(cfg.est_ch, inner), out[q + "ff1.w"].reshape(4 * cfg.est_ch, cfg.est_ch),
                                                      out[q + "ff2.w"].reshape(cfg.est_ch, 4 * cfg.est_ch), 8 if cfg.est_ch == 256 else 4, nxt)
    _conv(out, "est.down_conv", sd[s + "down_blocks.0.2.weight"], sd[s + "down_blocks.0.2.bias"], device, dtype)
    _conv(out, "est.up_conv", sd[s + "up_blocks.0.2.weight"], sd[s + "up_blocks.0.2.bias"], device, dtype)
    _conv(out, "est.final.conv", sd[s + "final_block.block.0.weight"], sd[s + "final_block.block.0.bias"], device, dtype)
    out["est.final.ln.g"] = _f32(sd[s + "final_block.block.2.weight"], device)
    out["est.final.ln.b"] = _f32(sd[s + "final_block.block.2.bias"], device)
    _conv(out, "est.final_proj", sd[s + "final_proj.weight"], sd[s + "final_proj.bias"], device, dtype)
    return out


def pack_unet1(sd, prefix, est_heads, device, dtype=torch.float32):
    """The ConditionalDecoder of CosyVoice-300M (flow/decoder.py:88-291; state-dict keys under `prefix`, e.g. "decoder.estimator.") for a flow handle with
    cfg.estimator == 2 (csrc/flow.hip::unet1_forward): stages in execution order (down..., mid..., up...), each with its ResnetBlock1D (GroupNorm scale / shift under the
    names the causal U-Net uses for its LayerNorms), its transformer blocks, and `post` = what follows them (Conv1d k 3 | Downsample1D as a Linear over 3 C-wide windows |
    Upsample1D in polyphase form).  dtype bf16 adds the band / QKV streams of the large-M kernels (C = 256 with 8 heads, or the 64-wide test shape)."""
    out, s = {}, prefix
    count = lambda stem: len({k[len(s + stem):].split(".")[0] for k in sd if k.startswith(s + stem)})
    n_down, n_mid, n_up = count("down_blocks."), count("mid_blocks."), count("up_blocks.")
    assert n_down == n_up and n_down >= 1
    _lin(out, "est.time1", sd[s + "time_mlp.linear_1.weight"], sd[s + "time_mlp.linear_1.bias"], device, dtype)
    _lin(out, "est.time2", sd[s + "time_mlp.linear_2.weight"], sd[s + "time_mlp.linear_2.bias"], device, dtype)
    stages = [(s + "down_blocks.%d." % i, "down") for i in range(n_down)] + [(s + "mid_blocks.%d." % i, "mid") for i in range(n_mid)] + \
             [(s + "up_blocks.%d." % i, "up") for i in range(n_up)]
    C = sd[s + "final_proj.weight"].shape[1]
    inner = est_heads * 64
    band_ok = dtype == torch.bfloat16 and (C, inner) in ((256, 512), (64, 64))
    for si, (src, kind) in enumerate(stages):
        dst = "est.stage.%d." % si
        r = src + "0."
        _lin(out, dst + "res.mlp", sd[r + "mlp.1.weight"], sd[r + "mlp.1.bias"], device, dtype)
        for b in ("block1", "block2"):                              # Block1D: Conv1d, GroupNorm(8), Mish
            _conv(out, dst + "res." + b + ".conv", sd[r + b + ".block.0.weight"], sd[r + b + ".block.0.bias"], device, dtype)
            out[dst + "res." + b + ".ln.g"] = _f32(sd[r + b + ".block.1.weight"], device)
            out[dst + "res." + b + ".ln.b"] = _f32(sd[r + b + ".block.1.bias"], device)
        _conv(out, dst + "res.res", sd[r + "res_conv.weight"], sd[r + "res_conv.bias"], device, dtype)
        n_blocks = len({k[len(src + "1."):].split(".")[0] for k in sd if k.startswith(src + "1.")})
        for j in range(n_blocks):
            t, q = src + "1.%d." % j, dst + "tf.%d." % j
            assert sd[t + "attn1.to_q.weight"].shape == (inner, C), "the estimator's attention heads are 64 wide"
            out[q + "norm1.g"] = _f32(sd[t + "norm1.weight"], device); out[q + "norm1.b"] = _f32(sd[t + "norm1.bias"], device)
            _lin(out, q + "qkv", torch.cat([sd[t + "attn1.to_q.weight"], sd[t + "attn1.to_k.weight"], sd[t + "attn1.to_v.weight"]], 0), None, device, dtype)
            _lin(out, q + "out", sd[t + "attn1.to_out.0.weight"], sd[t + "attn1.to_out.0.bias"], device, dtype)
            out[q + "norm3.g"] = _f32(sd[t + "norm3.weight"], device); out[q + "norm3.b"] = _f32(sd[t + "norm3.bias"], device)
            _lin(out, q + "ff1", sd[t + "ff.net.0.proj.weight"], sd[t + "ff.net.0.proj.bias"], device, dtype)
            _lin(out, q + "ff2", sd[t + "ff.net.2.weight"], sd[t + "ff.net.2.bias"], device, dtype)
        if band_ok:                                                 # the streams of flow_band_kernel / flow_lnqkv_kernel, as pack_flow() makes them for the causal U-Net
            waves = 8 if C == 256 else 4
            for j in range(n_blocks):
                q = dst + "tf.%d." % j
                nxt = out[dst + "tf.%d.qkv.w" % (j + 1)].reshape(3 * inner, C) if j + 1 < n_blocks else None
                z = torch.zeros(C, device=device)
                out[q + "tail_prm"] = torch.cat([out[q + "out.b"], out[q + "norm3.g"], out[q + "norm3.b"], out[q + "ff1.b"], out[q + "ff2.b"],
                                                 out[dst + "tf.%d.norm1.g" % (j + 1)] if nxt is not None else z, out[dst + "tf.%d.norm1.b" % (j + 1)] if nxt is not None else z]).contiguous()
                mats = (out[q + "out.w"].reshape(C, inner), out[q + "ff1.w"].reshape(4 * C, C), out[q + "ff2.w"].reshape(C, 4 * C))
                out[q + "band"] = pack_flow_band(*mats, waves)
                if j == 0:
                    out[q + "lnqkv"] = pack_flow_band_qkv(out[q + "qkv.w"].reshape(3 * inner, C), waves)
                if nxt is not None:
                    out[q + "bandq"] = pack_flow_band(*mats, waves, nxt)
        if kind == "mid":
            continue
        if (src + "2.conv.weight") in sd:                           # Downsample1D: Conv1d(k 3, stride 2, pad 1) / Upsample1D: ConvTranspose1d(k 4, stride 2, pad 1)
            w, b = sd[src + "2.conv.weight"].float(), sd[src + "2.conv.bias"].float()
            if kind == "down":
                _lin(out, dst + "post", w.permute(0, 2, 1).reshape(w.shape[0], -1), b, device, dtype)
            else:                                                   # [C_in, C_out, 4] -> [2 C_out][2 taps][C_in]: phase r, tap j holds kernel element r + 2 j
                cin, cout, k = w.shape
                assert k == 4
                wp = torch.zeros(2, cout, 2, cin)
                for ph in range(2):
                    for j in range(2):
                        wp[ph, :, j, :] = w[:, :, ph + 2 * j].t()
                wpk, _ = pack_weight(wp.reshape(2 * cout, 2, cin).to(device), dtype)
                out[dst + "post.w"] = wpk
                out[dst + "post.b"] = _f32(b.repeat(2), device)
        else:
            _conv(out, dst + "post", sd[src + "2.weight"], sd[src + "2.bias"], device, dtype)
    _conv(out, "est.final.conv", sd[s + "final_block.block.0.weight"], sd[s + "final_block.block.0.bias"], device, dtype)
    out["est.final.ln.g"] = _f32(sd[s + "final_block.block.1.weight"], device)
    out["est.final.ln.b"] = _f32(sd[s + "final_block.block.1.bias"], device)
    _conv(out, "est.final_proj", sd[s + "final_proj.weight"], sd[s + "final_proj.bias"], device, dtype)
    return out, dict(C=C, n_blocks=n_blocks, n_mid=n_mid, n_down=n_down)


def pack_flow_dit(sd, cfg, device, dtype=torch.bfloat16):
    """sd: CausalMaskedDiffWithDiT state dict (cosyvoice/flow/flow.py:284-318 + flow/DiT/dit.py:104-144).  Device layout: q / k / v fused; the
    in_proj columns permuted from the reference's cat order [x | cond | mu | spks] (dit.py:90-96) to the packed estimator input [x | mu | spks |
    cond]; the grouped position convs as [D rows][31 taps][D/16] (one 64-row panel per group); `1 +` of the adaLN scales folded into the bias
    of the modulation projections (modules.py:245,270)."""
    out = {}
    D, mel = cfg.est_ch, cfg.mel
    out["input_embedding"] = _bf16(sd["input_embedding.weight"], device) if dtype == torch.bfloat16 else _f32(sd["input_embedding.weight"], device)
    _lin(out, "spk_affine", sd["spk_embed_affine_layer.weight"], sd["spk_embed_affine_layer.bias"], device, dtype)
    _conv(out, "dit.pre.conv1", sd["pre_lookahead_layer.conv1.weight"], sd["pre_lookahead_layer.conv1.bias"], device, dtype)
    _conv(out, "dit.pre.conv2", sd["pre_lookahead_layer.conv2.weight"], sd["pre_lookahead_layer.conv2.bias"], device, dtype)
    e = "decoder.estimator."
    _lin(out, "dit.time1", sd[e + "time_embed.time_mlp.0.weight"], sd[e + "time_embed.time_mlp.0.bias"], device, dtype)
    _lin(out, "dit.time2", sd[e + "time_embed.time_mlp.2.weight"], sd[e + "time_embed.time_mlp.2.bias"], device, dtype)
    w = sd[e + "input_embed.proj.weight"]
    assert w.shape[1] == 4 * mel
    w = torch.cat([w[:, 0:mel], w[:, 2 * mel:3 * mel], w[:, 3 * mel:4 * mel], w[:, mel:2 * mel]], 1)
    _lin(out, "dit.in_proj", w, sd[e + "input_embed.proj.bias"], device, dtype)
    for c in ("conv1", "conv2"):
        _conv(out, "dit.pos.%s" % c, sd[e + "input_embed.conv_pos_embed.%s.0.weight" % c], sd[e + "input_embed.conv_pos_embed.%s.0.bias" % c], device, dtype)
    for i in range(cfg.est_blocks):
        p, q = e + "transformer_blocks.%d." % i, "dit.blk.%d." % i
        b = sd[p + "attn_norm.linear.bias"].clone().float()
        b[D:2 * D] += 1.0; b[4 * D:5 * D] += 1.0                     # (1 + scale_msa), (1 + scale_mlp)
        _lin(out, q + "mod", sd[p + "attn_norm.linear.weight"], b, device, dtype)
        _lin(out, q + "qkv", torch.cat([sd[p + "attn.to_q.weight"], sd[p + "attn.to_k.weight"], sd[p + "attn.to_v.weight"]], 0),
             torch.cat([sd[p + "attn.to_q.bias"], sd[p + "attn.to_k.bias"], sd[p + "attn.to_v.bias"]], 0), device, dtype)
        _lin(out, q + "out", sd[p + "attn.to_out.0.weight"], sd[p + "attn.to_out.0.bias"], device, dtype)
        _lin(out, q + "ff1", sd[p + "ff.ff.0.0.weight"], sd[p + "ff.ff.0.0.bias"], device, dtype)
        _lin(out, q + "ff2", sd[p + "ff.ff.2.weight"], sd[p + "ff.ff.2.bias"], device, dtype)
    b = sd[e + "norm_out.linear.bias"].clone().float()
    b[0:D] += 1.0                                                    # AdaLayerNormZero_Final: scale is the FIRST chunk (modules.py:268)
    _lin(out, "dit.final_mod", sd[e + "norm_out.linear.weight"], b, device, dtype)
    _lin(out, "dit.proj_out", sd[e + "proj_out.weight"], sd[e + "proj_out.bias"], device, dtype)
    return out


def fold_weight_norm(sd, p):
    g, v = sd[p + "parametrizations.weight.original0"].float(), sd[p + "parametrizations.weight.original1"].float()
    return g * v / v.reshape(v.shape[0], -1).norm(dim=1).reshape(-1, 1, 1)


def _conv_w(sd, p):
    """Conv / ConvTranspose weight under any of the three spellings a hift.pt can carry: the parametrization API
    (`parametrizations.weight.original0/1`, hifigan/generator.py:26-29), legacy torch.nn.utils.weight_norm (`weight_g` / `weight_v`, which the
    reference accepts through its ImportError fallback and torch's compat load hook) or a plain folded `weight`."""
    if (p + "parametrizations.weight.original0") in sd:
        return fold_weight_norm(sd, p)
    if (p + "weight_g") in sd:
        g, v = sd[p + "weight_g"].float(), sd[p + "weight_v"].float()
        return g * v / v.reshape(v.shape[0], -1).norm(dim=1).reshape(-1, 1, 1)
    return sd[p + "weight"].float()


def split3_planes(w):
    """fp32 matrix [N][K] -> bf16 [3 N][K], row 3 n + p = plane p of row n, with w = w1 + w2 + w3 EXACTLY (w1 = bf16(w), w2 = bf16(w - w1),
    w3 = w - w1 - w2: 3 x 8 mantissa bits cover the 24 of an fp32 number; every residual is exact in fp32).  Operand of the two-sided split GEMM
    (csrc/gemm_conv.h, WX3): fp32-accurate products on the bf16 matrix pipe."""
    w = w.float()
    w1 = w.to(torch.bfloat16)
    r1 = w - w1.float()
    w2 = r1.to(torch.bfloat16)
    r2 = r1 - w2.float()
    w3 = r2.to(torch.bfloat16)
    assert torch.equal(w3.float(), r2) or not torch.isfinite(w).all() or (r2 - w3.float()).abs().max() <= 2.0 ** -126, "split3_planes: residual not representable"
    return torch.stack([w1, w2, w3], 1).reshape(3 * w.shape[0], w.shape[1]).contiguous()


def pack_hift(sd, cfg, device):
    """sd: HiFTGenerator state dict (cosyvoice/hifigan/generator.py:378-476), `generator.` prefix already stripped
    (cli/model.py:70-71)."""
    out, f32 = {}, torch.float32
    for j in range(5):
        p = "f0_predictor.condnet.%d." % (2 * j)
        _conv(out, "f0.conv%d" % j, _conv_w(sd, p), sd[p + "bias"], device, f32)
    out["f0.cls.w"] = _f32(sd["f0_predictor.classifier.weight"].reshape(-1), device)
    out["f0.cls.b"] = _f32(sd["f0_predictor.classifier.bias"], device)
    out["source.w"] = _f32(sd["m_source.l_linear.weight"].reshape(-1), device)
    out["source.b"] = _f32(sd["m_source.l_linear.bias"], device)
    _conv(out, "conv_pre", _conv_w(sd, "conv_pre."), sd["conv_pre.bias"], device, f32)
    for i, (u, k) in enumerate(zip(cfg.ups, cfg.up_k)):
        if cfg.causal:                                       # CausalConv1dUpsample (transformer/convolution.py:226-259): a stride-1 Conv1d
            _conv(out, "ups.%d" % i, _conv_w(sd, "ups.%d." % i), sd["ups.%d.bias" % i], device, f32)
            continue
        w = _conv_w(sd, "ups.%d." % i)                       # ConvTranspose1d weight [Cin, Cout, k]
        cin, cout, _ = w.shape
        q = (k + u - 1) // u
        wp = torch.zeros(u, cout, q, cin)
        for r in range(u):
            for qq in range(q):
                if r + u * qq < k:
                    wp[r, :, qq, :] = w[:, :, r + u * qq].t()
        packed, _ = pack_weight(wp.reshape(u * cout, q, cin).to(device), f32)
        out["ups.%d.w" % i] = packed
        out["ups.%d.b" % i] = _f32(sd["ups.%d.bias" % i].repeat(u), device)
    import numpy as np
    rates = np.cumprod([1] + cfg.ups[::-1][:-1])[::-1]
    for i, r in enumerate(rates):
        w = sd["source_downs.%d.weight" % i].float()          # [C, 18, k]  -> one im2col row of k*18
        c, cin, k = w.shape
        packed, _ = pack_weight(w.permute(0, 2, 1).reshape(c, 1, k * cin).to(device), f32)
        out["source_downs.%d.w" % i] = packed
        out["source_downs.%d.b" % i] = _f32(sd["source_downs.%d.bias" % i], device)

    def resblock(src, dst):
        for j in range(len(cfg.res_d)):
            for grp, act in (("convs1", "activations1"), ("convs2", "activations2")):
                _conv(out, "%s%s.%d" % (dst, grp, j), _conv_w(sd, "%s%s.%d." % (src, grp, j)), sd["%s%s.%d.bias" % (src, grp, j)], device, f32)
                a = sd["%s%s.%d.alpha" % (src, act, j)].float()
                ap = torch.ones(round_up(a.numel(), 32))
                ap[: a.numel()] = a
                out["%s%s.%d.alpha" % (dst, grp, j)] = _f32(ap, device)

    for i in range(len(cfg.ups)):
        resblock("source_resblocks.%d." % i, "source_resblocks.%d." % i)
        for j in range(len(cfg.res_k)):
            n = i * len(cfg.res_k) + j
            resblock("resblocks.%d." % n, "resblocks.%d." % n)
    _conv(out, "conv_post", _conv_w(sd, "conv_post."), sd["conv_post.bias"], device, f32)
    # the convolutions that carry the FLOPs (ResBlocks, upsamplers, conv_pre / conv_post) also get their three bf16 planes: csrc/gemm_conv.h WX3
    for name in [k[:-2] for k in out if k.endswith(".w") and (k.startswith(("resblocks.", "source_resblocks.", "ups.", "conv_pre", "conv_post")))]:      # (the f0 predictor stays on the fp32 chain: its output is integrated into a phase)
        out[name + ".w3"] = split3_planes(out[name + ".w"])
    return out
