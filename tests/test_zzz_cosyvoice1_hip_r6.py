"""CosyVoice-300M's U-Net estimator inside the flow handle (csrc/flow.hip cfg.estimator == 2), the model's fp16 mode (bf16 matrices
under the decode step, bf16-mode estimator) and the decode GEMV layouts - against the launch-per-operator classes, the torch-eager port and the REAL classes' goldens.
(A file of its own: pytest-xdist hands out whole files, and tests/test_zzz_cosyvoice1_hip.py is minutes of emulator work already.)"""
import torch

from cosyvoice_amd import cosyvoice1 as C1
from cosyvoice_amd import cosyvoice1_hip as CK
from cosyvoice_amd import synthetic as W
from cv1k_common import CFG, build_flow, gold, greedy, t


def test_estimator_handle_matches_the_operator_sequence_and_the_reference(lib):
    """The U-Net of CosyVoice-300M inside one library handle (csrc/flow.hip cfg.estimator == 2, round 6) against (1) the launch-per-operator class on the same
    ConditionalDecoder.forward call (flow/decoder.py:204-291; an odd frame count: the up-sampled stream is one row longer than the skip it is cut to) and (2) the real
    MaskedDiffWithXvec's golden through both requests of the flow-cache test; the default of MaskedDiffWithXvec IS the handle (build_flow above)."""
    K = CK.Kernels(lib)
    sd = W.make_cv1_flow(CFG)
    ops = CK.ConditionalDecoder(sd, "decoder.estimator.", CFG.est_heads, K)
    hd = CK.EstimatorHandle(sd, "decoder.estimator.", CFG.est_heads, lib, "fp32")
    gen = torch.Generator().manual_seed(1)
    for T in (67, 40):
        x, mu, cond = (torch.randn(2, 80, T, generator=gen) for _ in range(3))
        spks, tt = torch.randn(2, 80, generator=gen), torch.tensor([0.3, 0.3])
        got = hd.forward(x, torch.ones(2, 1, T), mu, tt, spks, cond).cpu()
        tall, offs = ops.prepare([0.3])
        h = K.put(torch.cat([x, mu, spks.unsqueeze(2).expand(-1, -1, T), cond], 1).transpose(1, 2))
        want = ops(h, T, tall[0], offs)[0].view(2, T, 80).transpose(1, 2).cpu()
        torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)
    assert isinstance(build_flow(lib).estimator, CK.EstimatorHandle)
    g = gold("cv1k_flow")
    flow = build_flow(lib, "operators")
    torch.manual_seed(90)
    feat, _ = flow.inference(token=g["token_a"], token_len=t(50), prompt_token=g["prompt_token"], prompt_token_len=t(12), prompt_feat=g["prompt_feat"],
                             prompt_feat_len=t(25), embedding=g["embedding"], flow_cache=torch.zeros(1, 80, 0, 2))
    torch.testing.assert_close(feat.cpu(), g["feat_a"], rtol=1e-3, atol=1e-3)


def test_estimator_handle_bf16_mode_and_solve_graph(lib):
    """precision "bf16" (the analogue of the reference's fp16=True for this model): the fused transformer-block kernels of the CosyVoice2 estimator under the U-Net of
    CosyVoice-300M - within the bf16 tolerance of the real class's mel; the solve is captured at the second sighting of a shape and the replay returns the eager bits."""
    g = gold("cv1k_flow")
    flow = build_flow(lib, "handle", "bf16")
    flow.estimator.set_option("use_graph", 1)                    # (off by default for this model: measured slower than issuing the launches)
    kw = dict(token=g["token_a"], token_len=t(50), prompt_token=g["prompt_token"], prompt_token_len=t(12), prompt_feat=g["prompt_feat"], prompt_feat_len=t(25),
              embedding=g["embedding"], flow_cache=torch.zeros(1, 80, 0, 2))
    feats = []
    for _ in range(3):
        torch.manual_seed(90)
        feats.append(flow.inference(**kw)[0].cpu())
    want = g["feat_a"]
    rel = float((feats[0] - want).norm() / want.norm())
    assert rel < 2e-2, rel                                       # measured 4.2e-3 at this size (fp32 mode: 4e-7)
    assert torch.equal(feats[0], feats[1]) and torch.equal(feats[1], feats[2])
    assert flow.estimator.stat("graph_captures") == 1


def test_bf16_weight_mode_streams_the_rounded_matrices(lib):
    """weight_dtype=torch.bfloat16 (the model's fp16 mode, W16A32): the decode step reads bf16 copies of its matrices (cv_lm1_use_bf16).  With the fp32 kernels' lane
    layout (option gemv16_wide = 0) it must give, bit for bit, the logits of the fp32 step over the bf16-ROUNDED matrices (same lanes, same k order, the bf16 widened
    exactly); the default layout agrees to fp32 rounding; the tokens - host sampler and device loop - are those of the torch-eager port over the rounded state dict."""
    g = gold("cv1k_llm")
    kw = dict(text=g["text"], text_len=t(7), prompt_text=g["prompt_text"], prompt_text_len=t(4), prompt_speech_token=g["prompt_speech_token"],
              prompt_speech_token_len=t(9), embedding=g["embedding"])
    sd = W.make_cv1_llm(CFG)
    lm16 = CK.TransformerLM(sd, text_heads=CFG.text_heads, llm_heads=CFG.llm_heads, sampling=greedy, lib=lib, weight_dtype=torch.bfloat16)
    assert lm16.w16 and lm16.fused_step and lm16.llm._w16[0][0][0].dtype == torch.bfloat16
    rounded = lm16.sd
    assert not torch.equal(rounded["llm_decoder.weight"], sd["llm_decoder.weight"]) and torch.equal(rounded["llm_decoder.bias"], sd["llm_decoder.bias"])
    lm32 = CK.TransformerLM(rounded, text_heads=CFG.text_heads, llm_heads=CFG.llm_heads, sampling=greedy, lib=lib)          # fp32 storage of the same values
    lm16n = CK.TransformerLM(sd, text_heads=CFG.text_heads, llm_heads=CFG.llm_heads, sampling=greedy, lib=lib, weight_dtype=torch.bfloat16)
    lm16n.set_step_option("gemv16_wide", 0)                       # the fp32 kernels' lane layout over the bf16 matrices: their bits
    seen = {}
    for name, lm in (("w16", lm16), ("w16n", lm16n), ("w32", lm32)):
        rows = []
        lm.sampling = lambda scores, decoded, sampling, _r=rows: (_r.append(scores.clone()), int(scores.argmax().item()))[1]
        seen[name] = (list(lm.inference(max_token_text_ratio=6, min_token_text_ratio=2, **kw)), torch.stack(rows))
    assert seen["w16n"][0] == seen["w32"][0] and torch.equal(seen["w16n"][1], seen["w32"][1])
    # the default layout (lm1_gemv16_kernel: 16-byte loads, whole rows per lane group) sums a row in another k order: the same tokens, logits to fp32 rounding
    assert seen["w16"][0] == seen["w32"][0] and not torch.equal(seen["w16"][1], seen["w32"][1])
    torch.testing.assert_close(seen["w16"][1], seen["w32"][1], rtol=1e-4, atol=1e-4)
    ref = C1.TransformerLM(rounded, text_heads=CFG.text_heads, llm_heads=CFG.llm_heads, sampling=greedy)
    want = list(ref.inference(max_token_text_ratio=6, min_token_text_ratio=2, **kw))
    assert seen["w16"][0] == want and len(want) >= 14
    loop = CK.TransformerLM(sd, text_heads=CFG.text_heads, llm_heads=CFG.llm_heads, sampling="greedy", lib=lib, weight_dtype=torch.bfloat16, decode_chunk=5)
    assert list(loop.inference(max_token_text_ratio=6, min_token_text_ratio=2, **kw)) == want


def test_gemv_rows_option_keeps_the_bits(lib):
    """cv_lm1 options "gemv_rows" / "gemv_rows16": 1, 2 or 4 output rows per 16-lane group of the decode GEMVs - a row's products and their order do not depend on it."""
    g = gold("cv1k_llm")
    kw = dict(text=g["text"], text_len=t(7), prompt_text=g["prompt_text"], prompt_text_len=t(4), prompt_speech_token=g["prompt_speech_token"],
              prompt_speech_token_len=t(9), embedding=g["embedding"])
    sd = W.make_cv1_llm(CFG)
    for dtype, opt in ((None, "gemv_rows"), (torch.bfloat16, "gemv_rows16")):
        seen = {}
        for rows in (1, 2, 4):
            lm = CK.TransformerLM(sd, text_heads=CFG.text_heads, llm_heads=CFG.llm_heads, sampling=greedy, lib=lib, weight_dtype=dtype)
            if dtype is not None:
                lm.set_step_option("gemv16_wide", 0)              # (the rows option belongs to lm1_gemv_kernel)
            lm.set_step_option(opt, rows)
            logits = []
            lm.sampling = lambda scores, decoded, sampling, _r=logits: (_r.append(scores.clone()), int(scores.argmax().item()))[1]
            seen[rows] = (list(lm.inference(max_token_text_ratio=3, min_token_text_ratio=2, **kw)), torch.stack(logits))
        assert seen[1][0] == seen[2][0] == seen[4][0] and torch.equal(seen[1][1], seen[2][1]) and torch.equal(seen[1][1], seen[4][1])


def test_wide_bf16_gemv_kernels_at_a_middle_size(lib):
    """lm1_gemv16_kernel's other two forms, which the 128-wide test model does not reach: whole rows per lane group over 16 rows per workgroup (N >= 2048: the
    [4 d][d] and [ffn][d] products) and K split over 8 waves (the [d][ffn] product, ffn > 1024) - a 2-layer model with d = 512, ffn = 1280: tokens of the torch-eager
    port over the rounded state dict, logits of the narrow layout to fp32 rounding."""
    import dataclasses
    cfg = dataclasses.replace(CFG, llm_dim=512, text_heads=8, text_ffn=128, text_blocks=1, llm_heads=8, llm_ffn=1280, llm_blocks=2)
    sd = W.make_cv1_llm(cfg)
    gen = torch.Generator().manual_seed(5)
    text = torch.randint(0, cfg.text_vocab, (1, 6), generator=gen, dtype=torch.int32)
    e0 = torch.zeros(1, 0, dtype=torch.int32)
    kw = dict(text=text, text_len=t(6), prompt_text=e0, prompt_text_len=t(0), prompt_speech_token=e0, prompt_speech_token_len=t(0), embedding=torch.randn(1, cfg.spk_dim, generator=gen),
              max_token_text_ratio=2, min_token_text_ratio=2)
    seen = {}
    for wide in (1, 0, 32):                                        # 32: fp32 storage of the rounded values on the fp32 form of the wide kernel (option gemv_wide)
        lm = CK.TransformerLM(sd if wide != 32 else seen[1][2], text_heads=cfg.text_heads, llm_heads=cfg.llm_heads, sampling=greedy, lib=lib,
                              weight_dtype=torch.bfloat16 if wide != 32 else None)
        assert lm.fused_step
        lm.set_step_option("gemv16_wide" if wide != 32 else "gemv_wide", 1 if wide else 0)
        rows = []
        lm.sampling = lambda scores, decoded, sampling, _r=rows: (_r.append(scores.clone()), int(scores.argmax().item()))[1]
        seen[wide] = (list(lm.inference(**kw)), torch.stack(rows), lm.sd)
    ref = C1.TransformerLM(seen[1][2], text_heads=cfg.text_heads, llm_heads=cfg.llm_heads, sampling=greedy)
    assert seen[1][0] == seen[0][0] == seen[32][0] == list(ref.inference(**kw)) and len(seen[1][0]) == 12
    torch.testing.assert_close(seen[1][1], seen[0][1], rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(seen[32][1], seen[0][1], rtol=1e-4, atol=1e-4)
    assert not torch.equal(seen[1][1], seen[0][1]) and not torch.equal(seen[32][1], seen[0][1])
