"""The oracle held to the REAL reference classes at the FULL model dimensions (CPU only; nothing here reads /root/reference at run time).

tests/test_oracle_golden.py pins the oracle's arithmetic at test dimensions.  The full-size parity tests (tests/test_zz_fullsize.py, -m gpu) and every bench run then
compare the kernels with the ORACLE's full-size output - tests/golden/u10_oracle_tokens.json, cv3_u10_oracle_tokens.json, oracle.flow / oracle.hift evaluated on the
spot.  tests/golden/make_golden_fullsize.py ran the real `Qwen2LM`, `CosyVoice3LM`, `TransformerLM`, `CausalMaskedDiffWithXvec` and `HiFTGenerator` themselves on those
benchmark requests, with the same seeded full-size weights (loaded strict=True); this file closes the chain

    kernels == oracle (GPU tests, bench self-check)   and   oracle == real reference class (here)        at the size the benchmark runs.

The LM checks are exact (all 250 / 250 / 500 greedy ids); the flow estimator is held to the reference's own export tolerance (bin/export_onnx.py:109) and tighter;
the estimator's blocks are the restated Matcha classes on both sides (tests/golden/matcha_stub.py) - that part of the pin is unchanged by size."""
import json
import os

import numpy as np
import torch

from cosyvoice_amd import configs as CF, synthetic as W
from oracle import flow as OF
from oracle import hift as OH
from oracle import llm as OL

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
N_GEN, N_TEXT, N_PROMPT_TEXT, N_PROMPT_TOK = 250, 30, 12, 87


def load(name):
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(G, name + ".npz")).items()}


def _json(name):
    with open(os.path.join(G, name)) as f:
        return json.load(f)


def test_u10_oracle_tokens_are_the_real_qwen2lm_tokens():
    """All 250 greedy ids of U10 (what bench.py and test_zz_fullsize.py hold the MI355X path to) are the ids the real cosyvoice.llm.llm.Qwen2LM.inference yields at
    CosyVoice2-0.5B dimensions; the per-step top-2 margins of the two agree to fp32 round-off (so "near-tie" means the same thing on both sides)."""
    g, j = load("fullsize_llm"), _json("u10_oracle_tokens.json")
    assert len(j["tokens"]) == N_GEN and j["tokens"] == g["tokens"].tolist()
    assert np.abs(np.array(j["top2_margin"]) - g["top2_margin"].numpy()).max() < 2e-4


def test_u10_ras_oracle_tokens_are_the_real_sampler_tokens():
    """The 250 repetition-aware-sampled ids of tests/golden/u10_ras_oracle_tokens.json (149 distinct; bench.py replays them on the device with the stored variates) are what
    the real Qwen2LM produces with the REAL ras_sampling / nucleus_sampling / random_sampling (utils/common.py:138-167) when its multinomial draws are inverse-CDF draws on
    those variates - the same 14 steps take the repetition fallback."""
    g, j = load("fullsize_llm_ras"), _json("u10_ras_oracle_tokens.json")
    assert len(j["tokens"]) == N_GEN and j["tokens"] == g["tokens"].tolist()
    assert int(g["fallback_draws"]) == j["fallback_draws"] == 14 and int(g["distinct"]) == j["distinct"] == len(set(j["tokens"]))


def test_cv3_oracle_tokens_are_the_real_cosyvoice3lm_tokens():
    g, j = load("fullsize_llm_cv3"), _json("cv3_u10_oracle_tokens.json")
    assert len(j["tokens"]) == N_GEN and j["tokens"] == g["tokens"].tolist()
    assert np.abs(np.array(j["top2_margin"]) - g["top2_margin"].numpy()).max() < 2e-4


def test_llm_log_probs_fullsize():
    """The oracle's log-prob rows of the first three decode steps of U10 (prefill of 131 rows, contexts 131..133) against the real class's, full size."""
    g = load("fullsize_llm")
    lc, fc, _ = W.cv2()
    sd = W.make_llm(lc)
    u = W.synthetic_utterance(lc, fc, n_prompt_tok=N_PROMPT_TOK, n_prompt_text=N_PROMPT_TEXT, n_text=N_TEXT)
    trace = {}
    with torch.inference_mode():
        toks = OL.inference(sd, lc, u["text"], u["prompt_text"], u["llm_prompt_speech_token"], max_token_text_ratio=0.1, min_token_text_ratio=0.1, trace=trace)
    assert toks == g["tokens"][:3].tolist() and g["logp_steps"][:3].tolist() == [0, 1, 2]
    lp, ref = torch.stack(trace["logp"][:3]), g["logp"][:3].clone()
    ref[:, lc.speech_token_size] = lp[:, lc.speech_token_size]          # the reference logged its rows after the in-place EOS mask
    torch.testing.assert_close(lp, ref, rtol=1e-4, atol=1e-4)


def test_length_rule_against_the_reference():
    """The reference computes its length bounds as int((text_len - prompt_text_len) * ratio) with int32 TENSOR lengths, i.e. in float32 (llm/llm.py:497-498); the
    oracle and the product use python's double arithmetic.  The two agree for the reference's own ratios (20 and 2), for every ratio that is a
    multiple of 1/8 and for the bounded-length ratios of the golden generators; they part by one token for ratios like 250 / 30 (float32: 249.99998 -> 249), which is
    why tests/golden/make_golden_fullsize.py asks the real class for (N + 0.5) / n_text."""
    ref = lambda n, r: int(torch.tensor([n], dtype=torch.int32) * r)
    for n in list(range(1, 400)) + [1000, 4096]:
        for r in (20, 2, 1.5, 0.125, 4, 8.375, 10):
            assert ref(n, r) == int(n * r), (n, r)
    for n_gen, n_text in ((250, 30), (125, 30), (375, 30), (500, 30), (500, 25), (40, 30)):
        assert ref(n_text, (n_gen + 0.5) / n_text) == int(n_text * ((n_gen + 0.5) / n_text)) == n_gen
    assert ref(30, 250 / 30) == 249 and int(30 * (250 / 30)) == 250   # the known one-token difference, stated rather than hidden


def test_cv1_port_tokens_fullsize():
    """bench.py's cosyvoice300m record checks the kernels' 500 ids against the torch-eager port (cosyvoice1.py); the real TransformerLM (llm/llm.py:162-223) ran that
    request at CosyVoice-300M dimensions (500 ids, tests/golden/fullsize_cv1_llm.npz): the port yields the same ids (first 40 here; bench.py compares all 500 with the
    file on the GPU box) and the same log-prob rows."""
    from cosyvoice_amd import cosyvoice1 as C1
    g = load("fullsize_cv1_llm")
    cfg, _ = W.cv1()
    sd = W.make_cv1_llm(cfg)
    rows = []

    def greedy(scores, decoded, sampling):
        rows.append(scores.clone())
        return int(scores.argmax().item())
    gen = torch.Generator().manual_seed(300)                             # bench.py cv1_workload
    text = torch.randint(0, cfg.text_vocab, (1, 25), generator=gen, dtype=torch.int32)
    emb = torch.randn(1, cfg.spk_dim, generator=gen)
    assert torch.equal(text, g["text"]) and torch.equal(emb, g["embedding"])
    e0 = torch.zeros(1, 0, dtype=torch.int32)
    tl = lambda n: torch.tensor([n], dtype=torch.int32)
    n_chk = 40
    lm = C1.TransformerLM(sd, text_heads=cfg.text_heads, llm_heads=cfg.llm_heads, sampling=greedy)
    toks = list(lm.inference(text=text, text_len=tl(25), prompt_text=e0, prompt_text_len=tl(0), prompt_speech_token=e0, prompt_speech_token_len=tl(0), embedding=emb,
                             max_token_text_ratio=n_chk / 25, min_token_text_ratio=n_chk / 25))
    assert toks == g["tokens"][:n_chk].tolist() and len(g["tokens"]) == 500
    ref = g["logp"][:2].clone()
    lp = torch.stack(rows[:2])
    ref[:, cfg.speech_token_size] = lp[:, cfg.speech_token_size]
    torch.testing.assert_close(lp, ref, rtol=1e-4, atol=1e-4)


def test_flow_fullsize():
    """oracle.flow against the real CausalMaskedDiffWithXvec at CosyVoice2 dimensions on U10: the estimator boundary at T = 674 (both mask modes: the reference's own
    export tolerance rtol 1e-2 / atol 1e-4, and 5e-4), the encoder at 337 tokens, flow.inference with 10 Euler steps (mel [80, 500]) and the first streamed chunk."""
    g = load("fullsize_flow")
    lc, fc, _ = W.cv2()
    sd = W.make_flow(fc)
    u = W.synthetic_utterance(lc, fc, n_prompt_tok=N_PROMPT_TOK, n_prompt_text=N_PROMPT_TEXT, n_text=N_TEXT)
    token = torch.tensor(_json("u10_oracle_tokens.json")["tokens"], dtype=torch.int32).unsqueeze(0)
    gen = torch.Generator().manual_seed(12)
    T = 2 * (N_PROMPT_TOK + N_GEN)
    x = torch.randn(2, 80, T, generator=gen); mu = torch.randn(2, 80, T, generator=gen); cond = torch.randn(2, 80, T, generator=gen)
    spk = torch.randn(2, 80, generator=gen); t = torch.tensor([0.25, 0.25]); mask = torch.ones(2, 1, T)
    with torch.inference_mode():
        for streaming, key in ((False, "est_full"), (True, "est_stream")):
            out = OF.estimator(sd, fc, x, mask, mu, t, spk, cond, streaming)
            for got, want in ((out[0], g[key]), (out[1, :, ::8], g[key + "_row1"])):
                torch.testing.assert_close(got, want, rtol=1e-2, atol=1e-4)
                torch.testing.assert_close(got, want, rtol=5e-4, atol=5e-4)
        assert not torch.allclose(g["est_full"], g["est_stream"], atol=1e-3)             # the chunk mask bites at T = 674
        tok = torch.cat([u["flow_prompt_speech_token"], token], 1).long()
        h = OF.encoder(sd, fc, sd["input_embedding.weight"][tok[0]].unsqueeze(0), None, False)
        torch.testing.assert_close(h[0, ::4], g["enc_full"], rtol=5e-4, atol=5e-4)
        mel = OF.inference(sd, fc, token, u["flow_prompt_speech_token"], u["prompt_speech_feat"], u["flow_embedding"], streaming=False, finalize=True)
        assert mel.shape == (1, 80, 2 * N_GEN)
        torch.testing.assert_close(mel[0], g["mel_full"], rtol=2e-3, atol=2e-3)
        n1 = 25 + 13 + 3
        mel = OF.inference(sd, fc, token[:, :n1], u["flow_prompt_speech_token"], u["prompt_speech_feat"], u["flow_embedding"], streaming=True, finalize=False)
        torch.testing.assert_close(mel[0], g["mel_chunk"], rtol=2e-3, atol=2e-3)


def test_hift_fullsize():
    """oracle.hift against the real HiFTGenerator (24 kHz dimensions) on 100 frames of U10's mel: f0, the harmonic source (phase integration amplifies fp32 round-off:
    2e-3 as at test dimensions) and the decoder pinned tightly by feeding it the reference's own source."""
    g = load("fullsize_hift")
    hc = W.cv2()[2]
    sd = W.make_hift(hc)
    mel = g["mel"].unsqueeze(0)
    m = mel.shape[2]
    with torch.inference_mode():
        torch.testing.assert_close(OH.f0_predictor(sd, mel), g["f0"], rtol=1e-4, atol=1e-3)
        torch.manual_seed(99)                                            # the draws HiFTGenerator.inference consumed (generator.py:245, :312), in order
        rand_ini = torch.rand(1, 9); rand_ini[:, 0] = 0
        noise = torch.randn_like(torch.empty(1, 9, 480 * m).transpose(1, 2))
        speech, source = OH.inference(sd, hc, mel, None, rand_ini, noise)
        assert speech.shape == g["speech"].shape == (1, 480 * m)
        torch.testing.assert_close(source, g["source"], rtol=0, atol=2e-3)
        torch.testing.assert_close(OH.decode(sd, hc, mel, g["source"]), g["speech"], rtol=1e-4, atol=1e-4)
    assert (g["f0"] > hc.voiced_thr).any()


def test_mixed64_oracle_tokens_are_the_real_qwen2lm_tokens():
    """All 20 000 ids of the 64 utterances of the mixed64 workload (BASELINE.json configs[3]; bench.py checks every utterance of its batched / queued runs against
    tests/golden/mixed64_oracle_tokens.json) are the real Qwen2LM's - including the steps whose top-2 margin is a few 1e-5."""
    g, j = np.load(os.path.join(G, "fullsize_mixed64.npz")), _json("mixed64_oracle_tokens.json")
    assert len(j["utterances"]) == 64
    n = 0
    for ut in j["utterances"]:
        assert ut["tokens"] == g["tokens_%02d" % ut["index"]].tolist() and len(ut["tokens"]) == ut["n_gen"]
        n += len(ut["tokens"])
    assert n == 16 * (125 + 250 + 375 + 500)


def test_dit_flow_fullsize():
    """oracle.dit against the real CausalMaskedDiffWithDiT / DiT at Fun-CosyVoice3-0.5B dimensions (22 blocks x 1024) on bench.py's cosyvoice3 request: the estimator
    boundary at T = 674 in both mask modes and inference with two Euler steps."""
    from oracle import dit as OD
    g = load("fullsize_dit")
    lc, fc = CF.cv3_llm(), CF.cv3_flow()
    sd = W.make_flow_dit(fc)
    u = W.synthetic_utterance(lc, fc, n_prompt_tok=N_PROMPT_TOK, n_prompt_text=24, n_text=N_TEXT, seed=2025)
    token = torch.tensor(_json("cv3_u10_oracle_tokens.json")["tokens"], dtype=torch.int32).unsqueeze(0)
    gen = torch.Generator().manual_seed(14)
    T = 2 * (N_PROMPT_TOK + N_GEN)
    x = torch.randn(2, 80, T, generator=gen); mu = torch.randn(2, 80, T, generator=gen); cond = torch.randn(2, 80, T, generator=gen)
    spk = torch.randn(2, 80, generator=gen); t = torch.tensor([0.25, 0.25]); mask = torch.ones(2, 1, T)
    with torch.inference_mode():
        for streaming, key in ((False, "est_full"), (True, "est_stream")):
            out = OD.estimator(sd, fc, x, mask, mu, t, spk, cond, streaming)
            for got, want in ((out[0, :, ::2], g[key]), (out[1, :, ::8], g[key + "_row1"])):
                torch.testing.assert_close(got, want, rtol=1e-2, atol=1e-4)      # the reference's own export tolerance (bin/export_onnx.py:109)
                torch.testing.assert_close(got, want, rtol=5e-4, atol=5e-4)
        assert not torch.allclose(g["est_full"], g["est_stream"], atol=1e-3)
        mel = OD.inference(sd, fc, token, u["flow_prompt_speech_token"], u["prompt_speech_feat"], u["flow_embedding"], streaming=False, finalize=True, n_timesteps=2)
        assert mel.shape == (1, 80, 2 * N_GEN)
        torch.testing.assert_close(mel[0], g["mel_2steps"], rtol=2e-3, atol=2e-3)


def test_causal_hift_fullsize():
    """oracle.hift causal_* against the real CausalHiFTGenerator (float64 f0 predictor) at Fun-CosyVoice3-0.5B dimensions, one-shot and as a non-final chunk."""
    g = load("fullsize_causal_hift")
    hc = CF.cv3_hift()
    sd = W.make_hift(hc)
    mel = g["mel"].unsqueeze(0)
    m = mel.shape[2]
    gen = torch.Generator().manual_seed(16)                             # the noise buffers make_golden_fullsize.golden_causal_hift gave the real generator
    rand_ini = torch.rand(1, 9, generator=gen); rand_ini[:, 0] = 0
    noise = torch.rand(1, 480 * m, 9, generator=gen)
    with torch.inference_mode():
        torch.testing.assert_close(OH.causal_f0_predictor(sd, mel, True), g["f0"], rtol=1e-5, atol=1e-4)
        speech, source = OH.causal_inference(sd, hc, mel, True, rand_ini, noise)
        torch.testing.assert_close(source, g["source"], rtol=0, atol=2e-3)
        assert speech.shape == g["speech"].shape
        torch.testing.assert_close(OH.causal_decode(sd, hc, mel, g["source"], True), g["speech"], rtol=1e-4, atol=1e-4)
        speech_c, source_c = OH.causal_inference(sd, hc, mel[:, :, :60], False, rand_ini, noise)
        torch.testing.assert_close(source_c, g["source_c"], rtol=0, atol=2e-3)
        assert speech_c.shape == g["speech_c"].shape                   # a non-final chunk: the f0 predictor holds 3 frames back (generator.py:719-722), conv_pre 4 more
        torch.testing.assert_close(OH.causal_decode(sd, hc, mel[:, :, :60 - 3], g["source_c"], False), g["speech_c"], rtol=1e-4, atol=1e-4)


def test_cv1_port_flow_and_hift_fullsize():
    """The torch-eager CosyVoice-300M port (cosyvoice_amd/cosyvoice1.py - what tests/test_zzzz_cosyvoice1_fullsize.py and bench.py hold the kernels to at full size)
    against the real MaskedDiffWithXvec (InterpolateRegulator, flow cache, the U-Net ConditionalDecoder) and the real 22.05 kHz HiFTGenerator at CosyVoice-300M
    dimensions on bench.py's inference_sft request (500 ids -> 861 mel frames); global-RNG draws seeded as the generator seeded them."""
    from cosyvoice_amd import cosyvoice1 as C1
    cfg, hcfg = W.cv1()
    gl, gf, gh = load("fullsize_cv1_llm"), load("fullsize_cv1_flow"), load("fullsize_cv1_hift")
    tl = lambda n: torch.tensor([n], dtype=torch.int32)
    e0 = torch.zeros(1, 0, dtype=torch.int32)
    flow = C1.MaskedDiffWithXvec(W.make_cv1_flow(cfg), enc_heads=cfg.flow_heads, est_heads=cfg.est_heads, input_frame_rate=cfg.input_frame_rate)
    token = gl["tokens"].to(torch.int32).unsqueeze(0)
    torch.manual_seed(41)
    feat, cache = flow.inference(token=token, token_len=tl(500), prompt_token=e0, prompt_token_len=tl(0), prompt_feat=torch.zeros(1, 0, 80), prompt_feat_len=tl(0),
                                 embedding=gl["embedding"], flow_cache=torch.zeros(1, 80, 0, 2))
    assert feat.shape == (1, 80, int(500 / 50 * 22050 / 256))
    torch.testing.assert_close(feat[0], gf["feat"], rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(cache[0, :, -8:], gf["cache_tail"], rtol=1e-4, atol=1e-4)
    h = C1.HiFTGenerator(W.make_hift(hcfg), sampling_rate=hcfg.sr, upsample_rates=hcfg.ups, upsample_kernel_sizes=hcfg.up_k, source_resblock_kernel_sizes=hcfg.src_k)
    mel = gh["feat"].unsqueeze(0)
    torch.testing.assert_close(h.f0_predictor(mel), gh["f0"], rtol=1e-4, atol=1e-3)
    torch.manual_seed(77)
    speech, source = h.inference(speech_feat=mel)
    assert speech.shape == gh["speech"].shape == (1, 100 * 256)
    torch.testing.assert_close(source, gh["source"], rtol=0, atol=2e-3)      # (the harmonic phase is a cumulative sum: fp32 summation order shows at ~1e-4 in sin(phase))
    torch.testing.assert_close(speech, gh["speech"], rtol=0, atol=5e-3)


def test_cv1_whole_request_fullsize():
    """bench.py's CosyVoice-300M request end to end: the torch-eager port's CosyVoiceModel.tts (cosyvoice1.py) against the REAL cli.model.CosyVoiceModel.tts around the real
    full-size MaskedDiffWithXvec + 22.05 kHz HiFTGenerator, offline, scripted LLM = the real TransformerLM's 500 ids, global RNG seeded as the generator seeded it:
    220 416 samples, every 8th stored."""
    from cosyvoice_amd import cosyvoice1 as C1
    cfg, hcfg = W.cv1()
    gl, g = load("fullsize_cv1_llm"), load("fullsize_model_cv1")
    tokens = gl["tokens"].tolist()

    class ScriptedLLM:
        def inference(self, **kw):
            yield from tokens
    flow = C1.MaskedDiffWithXvec(W.make_cv1_flow(cfg), enc_heads=cfg.flow_heads, est_heads=cfg.est_heads, input_frame_rate=cfg.input_frame_rate)
    hift = C1.HiFTGenerator(W.make_hift(hcfg), sampling_rate=hcfg.sr, upsample_rates=hcfg.ups, upsample_kernel_sizes=hcfg.up_k, source_resblock_kernel_sizes=hcfg.src_k)
    m = C1.CosyVoiceModel(ScriptedLLM(), flow, hift)
    torch.manual_seed(55)
    chunks = [o["tts_speech"] for o in m.tts(text=gl["text"], flow_embedding=gl["embedding"], llm_embedding=gl["embedding"], stream=False)]
    assert [c.shape[1] for c in chunks] == g["offline_n"].tolist() == [220416]
    torch.testing.assert_close(torch.cat(chunks, 1)[:, ::8], g["offline"], rtol=0, atol=5e-3)
    assert float(g["offline"].abs().max()) > 0.05
