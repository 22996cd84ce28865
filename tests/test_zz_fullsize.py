"""Parity at the REAL CosyVoice2-0.5B dimensions (BASELINE.json configs[1]).  The other parity tests run emulator-sized models, so the
tile shapes, GEMV widths and attention sizes the real model launches were only exercised by bench.py, which checks nothing.  On the
MI355X these tests build the full-size LLM / flow / HiFT (seeded random weights) and compare a bounded piece of each stage with the CPU
oracle (a few seconds of CPU work each); under the CPU emulator the same code runs on the tiny configuration.

Criteria are relative L2 errors, calibrated on the MI355X: the errors measured there are recorded by `_record` (committed as
profiles/r2_fullsize_errors.json) and every bound below is <= 3x the recorded value (fp32 mode ~2e-6: summation order only; bf16 mode
~6e-3: the operand rounding, equal to the distance of the oracle's own bf16 mirror from its fp32 result).  Greedy ids must match
wherever the oracle's own top-2 margin is clear."""
import os

import pytest
import torch

from cosyvoice_amd.flow import CausalMaskedDiffWithXvec
from cosyvoice_amd.hift import HiFTGenerator
from cosyvoice_amd.llm import Qwen2LM
from oracle import flow as OF
from oracle import hift as OH
from oracle import llm as OL
from cosyvoice_amd import synthetic as W


def _cfgs(lib):
    return W.tiny() if lib.emulated else W.cv2()


def _rel(a, b):
    return ((a - b).norm() / b.norm().clamp_min(1e-12)).item()


def _record(lib, name, value):
    """Measured full-size errors go to gpurun_out/r3_fullsize_errors.json on the GPU box (copied to profiles/ afterwards): the bounds in this
    file are calibrated from that record (<= 3x what was measured)."""
    print("%s: %.3e" % (name, value))
    if lib.emulated:
        return
    import json, os
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(d, exist_ok=True)
    f = os.path.join(d, "r3_fullsize_errors.json")
    rec = json.load(open(f)) if os.path.exists(f) else {}
    rec[name] = value
    json.dump(rec, open(f, "w"), indent=1, sort_keys=True)


def test_llm_fullsize(lib):
    lc = _cfgs(lib)[0]
    sd = W.make_llm(lc)
    u = W.synthetic_utterance(lc, _cfgs(lib)[1], n_prompt_tok=20, n_prompt_text=6, n_text=8, seed=5)
    lm = Qwen2LM(sd, lc, lib=lib, max_len=256, sampling="greedy", decode_chunk=4)
    n_steps = 6
    trace = {}
    want = OL.inference(sd, lc, u["text"], u["prompt_text"], u["llm_prompt_speech_token"], max_token_text_ratio=n_steps / 8, min_token_text_ratio=n_steps / 8,
                        trace=trace)
    lm_input = lm.build_lm_input(u["text"], u["prompt_text"], u["llm_prompt_speech_token"])
    torch.testing.assert_close(lm_input.cpu(), OL.build_lm_input(sd, lc, u["text"], u["prompt_text"], u["llm_prompt_speech_token"]), rtol=0, atol=0)
    lm.prefill(lm_input)
    sp = lm.make_sampling(n_steps, n_steps)
    for i in range(len(want)):
        toks, _ = lm.decode(1, sp)
        logp = lm.last_logits().log_softmax(-1)
        ref = trace["logp"][i]
        err = _rel(logp, ref)
        print("llm step %d: rel L2 of log-probs %.2e" % (i, err))
        assert err < 1e-3, (i, err)
        top2 = torch.topk(ref.masked_fill(torch.arange(ref.numel()) == lc.speech_token_size, -float("inf")), 2).values
        if (top2[0] - top2[1]).item() > 2e-2:                     # not a near-tie in the oracle itself -> the id must match
            assert toks == [want[i]], (i, toks, want[i])
        else:
            break                                                  # past a near-tie the two sequences may legitimately diverge


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_flow_estimator_fullsize(lib, precision):
    fc = _cfgs(lib)[1]
    sd = W.make_flow(fc)
    flow = CausalMaskedDiffWithXvec(sd, fc, lib=lib, precision=precision)
    g = torch.Generator().manual_seed(12)
    T = 41 if lib.emulated else 200                                # 200 frames: every GEMM tile shape of the U10 run, 4 query tiles
    x = torch.randn(2, 80, T, generator=g); mu = torch.randn(2, 80, T, generator=g); cond = torch.randn(2, 80, T, generator=g)
    spk = torch.randn(2, 80, generator=g); t = torch.tensor([0.25, 0.25]); mask = torch.ones(2, 1, T)
    for streaming in (False, True):
        out = flow.decoder.estimator(x, mask, mu, t, spk, cond, streaming=streaming).cpu()
        ref = OF.estimator(sd, fc, x, mask, mu, t, spk, cond, streaming)
        err = _rel(out, ref)
        print("estimator %s streaming=%s: rel L2 %.2e" % (precision, streaming, err))
        assert err < (1e-5 if precision == "fp32" else 2e-2), (precision, streaming, err)
    h, _ = flow.encoder(torch.randn(1, 24 if lib.emulated else 60, fc.dim, generator=g), torch.tensor([60]), streaming=False)
    # (encoder output only has to be finite here; its parity is covered at tiny size and through inference() below at full size)
    assert torch.isfinite(h).all()


def test_flow_inference_fullsize(lib):
    fc = _cfgs(lib)[1]
    sd = W.make_flow(fc)
    n_p, n_t = (7, 13) if lib.emulated else (20, 40)
    g = torch.Generator().manual_seed(6)
    tok = torch.randint(0, fc.vocab, (1, n_t), generator=g, dtype=torch.int32); ptok = torch.randint(0, fc.vocab, (1, n_p), generator=g, dtype=torch.int32)
    pfeat = torch.randn(1, 2 * n_p, 80, generator=g) * 2 - 5; emb = torch.randn(1, fc.spk_dim, generator=g)
    n = lambda k: torch.tensor([k], dtype=torch.int32)
    flow = CausalMaskedDiffWithXvec(sd, fc, lib=lib, n_timesteps=2)
    mel, _ = flow.inference(token=tok, token_len=n(n_t), prompt_token=ptok, prompt_token_len=n(n_p), prompt_feat=pfeat, prompt_feat_len=n(2 * n_p),
                            embedding=emb, streaming=False, finalize=True)
    ref = OF.inference(sd, fc, tok, ptok, pfeat, emb, streaming=False, finalize=True, n_timesteps=2)
    err = _rel(mel.cpu(), ref)
    print("flow.inference (2 Euler steps): rel L2 %.2e" % err)
    assert mel.shape == ref.shape and err < 1e-5, err


def test_hift_decode_fullsize(lib):
    hc = _cfgs(lib)[2]
    sd = W.make_hift(hc)
    hift = HiFTGenerator(sd, hc, lib=lib)
    gen = torch.Generator().manual_seed(4)
    m = 7 if lib.emulated else 30
    mel = torch.randn(1, 80, m, generator=gen) * 2 - 5
    s = torch.tanh(torch.randn(1, 1, 480 * m, generator=gen))
    out = hift.decode(mel, s).cpu()
    ref = OH.decode(sd, hc, mel, s)
    err = _rel(out, ref)
    print("hift.decode: rel L2 %.2e" % err)
    assert out.shape == ref.shape and err < 1e-5, err


# ---------------------------------------------------------------------------------------------------------------------------------
# The benchmark workload itself (U10, BASELINE.json configs[1]) under parity: 250 greedy tokens with the context growing 131 -> 381,
# the estimator at T = 674, flow.inference with 10 Euler steps at 337 tokens, HiFT at 500 frames.
# ---------------------------------------------------------------------------------------------------------------------------------
N_GEN, N_TEXT, N_PROMPT_TEXT, N_PROMPT_TOK = 250, 30, 12, 87
E2E_MIN_SNR_DB = 42.0      # device f0 -> phase -> source -> waveform vs the oracle's, 500 frames: measured 52.3 dB (HiFT v2) / 51.6 dB (causal, vs the float64 f0) on the
                           # MI355X (profiles/r3_fullsize_errors.json); the bound allows 3x the measured error amplitude


def _u10(lib):
    import json, os
    if lib.emulated:                      # same code path at tiny dims: 12 tokens
        lc, fc, _ = W.tiny()
        return lc, fc, W.synthetic_utterance(lc, fc, n_prompt_tok=9, n_prompt_text=3, n_text=4, seed=3), 12, None
    lc, fc, _ = W.cv2()
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "u10_oracle_tokens.json")))
    return lc, fc, W.synthetic_utterance(lc, fc, n_prompt_tok=N_PROMPT_TOK, n_prompt_text=N_PROMPT_TEXT, n_text=N_TEXT), N_GEN, gold


def test_llm_u10_ras_replay(lib):
    """The second parity workload of the benchmark (round 4): U10 decoded with repetition-aware sampling from FIXED variates - 250 dependent decisions, ~150
    distinct ids, ~14 fallback draws - must reproduce the oracle's sampled sequence committed as tests/golden/u10_ras_oracle_tokens.json
    (tests/golden/make_u10_ras.py; every decision keeps a margin of 1e-3 by construction of the variates).  Hardware only: the file is at the real dimensions."""
    import json, numpy as np
    if lib.emulated:
        pytest.skip("hardware only: 250 sampled tokens at the real CosyVoice2-0.5B dimensions (the emulator runs the sampler's logic in tests/test_llm_ras.py)")
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "u10_ras_oracle_tokens.json")))
    lc, fc, u, n_gen, _ = _u10(lib)
    lm = Qwen2LM(W.make_llm(lc), lc, lib=lib, max_len=1024, sampling="ras", decode_chunk=64)
    lm.set_uniforms(np.asarray(gold["variates"], dtype=np.float32))
    t = lambda n: torch.tensor([n], dtype=torch.int32)
    ratio = n_gen / u["text"].shape[1]
    got = list(lm.inference(text=u["text"], text_len=t(u["text"].shape[1]), prompt_text=u["prompt_text"], prompt_text_len=t(u["prompt_text"].shape[1]),
                            prompt_speech_token=u["llm_prompt_speech_token"], prompt_speech_token_len=t(u["llm_prompt_speech_token"].shape[1]),
                            max_token_text_ratio=ratio, min_token_text_ratio=ratio))
    div = next((i for i, (a, b) in enumerate(zip(got, gold["tokens"])) if a != b), None)
    _record(lib, "llm_u10_ras_first_divergence_index", float(n_gen if div is None else div))
    assert len(got) == len(gold["tokens"]) == n_gen and div is None, (div, got[div], gold["tokens"][div], gold["margin"][div])


def test_llm_u10_all_tokens(lib):
    """Greedy ids of the whole benchmark decode (north_star: "speech-token ids bit-exact under greedy decode"), evaluated the way SURVEY.md
    section 8c prescribes: free-running first-divergence index against the oracle's committed tokens, and teacher-forced on the tokens the
    device produced - the oracle re-scores every position in ONE full-sequence pass and the device's choice must be the oracle's arg-max
    wherever the oracle's own top-2 margin is not a near-tie (<= 1e-3 in log-prob: fp32 summation-order noise is ~1e-5)."""
    lc, fc, u, n_gen, gold = _u10(lib)
    sd = W.make_llm(lc)
    lm = Qwen2LM(sd, lc, lib=lib, max_len=1024 if not lib.emulated else 128, sampling="greedy", decode_chunk=64)
    t = lambda n: torch.tensor([n], dtype=torch.int32)
    n_text = u["text"].shape[1]
    ratio = n_gen / n_text
    got = list(lm.inference(text=u["text"], text_len=t(n_text), prompt_text=u["prompt_text"], prompt_text_len=t(u["prompt_text"].shape[1]),
                            prompt_speech_token=u["llm_prompt_speech_token"], prompt_speech_token_len=t(u["llm_prompt_speech_token"].shape[1]),
                            max_token_text_ratio=ratio, min_token_text_ratio=ratio))
    assert len(got) == n_gen
    last_logp = lm.last_logits().log_softmax(-1)                # the head ran once more after the last token: the prediction for position n_gen
    # teacher-forced: one causal pass of the oracle over [lm_input ; embeddings of the device's tokens]
    x = torch.cat([OL.build_lm_input(sd, lc, u["text"], u["prompt_text"], u["llm_prompt_speech_token"]), sd["speech_embedding.weight"][torch.tensor(got)]], 0)
    with torch.no_grad():
        y = OL.Qwen2Oracle(sd, lc).forward(x)[-(n_gen + 1):]
        logp_all = torch.nn.functional.linear(y, sd["llm_decoder.weight"], sd.get("llm_decoder.bias")).log_softmax(-1).clone()
    logp, logp_next = logp_all[:n_gen], logp_all[n_gen]
    logp[:, lc.speech_token_size] = -float("inf")              # eos masked below min_len (every step of this workload)
    top2 = torch.topk(logp, 2, dim=-1)
    margin = top2.values[:, 0] - top2.values[:, 1]
    want = top2.indices[:, 0].tolist()
    mism = [i for i in range(n_gen) if got[i] != want[i]]
    near = [i for i in mism if margin[i].item() <= 1e-3 and got[i] == top2.indices[i, 1].item()]
    _record(lib, "llm_u10_teacher_forced_mismatches", float(len(mism)))
    _record(lib, "llm_u10_min_oracle_margin", margin.min().item())
    assert mism == near, ("device token is not the oracle arg-max at a clear margin", [(i, got[i], want[i], margin[i].item()) for i in mism if i not in near][:5])
    err = (last_logp - logp_next).abs().max().item()
    _record(lib, "llm_u10_logp_after_last_token_max_abs_err", err)
    assert err < 6e-5                                            # measured 1.8e-5 on the MI355X after 250 steps / context 381
    if gold is not None:                                         # free-running vs the committed oracle run
        ref = gold["tokens"]
        div = next((i for i in range(n_gen) if got[i] != ref[i]), n_gen)
        _record(lib, "llm_u10_first_divergence_index", float(div))
        if div < n_gen:
            assert gold["top2_margin"][div] <= 1e-3, ("free-running divergence at a clear margin", div, got[div], ref[div], gold["top2_margin"][div])
    else:
        assert got == OL.inference(sd, lc, u["text"], u["prompt_text"], u["llm_prompt_speech_token"], max_token_text_ratio=ratio, min_token_text_ratio=ratio)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_flow_estimator_u10(lib, precision):
    """The estimator exactly as the benchmark calls it: CFG batch 2, T = 674 frames (337 tokens), both mask modes."""
    fc = _cfgs(lib)[1]
    sd = W.make_flow(fc)
    flow = CausalMaskedDiffWithXvec(sd, fc, lib=lib, precision=precision)
    g = torch.Generator().manual_seed(13)
    T = 57 if lib.emulated else 674
    x = torch.randn(2, 80, T, generator=g); mu = torch.randn(2, 80, T, generator=g); cond = torch.randn(2, 80, T, generator=g)
    spk = torch.randn(2, 80, generator=g); t = torch.tensor([0.6, 0.6]); mask = torch.ones(2, 1, T)
    mu[1] = 0; cond[1] = 0; spk[1] = 0                          # the unconditional CFG row (flow_matching.py:103-108)
    for streaming in (False, True):
        out = flow.decoder.estimator(x, mask, mu, t, spk, cond, streaming=streaming).cpu()
        ref = OF.estimator(sd, fc, x, mask, mu, t, spk, cond, streaming)
        err = _rel(out, ref)
        _record(lib, "estimator_T674_%s_streaming%d_rel_l2" % (precision, int(streaming)), err)
        _record(lib, "estimator_T674_%s_streaming%d_max_abs" % (precision, int(streaming)), (out - ref).abs().max().item())
        # measured on the MI355X (profiles/r2_fullsize_errors.json): fp32 1.8e-6, bf16 6.0e-3 -> bounds at 3x
        assert err < (6e-6 if precision == "fp32" else 1.8e-2), (precision, streaming, err)


@pytest.mark.parametrize("band", [1, 0])
def test_flow_estimator_u10_large_m_kernels(lib, band):
    """Round 5: the LARGE-M kernel set of a shared pass (flow_big.h; selected by the row count, big_rows = 1 forces it) held to the fp32 ORACLE at
    full size - C = 256, 8 heads, K up to 1024, T = 674, CFG batch 2 - not only to the small-tile kernels at toy size.  band = 1: with the 64-row band launch of
    flow_band.h between attention and the next QKV GEMM (the default of a large pass); band = 0: the seven-launch form.  Same bounds as the small-tile test above, and
    bit-identical to it (the kernel choice may follow the row count of a pass)."""
    import ctypes as C
    fc = _cfgs(lib)[1]
    sd = W.make_flow(fc)
    flow = CausalMaskedDiffWithXvec(sd, fc, lib=lib, precision="bf16")
    g = torch.Generator().manual_seed(13)
    T = 57 if lib.emulated else 674
    x = torch.randn(2, 80, T, generator=g); mu = torch.randn(2, 80, T, generator=g); cond = torch.randn(2, 80, T, generator=g)
    spk = torch.randn(2, 80, generator=g); t = torch.tensor([0.6, 0.6]); mask = torch.ones(2, 1, T)
    mu[1] = 0; cond[1] = 0; spk[1] = 0
    try:
        for streaming in (False, True):
            lib.cv_flow_set_option(flow._h, b"big_rows", C.c_int32(0)); lib.cv_flow_set_option(flow._h, b"attn2_rows", C.c_int32(0))
            small = flow.decoder.estimator(x, mask, mu, t, spk, cond, streaming=streaming).cpu().clone()
            lib.cv_flow_set_option(flow._h, b"big_rows", C.c_int32(1)); lib.cv_flow_set_option(flow._h, b"fused_band", C.c_int32(band))
            out = flow.decoder.estimator(x, mask, mu, t, spk, cond, streaming=streaming).cpu()
            ref = OF.estimator(sd, fc, x, mask, mu, t, spk, cond, streaming)
            err = _rel(out, ref)
            _record(lib, "estimator_T674_bf16_large_m_band%d_streaming%d_rel_l2" % (band, int(streaming)), err)
            _record(lib, "estimator_T674_bf16_large_m_band%d_streaming%d_max_abs" % (band, int(streaming)), (out - ref).abs().max().item())
            assert err < 1.8e-2, (band, streaming, err)
            assert torch.equal(out, small), (band, streaming, (out - small).abs().max().item())
    finally:
        lib.cv_flow_set_option(flow._h, b"big_rows", C.c_int32(2000)); lib.cv_flow_set_option(flow._h, b"fused_band", C.c_int32(1))


def test_flow_inference_u10(lib):
    """flow.inference as token2wav calls it for U10 (87 prompt + 250 generated tokens, 174 prompt frames, 10 Euler steps, bf16 mode -
    the configuration bench.py times) against the fp32 oracle; SURVEY.md section 8c tolerance: mel max |diff| <= 5e-2."""
    fc = _cfgs(lib)[1]
    sd = W.make_flow(fc)
    n_p, n_t, steps = (7, 13, 2) if lib.emulated else (N_PROMPT_TOK, N_GEN, 10)
    g = torch.Generator().manual_seed(6)
    tok = torch.randint(0, fc.vocab, (1, n_t), generator=g, dtype=torch.int32); ptok = torch.randint(0, fc.vocab, (1, n_p), generator=g, dtype=torch.int32)
    pfeat = torch.randn(1, 2 * n_p, 80, generator=g) * 2 - 5; emb = torch.randn(1, fc.spk_dim, generator=g)
    n = lambda k: torch.tensor([k], dtype=torch.int32)
    ref = OF.inference(sd, fc, tok, ptok, pfeat, emb, streaming=False, finalize=True, n_timesteps=steps)
    # measured on the MI355X: fp32 6.8e-7 / 4.5e-6, bf16 2.2e-3 / 1.23e-2 (outputs of std 1.33) -> bounds at 3x; the bf16 max stays inside
    # the stated 5e-2 mel tolerance of the mode
    mels = {}
    for precision, bound_l2, bound_max in (("fp32", 3e-6, 1.5e-5), ("bf16", 7e-3, 4e-2)):
        flow = CausalMaskedDiffWithXvec(sd, fc, lib=lib, n_timesteps=steps, precision=precision)
        mel, _ = flow.inference(token=tok, token_len=n(n_t), prompt_token=ptok, prompt_token_len=n(n_p), prompt_feat=pfeat, prompt_feat_len=n(2 * n_p),
                                embedding=emb, streaming=False, finalize=True)
        err, mx = _rel(mel.cpu(), ref), (mel.cpu() - ref).abs().max().item()
        _record(lib, "flow_inference_u10_%s_rel_l2" % precision, err)
        _record(lib, "flow_inference_u10_%s_max_abs" % precision, mx)
        _record(lib, "flow_inference_u10_ref_std", ref.std().item())
        assert mel.shape == ref.shape == (1, 80, 2 * n_t) and err < bound_l2 and mx < bound_max, (precision, err, mx)
        mels[precision] = mel
    # SURVEY.md section 8c, second half of the bf16 tolerance at FULL size: waveform SNR >= 30 dB through HiFT given an identical harmonic
    # source.  Reference side entirely on the CPU oracle (fp32 flow mel -> oracle source -> oracle decode); device side = the bf16-mode mel
    # through the device HiFT decoder with that same source.  The fp32-mode mel goes through the same comparison (summation order only).
    hc = _cfgs(lib)[2]
    hsd = W.make_hift(hc)
    hift = HiFTGenerator(hsd, hc, lib=lib)
    m = ref.shape[2]
    _, src_ref = OH.inference(hsd, hc, ref, None, None, torch.zeros(1, 480 * m, hc.harmonics + 1))
    w_ref = OH.decode(hsd, hc, ref, src_ref)
    for precision, min_snr in (("fp32", 100.0), ("bf16", 30.0)):                 # measured 114 dB / 52 dB on the MI355X; 30 dB is the stated bf16 tolerance
        w = hift.decode(mels[precision], src_ref).cpu()
        snr = (10 * torch.log10(w_ref.pow(2).sum() / (w_ref - w).pow(2).sum().clamp_min(1e-30))).item()
        _record(lib, "u10_waveform_snr_db_%s_flow_same_source" % precision, snr)
        assert w.shape == w_ref.shape == (1, 480 * m) and snr >= min_snr, (precision, snr)


def test_hift_u10(lib):
    """HiFT on the benchmark's 500 frames: f0 -> source -> decode with the oracle's source fed to both decoders (the source integrates f0
    into a phase of thousands of radians, so it is compared separately at the tolerance of the tiny tests)."""
    hc = _cfgs(lib)[2]
    sd = W.make_hift(hc)
    hift = HiFTGenerator(sd, hc, lib=lib)
    gen = torch.Generator().manual_seed(14)
    m = 11 if lib.emulated else 500
    mel = torch.randn(1, 80, m, generator=gen) * 2 - 5
    noise = torch.zeros(480 * m, hc.harmonics + 1)
    speech, source = hift.inference(mel, None, noise=noise)
    f0_ref = OH.f0_predictor(sd, mel)
    speech_ref, src_ref = OH.inference(sd, hc, mel, None, None, torch.zeros(1, 480 * m, hc.harmonics + 1))
    out = hift.decode(mel, src_ref).cpu()
    ref = OH.decode(sd, hc, mel, src_ref)
    err = _rel(out, ref)
    _record(lib, "hift_decode_500f_rel_l2", err)
    _record(lib, "hift_decode_500f_max_abs", (out - ref).abs().max().item())
    assert out.shape == ref.shape == (1, 480 * m) and err < 8e-6, err       # measured 2.4e-6 on the MI355X
    # the device's OWN chain: f0 predictor -> harmonic phase (a running sum over 240 000 samples, thousands of radians) -> source -> waveform.
    # The source is compared sample by sample (|sin| <= 0.1 x 9 harmonics through tanh: values in (-1, 1)); measured 8.2e-3 max on the MI355X
    # (profiles/r2_fullsize_errors.json: fp32 rounding of the f0 predictor integrated over the utterance) -> bound at 3x.
    src_err = (source.cpu() - src_ref).abs().max().item()
    _record(lib, "hift_source_500f_max_abs", src_err)
    _record(lib, "hift_source_500f_rel_l2", _rel(source.cpu(), src_ref))
    assert source.shape == src_ref.shape and src_err < 2.5e-2, src_err
    f0_dev = hift.f0_predictor(mel).cpu()
    f0_err = (f0_dev - f0_ref).abs().max().item()
    _record(lib, "hift_f0_500f_max_abs_hz", f0_err)
    assert f0_dev.shape == f0_ref.shape and f0_ref.shape[-1] == m and f0_err < (2.5e-3 if not lib.emulated else 1e-3 * max(1.0, f0_ref.abs().max().item())), f0_err   # measured 7.9e-4 Hz
    # ... and the end-to-end waveform of the device chain against the oracle's end-to-end waveform (bounded by the source error above)
    e2e = _rel(speech.cpu(), speech_ref)
    snr = (10 * torch.log10(speech_ref.pow(2).sum() / (speech_ref - speech.cpu()).pow(2).sum().clamp_min(1e-30))).item()
    _record(lib, "hift_e2e_500f_rel_l2", e2e)
    _record(lib, "hift_e2e_500f_snr_db", snr)
    assert speech.shape == speech_ref.shape == (1, 480 * m) and torch.isfinite(speech).all() and snr >= E2E_MIN_SNR_DB, (e2e, snr)


# ------------------------------------------------------------------------------------------------------------------------------------
# Fun-CosyVoice3-0.5B (BASELINE.json configs[4]) at its real dimensions: DiT with 22 blocks of width 1024,
# 16 heads; CosyVoice3LM (6561 + 200 head rows); the causal HiFT generator.
# ------------------------------------------------------------------------------------------------------------------------------------
def _cv3_cfgs(lib):
    from cosyvoice_amd import configs as CF
    if lib.emulated:
        return CF.tiny_cv3_llm(), CF.tiny_cv3_flow(), None
    return CF.cv3_llm(), CF.cv3_flow(), CF.cv3_hift()


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_cv3_dit_estimator_fullsize(lib, precision):
    """DiT.forward (flow/DiT/dit.py:145-176) as solve_euler calls it: CFG batch 2, T = 674, both mask modes, against the fp32 oracle."""
    from cosyvoice_amd.flow import CausalMaskedDiffWithDiT
    from oracle import dit as OD
    fc = _cv3_cfgs(lib)[1]
    sd = W.make_flow_dit(fc)
    flow = CausalMaskedDiffWithDiT(sd, fc, lib=lib, precision=precision)
    g = torch.Generator().manual_seed(23)
    T = 41 if lib.emulated else 674
    x = torch.randn(2, 80, T, generator=g); mu = torch.randn(2, 80, T, generator=g); cond = torch.randn(2, 80, T, generator=g)
    spk = torch.randn(2, 80, generator=g); t = torch.tensor([0.35, 0.35]); mask = torch.ones(2, 1, T)
    mu[1] = 0; cond[1] = 0; spk[1] = 0
    for streaming in (False, True):
        out = flow.decoder.estimator(x, mask, mu, t, spk, cond, streaming=streaming).cpu()
        ref = OD.estimator(sd, fc, x, mask, mu, t, spk, cond, streaming)
        err = _rel(out, ref)
        _record(lib, "cv3_dit_T674_%s_streaming%d_rel_l2" % (precision, int(streaming)), err)
        _record(lib, "cv3_dit_T674_%s_streaming%d_max_abs" % (precision, int(streaming)), (out - ref).abs().max().item())
        # measured on the MI355X (profiles/r2_fullsize_errors.json): fp32 1.4e-6 (summation order only), bf16 2.8e-3 (operand rounding through 22
        # blocks; max |diff| 1.3e-2) -> bounds at 3x
        assert err < (5e-6 if precision == "fp32" else 9e-3), (precision, streaming, err)


def test_cv3_llm_tokens_fullsize(lib):
    """CosyVoice3LM at the real dimensions: instruct-style request (prompt text with <|endofprompt|>, no speech prompt - what frontend_instruct2
    leaves, cli/frontend.py:209-213), greedy ids against the oracle."""
    from cosyvoice_amd.llm import CosyVoice3LM
    lc = _cv3_cfgs(lib)[0]
    sd = W.make_llm(lc)
    n_gen = 8 if lib.emulated else 48
    lm = CosyVoice3LM(sd, lc, lib=lib, max_len=256, sampling="greedy", decode_chunk=16)
    g = torch.Generator().manual_seed(31)
    text = torch.randint(0, 1000, (1, 6), generator=g, dtype=torch.int32)
    prompt_text = torch.cat([torch.randint(0, 1000, (1, 5), generator=g, dtype=torch.int32), torch.tensor([[lc.endofprompt_id]], dtype=torch.int32)], 1)
    e0 = torch.zeros(1, 0, dtype=torch.int32)
    t = lambda n: torch.tensor([n], dtype=torch.int32)
    ratio = n_gen / 6
    got = list(lm.inference(text=text, text_len=t(6), prompt_text=prompt_text, prompt_text_len=t(6), prompt_speech_token=e0, prompt_speech_token_len=t(0),
                            max_token_text_ratio=ratio, min_token_text_ratio=ratio))
    want = OL.inference(sd, lc, text, prompt_text, e0, max_token_text_ratio=ratio, min_token_text_ratio=ratio)
    assert len(got) == len(want) == n_gen
    div = next((i for i, (a, b) in enumerate(zip(got, want)) if a != b), None)
    _record(lib, "cv3_llm_first_divergence_index", float(n_gen if div is None else div))
    assert div is None, (div, got[div], want[div])


def test_cv3_causal_hift_fullsize(lib):
    """CausalHiFTGenerator at the real dimensions, 200 frames: decoder against the oracle given the oracle's source; a non-final chunk's samples
    equal the one-shot waveform (generator.py:729-746)."""
    from cosyvoice_amd.hift import CausalHiFTGenerator
    hc = _cv3_cfgs(lib)[2]
    if hc is None:
        pytest.skip("hardware only: the tiny causal generator is covered by tests/test_causal_hift.py")
    sd = W.make_hift(hc)
    h = CausalHiFTGenerator(sd, hc, lib=lib)
    gen = torch.Generator().manual_seed(19)
    m = 200
    mel = torch.randn(1, 80, m, generator=gen) * 2 - 5
    noise = torch.zeros(480 * m, hc.harmonics + 1)
    _, src_ref = OH.causal_inference(sd, hc, mel, True, None, noise.unsqueeze(0))
    out = h.decode(mel, src_ref, True).cpu()
    ref = OH.causal_decode(sd, hc, mel, src_ref, True)
    err = _rel(out, ref)
    _record(lib, "cv3_causal_hift_decode_200f_rel_l2", err)
    assert out.shape == ref.shape == (1, 480 * m) and err < 1.6e-5, err      # measured 5.2e-6 on the MI355X
    full, _ = h.inference(mel, True)
    part, _ = h.inference(mel[:, :, :108], False)
    assert part.shape[1] == 480 * 100 and torch.equal(part.cpu(), full.cpu()[:, : part.shape[1]])
    # The fp32 OPTION of the f0 predictor (f0_float64=False; the default since round 4 is the reference's float64, hifigan/generator.py:716-717).
    # Bounded here against the FLOAT64 oracle at the benchmark's 500 frames: f0 itself, then what the f0 error
    # becomes once integrated into the harmonic phase (the device's own source against the float64-f0 oracle's source), then the waveform.
    h = CausalHiFTGenerator(sd, hc, lib=lib, f0_float64=False)
    m5 = 500
    mel5 = torch.randn(1, 80, m5, generator=gen) * 2 - 5
    noise5 = torch.zeros(480 * m5, hc.harmonics + 1)
    f0_64 = OH.causal_f0_predictor(sd, mel5, True, torch.float64)
    f0_dev = h.f0(mel5, True).cpu()
    f0_err = (f0_dev - f0_64).abs().max().item()
    _record(lib, "cv3_f0_fp32_vs_float64_500f_max_abs_hz", f0_err)
    _record(lib, "cv3_f0_500f_max_hz", f0_64.abs().max().item())
    assert f0_dev.shape == f0_64.shape and f0_err < 2.5e-3, f0_err                         # measured 8.3e-4 Hz on f0 values up to 303 Hz
    speech64, src64 = OH.causal_inference(sd, hc, mel5, True, None, noise5.unsqueeze(0), f0_dtype=torch.float64)
    speech_dev, src_dev = h.inference(mel5, True, noise=noise5)
    src_err = (src_dev.cpu() - src64).abs().max().item()
    snr = (10 * torch.log10(speech64.pow(2).sum() / (speech64 - speech_dev.cpu()).pow(2).sum().clamp_min(1e-30))).item()
    _record(lib, "cv3_source_fp32f0_vs_float64f0_500f_max_abs", src_err)
    _record(lib, "cv3_e2e_fp32f0_vs_float64f0_500f_snr_db", snr)
    assert src_dev.shape == src64.shape and src_err < 9e-3, src_err            # measured 2.8e-3
    assert speech_dev.shape == speech64.shape and snr >= E2E_MIN_SNR_DB, snr
