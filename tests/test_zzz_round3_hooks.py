"""Round-3 additions to established stages that reach the MI355X for the first time at the end of the round: the B2 finer hook (forward_one_step) and the float64
f0 predictor option.  They live here - after every established GPU test in file order - so that a first-run failure cannot hide the rest of the suite under -x."""
import os

import numpy as np
import pytest
import torch

from cosyvoice_amd import synthetic as W
from cosyvoice_amd.hift import CausalHiFTGenerator
from cosyvoice_amd.llm import Qwen2LM
from oracle import hift as OH
from oracle import llm as OL

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def tiny_sd():
    cfg = W.tiny()[0]
    return cfg, W.make_llm(cfg)


@pytest.fixture(scope="module")
def tiny():
    import dataclasses
    cfg = dataclasses.replace(W.tiny()[2], causal=True)
    return cfg, W.make_hift(cfg)


def _gold():
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(G, "causal_hift_tiny.npz")).items()}


def _utt(cfg, n_text=6, n_prompt_text=5, n_prompt_tok=11, seed=1986):
    return W.synthetic_utterance(cfg, W.tiny()[1], n_prompt_tok=n_prompt_tok, n_prompt_text=n_prompt_text, n_text=n_text, seed=seed)


def _kw(u):
    t = lambda n: torch.tensor([n], dtype=torch.int32)
    return dict(text=u["text"], text_len=t(u["text"].shape[1]), prompt_text=u["prompt_text"], prompt_text_len=t(u["prompt_text"].shape[1]),
                prompt_speech_token=u["llm_prompt_speech_token"], prompt_speech_token_len=t(u["llm_prompt_speech_token"].shape[1]),
                embedding=u["llm_embedding"])


def test_reference_decode_loop_on_forward_one_step(lib, tiny_sd):
    """Boundary B2, the finer hook: the reference's OWN decode loop (llm/llm.py:535-549: llm.forward_one_step -> llm_decoder -> log_softmax -> sampling ->
    speech_embedding) written out here over Qwen2LM.llm / llm_decoder / speech_embedding gives the oracle's greedy tokens, and the per-step log-probabilities
    of the oracle's trace; a second sequence on the same handle invalidates the first one's cache object."""
    cfg, sd = tiny_sd
    u = _utt(cfg, seed=7)
    lm = Qwen2LM(sd, cfg, lib=lib, max_len=128, sampling="greedy")
    trace = {}
    want = OL.inference(sd, cfg, u["text"], u["prompt_text"], u["llm_prompt_speech_token"], max_token_text_ratio=3, min_token_text_ratio=2, trace=trace)
    n_text = u["text"].shape[1]
    min_len, max_len = 2 * n_text, 3 * n_text
    lm_input = lm.build_lm_input(u["text"], u["prompt_text"], u["llm_prompt_speech_token"]).unsqueeze(0)
    out, cache, first_cache = [], None, None
    for i in range(max_len):
        masks = torch.tril(torch.ones(1, lm_input.shape[1], lm_input.shape[1], dtype=torch.bool))                  # what the reference passes (ignored here)
        y_pred, cache = lm.llm.forward_one_step(lm_input, masks=masks, cache=cache)
        first_cache = first_cache or cache
        logp = lm.llm_decoder(y_pred[:, -1]).log_softmax(dim=-1).cpu()
        torch.testing.assert_close(logp.reshape(-1), trace["logp"][i], rtol=1e-4, atol=1e-4)
        scores = logp.reshape(-1).clone()
        if i < min_len:
            scores[cfg.speech_token_size] = -float("inf")
        top = int(scores.argmax())
        if top >= cfg.speech_token_size:
            break
        out.append(top)
        lm_input = lm.speech_embedding(torch.tensor([[top]]))
        assert lm_input.shape == (1, 1, cfg.hidden)
    assert out == want and len(out) >= 12
    _, other = lm.llm.forward_one_step(lm.speech_embedding(torch.tensor([[1, 2, 3]])), cache=None)                      # a new sequence on the handle
    assert other != first_cache
    with pytest.raises(ValueError):
        lm.llm.forward_one_step(lm_input, cache=first_cache)
    # the device loop still works on the same handle afterwards - and it takes the KV cache too: the live forward_one_step cache goes stale (only a
    # replacing forward_one_step sequence used to be noticed; an interleaved inference() silently left the old cache object appending onto a foreign sequence)
    assert list(lm.inference(**_kw(u), max_token_text_ratio=3, min_token_text_ratio=2)) == want
    with pytest.raises(ValueError):
        lm.llm.forward_one_step(lm.speech_embedding(torch.tensor([[4]])), cache=other)


def test_f0_float64_option_matches_the_reference_mode(lib, tiny):
    """`f0_float64=True`: the predictor with every sum in double, like the reference (generator.py:716-717) - the golden f0 of the REAL class (made in float64,
    returned in fp32) is met to fp32 rounding instead of the fp32 mode's 2e-3; the non-final chunk (3 look-ahead frames) against the float64 oracle; and the
    non-causal HiFTGenerator's predictor takes the same option."""
    from oracle import hift as OH
    cfg, sd = tiny
    g = _gold()
    h = CausalHiFTGenerator(sd, cfg, lib=lib, f0_float64=True)
    f0 = h.f0(g["mel"], True).cpu()
    torch.testing.assert_close(f0, g["f0"], rtol=2e-7, atol=1e-5)
    assert (h.f0(g["mel"], True).cpu() - g["f0"]).abs().max() <= (CausalHiFTGenerator(sd, cfg, lib=lib, f0_float64=False).f0(g["mel"], True).cpu() - g["f0"]).abs().max()
    want = OH.causal_f0_predictor(sd, g["mel"][:, :, :13], False, torch.float64).float()
    torch.testing.assert_close(h.f0(g["mel"][:, :, :13], False).cpu(), want.reshape(1, -1), rtol=2e-7, atol=1e-5)
    assert h.clone().f0_float64 and CausalHiFTGenerator(sd, cfg, lib=lib).f0_float64        # the reference's arithmetic is the default since round 4
    speech, source = h.inference(g["mel"], True, noise=g["noise"])             # the whole chain on the float64 f0
    torch.testing.assert_close(source.cpu(), g["source"], rtol=0, atol=5e-3)
    c2 = W.tiny()[2]
    sd2 = W.make_hift(c2)
    mel = torch.randn(1, 80, 21, generator=torch.Generator().manual_seed(5)) * 2 - 5
    from cosyvoice_amd.hift import HiFTGenerator
    ref64 = OH.f0_predictor({k: v.double() for k, v in sd2.items()}, mel.double()).float()
    torch.testing.assert_close(HiFTGenerator(sd2, c2, lib=lib, f0_float64=True).f0_predictor(mel).cpu(), ref64.reshape(1, -1), rtol=2e-7, atol=1e-5)


def test_ras_sampling_teacher_forced_at_full_size(lib):
    """A NON-greedy decode of the benchmark utterance at the real CosyVoice2-0.5B dimensions (under the emulator: tiny dims): repetition-aware sampling on the
    device with injected uniform variates, 250 sampled tokens - a sequence that does not fall into the short loop greedy decoding of random weights ends in.
    Checked teacher-forced: the oracle scores the device's own sequence in ONE causal pass, and its sampler (oracle/sampling.py, the restated ras_sampling of
    utils/common.py:138-167) replays every decision with the same variates and the same history.  A step may differ from the device's token only where the decision
    is borderline (moving the variates or top_p by 2e-4 reproduces the device's choice: a CDF / nucleus boundary within fp32 summation noise)."""
    from oracle import sampling as OS
    if lib.emulated:
        lc, fc, _ = W.tiny()
        u, n_gen, max_kv = W.synthetic_utterance(lc, fc, n_prompt_tok=9, n_prompt_text=3, n_text=4, seed=3), 28, 128
    else:
        lc, fc, _ = W.cv2()
        u, n_gen, max_kv = W.synthetic_utterance(lc, fc, n_prompt_tok=87, n_prompt_text=12, n_text=30), 250, 1024
    sd = W.make_llm(lc)
    us = np.random.default_rng(11).random(2 * n_gen + 4).astype(np.float32)
    lm = Qwen2LM(sd, lc, lib=lib, max_len=max_kv, sampling="ras", decode_chunk=64)
    lm.set_uniforms(us)
    n_text = u["text"].shape[1]
    ratio = n_gen / n_text
    got = list(lm.inference(**_kw(u), max_token_text_ratio=ratio, min_token_text_ratio=ratio))
    n = len(got)
    stop = [lc.speech_token_size + i for i in range(lc.n_special)]
    assert 0 < n <= n_gen and all(t < lc.speech_token_size for t in got)
    x = torch.cat([OL.build_lm_input(sd, lc, u["text"], u["prompt_text"], u["llm_prompt_speech_token"]), sd["speech_embedding.weight"][torch.tensor(got)]], 0)
    with torch.no_grad():
        y = OL.Qwen2Oracle(sd, lc).forward(x)[-(n + 1):]
        logp_all = torch.nn.functional.linear(y, sd["llm_decoder.weight"], sd.get("llm_decoder.bias")).log_softmax(-1)
    clip = lambda v: float(min(max(v, 0.0), 1.0 - 1e-7))
    exact = borderline = 0
    for i in range(n + (1 if n < n_gen else 0)):                    # (a sequence that ended early ended on a sampled stop id: that decision is replayed too)
        def decide(du, dp):
            scores = logp_all[i].clone()
            scores[lc.speech_token_size] = -float("inf")            # ignore_eos below min_len (= every step of this workload)
            return OS.ras_sampling(scores, got[:i], 25, top_p=0.8 + dp, u=(clip(us[2 * i] + du), clip(us[2 * i + 1] + du)))
        hit = lambda t: (t == got[i]) if i < n else (t in stop)
        if hit(decide(0.0, 0.0)):
            exact += 1
        else:
            assert any(hit(decide(du, dp)) for du in (0.0, -2e-4, 2e-4) for dp in (0.0, -2e-4, 2e-4)), ("step %d: the device's token %s is not a decision of the oracle's sampler"
                                                                                                       % (i, got[i] if i < n else "stop"))
            borderline += 1
    print("ras teacher-forced: %d tokens, %d distinct, %d exact decisions, %d borderline" % (n, len(set(got)), exact, borderline))
    assert borderline <= max(2, n // 50) and len(set(got)) > min(20, n // 2)


def test_hift_inference_batch_is_bit_identical_per_utterance(lib):
    """cv_hift_inference_batch: utterances of equal length through one launch sequence (every convolution with batch = n, the phase walk / STFT / iSTFT per
    utterance) - each row equals cv_hift_inference of that utterance alone with its RNG key, bit for bit; on the MI355X also with injected noise; and a
    single-utterance call afterwards is unchanged.  (Emulator: 2 utterances of 7 frames - it runs a whole vocoder per call.)"""
    from cosyvoice_amd.hift import HiFTGenerator
    cfg = W.tiny()[2]
    h = HiFTGenerator(W.make_hift(cfg), cfg, lib=lib)
    gen = torch.Generator().manual_seed(8)
    n, m = (2, 7) if lib.emulated else (5, 33)
    mels = torch.randn(n, 80, m, generator=gen) * 2 - 5
    seeds = [101 * (i + 1) for i in range(n)]
    alone = [h.inference(mels[i:i + 1], seed=seeds[i]) for i in range(n)]
    speech, source = h.inference_batch(mels, seeds)
    assert speech.shape == (n, m * 480) and source.shape == (n, 1, m * 480)
    for i in range(n):
        assert torch.equal(speech[i:i + 1].cpu(), alone[i][0].cpu()) and torch.equal(source[i:i + 1].cpu(), alone[i][1].cpu()), i
    assert not torch.equal(speech[0], speech[1])
    if not lib.emulated:
        noise = torch.randn(n, m * 480, 9, generator=gen)
        sp_n, so_n = h.inference_batch(mels, seeds, noise=noise)
        one = h.inference(mels[1:2], noise=noise[1])
        assert torch.equal(sp_n[1:2].cpu(), one[0].cpu()) and torch.equal(so_n[1:2].cpu(), one[1].cpu())
        again = h.inference(mels[n - 1:n], seed=seeds[n - 1])
        assert torch.equal(again[0].cpu(), alone[n - 1][0].cpu())


def test_tts_batch_with_batched_vocoding(lib):
    """CosyVoice2Model.hift_batch (opt-in): the equal-length members of a flow group go through HiFT in one launch sequence; every waveform equals the one the
    per-utterance vocoding gives, bit for bit."""
    import dataclasses
    from cosyvoice_amd.model import CosyVoice2Model
    lc, fc, hc = W.tiny()
    fc = dataclasses.replace(fc, chunk=5, n_timesteps=1)
    m = CosyVoice2Model.from_state_dicts(W.make_llm(lc), W.make_flow(fc), W.make_hift(hc), (lc, fc, hc), lib=lib, max_len=160, sampling="greedy")
    inf_b = m.llm.inference_batch
    m.llm.inference_batch = lambda reqs: inf_b(reqs, max_token_text_ratio=3, min_token_text_ratio=3)      # 6 tokens each: equal shapes
    us = [W.synthetic_utterance(lc, fc, n_prompt_tok=6, n_prompt_text=2, n_text=2, seed=80 + i) for i in range(3)]
    keys = ("text", "flow_embedding", "llm_embedding", "prompt_text", "llm_prompt_speech_token", "flow_prompt_speech_token", "prompt_speech_feat")
    reqs = [{k: x[k] for k in keys} for x in us]
    assert m.hift_batch is False                                    # off until measured on the hardware
    one_by_one = m.tts_batch(reqs)
    calls = []
    hb = m.hift.inference_batch
    m.hift.inference_batch = lambda mels, seeds, **kw: (calls.append(mels.shape[0]), hb(mels, seeds, **kw))[1]
    m.hift_batch = True
    batched = m.tts_batch(reqs)
    lens = [g["tts_speech"].shape[1] for g in batched]
    assert calls and sum(calls) == sum(n for n in (lens.count(v) for v in set(lens)) if n > 1)
    for a, b in zip(one_by_one, batched):
        assert torch.equal(a["tts_speech"], b["tts_speech"]) and a["tts_speech"].abs().max() > 0
