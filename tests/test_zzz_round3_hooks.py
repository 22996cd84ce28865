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
    # the device loop still works on the same handle afterwards
    assert list(lm.inference(**_kw(u), max_token_text_ratio=3, min_token_text_ratio=2)) == want


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
    assert (h.f0(g["mel"], True).cpu() - g["f0"]).abs().max() <= (CausalHiFTGenerator(sd, cfg, lib=lib).f0(g["mel"], True).cpu() - g["f0"]).abs().max()
    want = OH.causal_f0_predictor(sd, g["mel"][:, :, :13], False, torch.float64).float()
    torch.testing.assert_close(h.f0(g["mel"][:, :, :13], False).cpu(), want.reshape(1, -1), rtol=2e-7, atol=1e-5)
    assert h.clone().f0_float64
    speech, source = h.inference(g["mel"], True, noise=g["noise"])             # the whole chain on the float64 f0
    torch.testing.assert_close(source.cpu(), g["source"], rtol=0, atol=5e-3)
    c2 = W.tiny()[2]
    sd2 = W.make_hift(c2)
    mel = torch.randn(1, 80, 21, generator=torch.Generator().manual_seed(5)) * 2 - 5
    from cosyvoice_amd.hift import HiFTGenerator
    ref64 = OH.f0_predictor({k: v.double() for k, v in sd2.items()}, mel.double()).float()
    torch.testing.assert_close(HiFTGenerator(sd2, c2, lib=lib, f0_float64=True).f0_predictor(mel).cpu(), ref64.reshape(1, -1), rtol=2e-7, atol=1e-5)
