"""Boundary B1, batched and multi-lane paths of CosyVoice2Model (token2wav lanes, tts_batch with shared / padded flow passes).  Split from
test_model.py so that the CPU suite's files balance over the pytest-xdist workers (the emulator runs a whole vocoder per request)."""
import dataclasses

import numpy as np
import pytest
import torch

from cosyvoice_amd.model import CosyVoice2Model
from cosyvoice_amd import synthetic as W
from test_model import setup  # noqa: F401  (module-scoped fixture: tiny configs, seeded state dicts, one utterance)


def test_token2wav_lanes(lib, setup):
    """set_lanes(n): token2wav calls of different requests run concurrently on cloned flow / HiFT handles (same weights, own workspaces,
    one HIP stream each).  Every waveform - harmonic-source noise included (its RNG key comes from the request's tokens) - must equal the
    single-lane result bit for bit, whatever lane served it and in whatever order."""
    cfgs, sds, u = setup
    lc, fc, hc = cfgs
    fc1 = dataclasses.replace(fc, n_timesteps=1)
    m = CosyVoice2Model.from_state_dicts(sds[0], sds[1], sds[2], (lc, fc1, hc), lib=lib, max_len=160, sampling="greedy")
    inf_b = m.llm.inference_batch
    m.llm.inference_batch = lambda reqs: inf_b(reqs, max_token_text_ratio=4, min_token_text_ratio=2)
    us = [W.synthetic_utterance(lc, fc, n_prompt_tok=5 + i, n_prompt_text=2, n_text=1 + i % 2, seed=60 + i) for i in range(3 if not lib.emulated else 2)]
    keys = ("text", "flow_embedding", "llm_embedding", "prompt_text", "llm_prompt_speech_token", "flow_prompt_speech_token", "prompt_speech_feat")
    reqs = [{k: x[k] for k in keys} for x in us]
    one = m.tts_batch(reqs)
    m.set_lanes(2)
    assert m.n_lanes == 2 and m._lane_q.qsize() == 2
    two = m.tts_batch(reqs)
    rev = m.tts_batch(reqs[::-1])[::-1] if not lib.emulated else two      # (the emulator run is kept short)
    for a, b, c in zip(one, two, rev):
        assert a["tts_speech"].abs().max() > 0
        assert torch.equal(a["tts_speech"], b["tts_speech"]) and torch.equal(a["tts_speech"], c["tts_speech"])
    assert not m.hift_cache_dict and m._lane_q.qsize() == 2


def test_tts_batch_shares_one_flow_pass_between_equal_shapes(lib, setup):
    """tts_batch groups finished sequences of equal shape (token count, prompt tokens, prompt frames) into ONE flow pass
    (CausalMaskedDiffWithXvec.inference_batch, cv_flow_inference_batch); every waveform must equal tts() of that request alone bit for bit."""
    cfgs, sds, u = setup
    lc, fc, hc = cfgs
    fc1 = dataclasses.replace(fc, n_timesteps=1)
    m = CosyVoice2Model.from_state_dicts(sds[0], sds[1], sds[2], (lc, fc1, hc), lib=lib, max_len=160, sampling="greedy")
    inf_b, inf_1 = m.llm.inference_batch, m.llm.inference
    m.llm.inference_batch = lambda reqs: inf_b(reqs, max_token_text_ratio=3, min_token_text_ratio=3)      # 6 tokens each: equal shapes
    m.llm.inference = lambda **kw: inf_1(**{**kw, "max_token_text_ratio": 3, "min_token_text_ratio": 3})
    us = [W.synthetic_utterance(lc, fc, n_prompt_tok=6, n_prompt_text=2, n_text=2, seed=80 + i) for i in range(3)]
    keys = ("text", "flow_embedding", "llm_embedding", "prompt_text", "llm_prompt_speech_token", "flow_prompt_speech_token", "prompt_speech_feat")
    reqs = [{k: x[k] for k in keys} for x in us]
    calls = []
    fb = m.flow.inference_batch
    m.flow.inference_batch = lambda items, **kw: (calls.append(len(items)), fb(items, **kw))[1]
    got = m.tts_batch(reqs)
    lens = [g["tts_speech"].shape[1] for g in got]
    assert sum(calls) >= 2 and sorted(calls) == sorted(n for n in (lens.count(v) for v in set(lens)) if n > 1)      # equal shapes shared a pass (a stop id other than eos may end a sequence early)
    alone = [next(iter(m.tts(**r, stream=False)))["tts_speech"] for r in reqs]
    for a, g in zip(alone, got):
        assert torch.equal(a, g["tts_speech"])
    assert not torch.equal(alone[0], alone[1]) and not m.hift_cache_dict
