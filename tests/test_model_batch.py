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


def test_tts_batch_pads_similar_lengths_into_one_flow_pass(lib, setup):
    """tts_batch buckets finished sequences by length: sequences within `flow_pad` of the group's longest share ONE padded flow pass
    (cv_flow_inference_ragged), a much shorter one goes alone; every waveform equals tts() of that request alone bit for bit."""
    cfgs, sds, u = setup
    lc, fc, hc = cfgs
    fc1 = dataclasses.replace(fc, n_timesteps=1)
    m = CosyVoice2Model.from_state_dicts(sds[0], sds[1], sds[2], (lc, fc1, hc), lib=lib, max_len=160, sampling="greedy")
    g = torch.Generator().manual_seed(41)
    lens = [8, 7, 7, 3]
    scripts = [torch.randint(0, fc.vocab, (k,), generator=g).tolist() for k in lens]
    us = [W.synthetic_utterance(lc, fc, n_prompt_tok=6 + (i % 2), n_prompt_text=2, n_text=2, seed=60 + i) for i in range(4)]
    keys = ("text", "flow_embedding", "llm_embedding", "prompt_text", "llm_prompt_speech_token", "flow_prompt_speech_token", "prompt_speech_feat")
    reqs = [{k: x[k] for k in keys} for x in us]
    which = lambda text: next(i for i, r in enumerate(reqs) if torch.equal(r["text"].cpu(), text.cpu()))

    class ScriptedLLM:
        def inference_batch(self, rs):
            return [list(scripts[which(r["text"])]) for r in rs]

        def inference(self, **kw):
            yield from scripts[which(kw["text"])]
    m.llm = ScriptedLLM()
    calls = []
    fb = m.flow.inference_batch
    m.flow.inference_batch = lambda items, **kw: (calls.append(sorted(int(it["token"].shape[1]) for it in items)), fb(items, **kw))[1]
    got = m.tts_batch(reqs)
    assert calls == [[7, 7, 8]]                                # the 3-token request is too short for the group (flow_pad 1.25) and goes alone
    alone = [next(iter(m.tts(**r, stream=False)))["tts_speech"] for r in reqs]
    for a, b, k in zip(alone, got, lens):
        assert a.shape[1] == k * 2 * 480 and torch.equal(a, b["tts_speech"])
    assert not m.hift_cache_dict


def test_token2wav_batch_equals_token2wav_chunk_for_chunk(lib, setup):
    """Round 3 (the serving scheduler's chunk batches): token2wav_batch runs the flow ONCE over several requests' streaming chunks (different
    lengths: the padded pass) and then each request's own HiFT call with its own cache; every chunk of every request - first chunk, a later
    chunk that crosses the mel / source / speech caches and the fade, and the final call - equals token2wav of that request alone bit for bit."""
    cfgs, sds, u = setup
    lc, fc, hc = cfgs
    fc1 = dataclasses.replace(fc, n_timesteps=1, chunk=5)
    m = CosyVoice2Model.from_state_dicts(sds[0], sds[1], sds[2], (lc, fc1, hc), lib=lib, max_len=160, sampling="greedy")
    g = torch.Generator().manual_seed(5)
    us = [W.synthetic_utterance(lc, fc, n_prompt_tok=(3 if lib.emulated else 5) + i, n_prompt_text=2, n_text=1, seed=90 + i) for i in range(2)]
    la = m.flow.pre_lookahead_len
    if lib.emulated:                                                                    # (the emulator run is kept short: two chunks of a few frames)
        toks = [torch.randint(0, fc.vocab, (1, 9 + i), generator=g, dtype=torch.int32) for i in range(2)]
        plan = [(0, 5 + la, False), (5, None, True)]                                    # (token_offset, tokens seen, finalize)
    else:
        toks = [torch.randint(0, fc.vocab, (1, 26 + 2 * i), generator=g, dtype=torch.int32) for i in range(2)]
        plan = [(0, 8 + la, False), (8, 18 + la, False), (18, None, True)]

    def job(i, key, off, n):
        return dict(token=toks[i][:, :n] if n else toks[i], prompt_token=us[i]["flow_prompt_speech_token"], prompt_feat=us[i]["prompt_speech_feat"],
                    embedding=us[i]["flow_embedding"], token_offset=off, uuid=key)
    alone = []
    for i in range(2):
        m.hift_cache_dict["a%d" % i] = None
        alone.append([m.token2wav(stream=not fin, finalize=fin, **job(i, "a%d" % i, off, n)).clone() for off, n, fin in plan])
        m.hift_cache_dict.pop("a%d" % i)
    for i in range(2):
        m.hift_cache_dict["b%d" % i] = None
    calls = []
    fb = m.flow.inference_batch
    m.flow.inference_batch = lambda items, **kw: (calls.append(len(items)), fb(items, **kw))[1]
    for k, (off, n, fin) in enumerate(plan):
        got = m.token2wav_batch([job(i, "b%d" % i, off, n) for i in range(2)], stream=not fin, finalize=fin)
        for i in range(2):
            assert got[i].abs().max() > 0 and torch.equal(got[i], alone[i][k]), (k, i)
    assert calls == [2] * len(plan)
    assert not torch.equal(alone[0][1][:, :4000], alone[1][1][:, :4000])
