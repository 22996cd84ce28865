"""Boundary B1, batched paths of CosyVoice2Model, second half (padded flow passes of tts_batch, token2wav_batch of the serving scheduler).  Split from
test_model.py so that the CPU suite's files balance over the pytest-xdist workers (the emulator runs a whole vocoder per request)."""
import dataclasses

import numpy as np
import pytest
import torch

from cosyvoice_amd.model import CosyVoice2Model
from cosyvoice_amd import synthetic as W
from test_model import setup  # noqa: F401  (module-scoped fixture: tiny configs, seeded state dicts, one utterance)


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
