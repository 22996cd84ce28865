"""CosyVoice3Model (CausalMaskedDiffWithDiT + CausalHiFTGenerator) end to end.  Split from test_model.py (xdist balance)."""
import dataclasses

import numpy as np
import pytest
import torch

from cosyvoice_amd import synthetic as W
# (split from test_model_cv3.py so that the CPU suite's files balance over the pytest-xdist workers)


def test_cosyvoice3_tts_batch_shares_one_flow_pass(lib):
    """CosyVoice3Model.tts_batch: finished sequences of similar length go through the DiT flow in ONE padded pass (cv_flow_inference_ragged, estimator
    batch rows 2 x utterances), then through the causal HiFT one by one; every waveform equals tts() of that request alone bit for bit."""
    from cosyvoice_amd.flow import CausalMaskedDiffWithDiT
    from cosyvoice_amd.hift import CausalHiFTGenerator
    from cosyvoice_amd.model import CosyVoice3Model
    lc, _, hc0 = W.tiny()
    fc, hc = dataclasses.replace(W.tiny_cv3_flow(), n_timesteps=1), dataclasses.replace(hc0, causal=True)
    m = CosyVoice3Model(None, CausalMaskedDiffWithDiT(W.make_flow_dit(fc), fc, lib=lib), CausalHiFTGenerator(W.make_hift(hc), hc, lib=lib), lib=lib)
    g = torch.Generator().manual_seed(31)
    scripts = [torch.randint(3, fc.vocab, (6,), generator=g).tolist() for _ in range(3)] + [torch.randint(3, fc.vocab, (5,), generator=g).tolist()]
    us = [W.synthetic_utterance(lc, fc, n_prompt_tok=6, n_prompt_text=2, n_text=2, seed=90 + i) for i in range(4)]
    keys = ("text", "flow_embedding", "llm_embedding", "prompt_text", "llm_prompt_speech_token", "flow_prompt_speech_token", "prompt_speech_feat")
    reqs = [{k: x[k] for k in keys} for x in us]

    class ScriptedLLM:                                            # request i is recognised by its text tensor
        def _which(self, text):
            return next(i for i, r in enumerate(reqs) if torch.equal(r["text"], text))

        def inference_batch(self, rs):
            return [list(scripts[self._which(r["text"])]) for r in rs]

        def inference(self, **kw):
            yield from scripts[self._which(kw["text"])]
    m.llm = ScriptedLLM()
    calls = []
    fb = m.flow.inference_batch
    m.flow.inference_batch = lambda items, **kw: (calls.append(len(items)), fb(items, **kw))[1]
    got = m.tts_batch(reqs)
    assert calls == [4]                                           # one padded pass: the 5-token request is within flow_pad of the three 6-token ones
    alone = [next(iter(m.tts(**r, stream=False)))["tts_speech"] for r in reqs]
    for a, b in zip(alone, got):
        assert torch.equal(a, b["tts_speech"])
    assert not torch.equal(alone[0], alone[1]) and not m.hift_cache_dict
