"""CosyVoice3Model (CausalMaskedDiffWithDiT + CausalHiFTGenerator) end to end.  Split from test_model.py (xdist balance)."""
import dataclasses

import numpy as np
import pytest
import torch

from cosyvoice_amd import synthetic as W
# (split from test_model_cv3.py so that the CPU suite's files balance over the pytest-xdist workers)


def test_cosyvoice3_silent_token_runs_filtered_on_every_path(lib):
    """the silent / breath-token rule of llm_job (cli/model.py:122-128: a token of silent_tokens is dropped once more than 5 came in a
    row) must also hold where the tokens do not pass through llm_job - tts_batch, tts_queue and the serving scheduler - or their audio differs
    from tts() whenever the LM emits a run of silence.  Scripts with runs of 8 and 7 silent ids; every path must equal tts() bit for bit, and
    tts() must have vocoded the FILTERED sequence."""
    from cosyvoice_amd.flow import CausalMaskedDiffWithDiT
    from cosyvoice_amd.hift import CausalHiFTGenerator
    from cosyvoice_amd.model import CosyVoice3Model, SilentTokenFilter
    from cosyvoice_amd.serving import StreamScheduler
    lc, _, hc0 = W.tiny()
    fc, hc = dataclasses.replace(W.tiny_cv3_flow(), n_timesteps=1), dataclasses.replace(hc0, causal=True)
    m = CosyVoice3Model(None, CausalMaskedDiffWithDiT(W.make_flow_dit(fc), fc, lib=lib), CausalHiFTGenerator(W.make_hift(hc), hc, lib=lib), lib=lib)
    sil = [t for t in m.silent_tokens if t < fc.vocab]
    assert len(sil) >= 2
    a, b = sil[0], sil[1]
    scripts = [[40, a, a, b, a, b, a, a, b, 41, 42], [43, b, b, b, b, b, b, b, 44, a, 45]]
    want = [SilentTokenFilter(m.silent_tokens)(s) for s in scripts]
    assert [len(w) for w in want] == [8, 9]
    us = [W.synthetic_utterance(lc, fc, n_prompt_tok=6, n_prompt_text=2, n_text=2, seed=190 + i) for i in range(2)]
    keys = ("text", "flow_embedding", "llm_embedding", "prompt_text", "llm_prompt_speech_token", "flow_prompt_speech_token", "prompt_speech_feat")
    reqs = [{k: x[k] for k in keys} for x in us]

    class ScriptedLLM:
        def _which(self, text):
            return next(i for i, r in enumerate(reqs) if torch.equal(r["text"], text))

        def inference_batch(self, rs):
            return [list(scripts[self._which(r["text"])]) for r in rs]

        def inference_queue(self, rs, slots=8):
            for i, r in enumerate(rs):
                yield i, list(scripts[self._which(r["text"])])

        def inference(self, **kw):
            yield from scripts[self._which(kw["text"])]

        def serve_stream(self, source, on_tokens, slots=8, step_chunk=3, **kw):      # tokens arrive 3 at a time: the runs straddle the chunks
            while True:
                item = source.get()
                if item is None:
                    return
                key, r = item
                s = scripts[self._which(r["text"])]
                for k in range(0, len(s), 3):
                    on_tokens(key, s[k:k + 3], k + 3 >= len(s), None)
    m.llm = ScriptedLLM()
    seen = []
    t2w = m.token2wav
    m.token2wav = lambda **kw: (seen.append(kw["token"].flatten().tolist()), t2w(**kw))[1]
    alone = [next(iter(m.tts(**r, stream=False)))["tts_speech"] for r in reqs]
    assert seen == want
    m.token2wav = t2w
    for got in (m.tts_batch(reqs), [o for _, o in sorted(m.tts_queue(reqs, slots=2), key=lambda x: x[0])]):
        for x, y in zip(alone, got):
            assert torch.equal(x, y["tts_speech"])
    sch = StreamScheduler(m, slots=2, step_chunk=3)
    try:
        for x, r in zip(alone, reqs):
            outs = [o["tts_speech"] for o in sch.submit(stream=False, **r)]
            assert len(outs) == 1 and torch.equal(outs[0], x)
    finally:
        sch.shutdown()
    assert not m.hift_cache_dict
