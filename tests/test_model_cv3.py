"""CosyVoice3Model (CausalMaskedDiffWithDiT + CausalHiFTGenerator) end to end.  Split from test_model.py (xdist balance)."""
import dataclasses

import numpy as np
import pytest
import torch

from cosyvoice_amd import synthetic as W

def test_cosyvoice3_model_matches_reference_golden(lib):
    """a17: CosyVoice3Model.tts / token2wav on the device (CausalMaskedDiffWithDiT + CausalHiFTGenerator, accumulating mel cache, speech offsets,
    silent-token filter) against the REAL cosyvoice.cli.model.CosyVoice3Model driving the real tiny modules with the same scripted tokens."""
    import os
    from cosyvoice_amd.flow import CausalMaskedDiffWithDiT
    from cosyvoice_amd.hift import CausalHiFTGenerator
    from cosyvoice_amd.model import CosyVoice3Model
    g = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(os.path.dirname(__file__), "golden", "model_cv3_tiny.npz")).items()}
    lc, _, hc0 = W.tiny()
    fc, hc = W.tiny_cv3_flow(), dataclasses.replace(hc0, causal=True)
    u = W.synthetic_utterance(lc, fc, n_prompt_tok=8, n_prompt_text=4, n_text=2, seed=21)
    tokens = g["tokens"].tolist()
    m = CosyVoice3Model(None, CausalMaskedDiffWithDiT(W.make_flow_dit(fc), fc, lib=lib), CausalHiFTGenerator(W.make_hift(hc), hc, lib=lib), lib=lib)
    assert m.silent_tokens == [1, 2, 28, 29, 55, 248, 494, 2241, 2242, 2322, 2323]

    class ScriptedLLM:
        def inference(self, **kw):
            yield from tokens
    m.llm = ScriptedLLM()
    m.token_hop_len, m.token_max_hop_len = 5, 20
    inf = m.hift.inference
    m.hift.inference = lambda speech_feat, finalize=True: inf(speech_feat, finalize, noise=torch.zeros(speech_feat.shape[2] * 480, 9))
    for key, stream in (("offline", False), ("stream", True)):
        outs = [o["tts_speech"] for o in m.tts(text=u["text"], flow_embedding=u["flow_embedding"], llm_embedding=u["llm_embedding"], prompt_text=u["prompt_text"],
                                               llm_prompt_speech_token=u["llm_prompt_speech_token"], flow_prompt_speech_token=u["flow_prompt_speech_token"],
                                               prompt_speech_feat=u["prompt_speech_feat"], stream=stream)]
        assert [o.shape[1] for o in outs] == g[key + "_n"].tolist()
        torch.testing.assert_close(torch.cat(outs, 1), g[key], rtol=0, atol=5e-3)
        assert not m.hift_cache_dict and not m.tts_speech_token_dict
