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


def test_cosyvoice3_fp16_mode(lib):
    """cli/model.py:403-447: CosyVoice3Model(fp16=True) runs flow and vocoder under autocast.  from_state_dicts(fp16=True) selects the bf16-MFMA flow AND the
    vocoder's reduced-products mode (HiFT option "terms" = 3) on every lane; the scripted request of the golden comes out finite with the real class's chunk length
    and not bit-equal to the fp32 golden (the waveform bounds of the two modes are asserted with a fixed source in tests/test_flow.py / tests/test_causal_hift.py -
    here the source is recomputed from a bf16-mode mel, see test_model_load.py::test_fp16_flag_selects_bf16_flow)."""
    import os
    from cosyvoice_amd.model import CosyVoice3Model
    g = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(os.path.dirname(__file__), "golden", "model_cv3_tiny.npz")).items()}
    lc, _, hc0 = W.tiny()
    llc = W.tiny_cv3_llm()
    fc, hc = W.tiny_cv3_flow(), dataclasses.replace(hc0, causal=True)
    u = W.synthetic_utterance(lc, fc, n_prompt_tok=8, n_prompt_text=4, n_text=2, seed=21)
    m = CosyVoice3Model.from_state_dicts(W.make_llm(llc), W.make_flow_dit(fc), W.make_hift(hc), (llc, fc, hc), lib=lib, fp16=True, max_len=160, sampling="greedy")
    assert m.fp16 and m.flow.precision == "bf16" and m.hift.terms == 3
    m.set_lanes(2)
    assert all(lane.hift.terms == 3 and lane.flow.precision == "bf16" for lane in list(m._lane_q.queue))
    m.set_lanes(1)
    tokens = g["tokens"].tolist()

    class ScriptedLLM:
        def inference(self, **kw):
            yield from tokens
    m.llm = ScriptedLLM()
    inf = m.hift.inference
    m.hift.inference = lambda speech_feat, finalize=True: inf(speech_feat, finalize, noise=torch.zeros(speech_feat.shape[2] * 480, 9))
    outs = [o["tts_speech"].cpu() for o in m.tts(text=u["text"], flow_embedding=u["flow_embedding"], llm_embedding=u["llm_embedding"], prompt_text=u["prompt_text"],
                                                 llm_prompt_speech_token=u["llm_prompt_speech_token"], flow_prompt_speech_token=u["flow_prompt_speech_token"],
                                                 prompt_speech_feat=u["prompt_speech_feat"], stream=False)]
    assert [o.shape[1] for o in outs] == g["offline_n"].tolist()
    wav = torch.cat(outs, 1)
    assert torch.isfinite(wav).all() and not torch.equal(wav, g["offline"]) and not m.hift_cache_dict
