"""Shared helpers of the kernel-backed CosyVoice-300M tests (tests/test_zzz_cosyvoice1_hip*.py; TEST INFRASTRUCTURE)."""
import os

import numpy as np
import torch

from cosyvoice_amd import cosyvoice1_hip as CK
from cosyvoice_amd import synthetic as W

HERE = os.path.dirname(os.path.abspath(__file__))
CFG, HCFG = W.tiny_cv1_k()


def gold(name):
    d = np.load(os.path.join(HERE, "golden", name + ".npz"))
    return {k: torch.from_numpy(d[k]) for k in d.files}


t = lambda n: torch.tensor([n], dtype=torch.int32)
greedy = lambda scores, decoded, sampling: int(scores.argmax().item())


def build_flow(lib, estimator="handle", precision="fp32"):
    """estimator: "handle" (the product default: the U-Net inside one library handle) | "operators" (one launch per operator, sequenced in python)."""
    return CK.MaskedDiffWithXvec(W.make_cv1_flow(CFG), enc_heads=CFG.flow_heads, est_heads=CFG.est_heads, input_frame_rate=CFG.input_frame_rate, lib=lib,
                                 estimator=estimator, precision=precision)


def build_hift(lib, rng="host"):
    return CK.HiFTGenerator(W.make_hift(HCFG), HCFG, lib=lib, rng=rng)


def run_model_tts(lib, stream):
    """cli.model.CosyVoiceModel around the kernel-backed flow + HiFT against the golden of the real class: offline, or streaming with the 100-token hop +
    20-token overlap, flow cache, mel-overlap fade and HiFT cache."""
    g = gold("cv1k_model")
    tokens = g["tokens"].tolist()

    class ScriptedLLM:
        def inference(self, **kw):
            yield from tokens

    m = CK.CosyVoiceModel(ScriptedLLM(), build_flow(lib), build_hift(lib))
    torch.manual_seed(55)
    chunks = [o["tts_speech"] for o in m.tts(text=torch.zeros(1, 3, dtype=torch.int32), flow_embedding=g["embedding"], llm_embedding=g["embedding"],
                                            flow_prompt_speech_token=g["prompt_token"], prompt_speech_feat=g["prompt_feat"], stream=stream)]
    key = "stream" if stream else "offline"
    assert [c.shape[1] for c in chunks] == g[key + "_n"].tolist() and all(c.device.type == "cpu" for c in chunks)
    got = torch.cat(chunks, 1)
    want = g[key]
    torch.testing.assert_close(got[:, : want.shape[1]], want, rtol=0, atol=5e-3)
    assert len(chunks) == (2 if stream else 1)
    assert not (m.tts_speech_token_dict or m.llm_end_dict or m.mel_overlap_dict or m.flow_cache_dict or m.hift_cache_dict)
