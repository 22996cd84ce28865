"""SURVEY.md section 8 row a18 / BASELINE.json configs[0]: the CosyVoice-300M plumbing (cosyvoice_amd/cosyvoice1.py, torch fp32 eager on the CPU)
against golden vectors made by the REAL reference classes (tests/golden/make_golden_cv1.py: TransformerLM, MaskedDiffWithXvec, HiFTGenerator at
22.05 kHz, cli.model.CosyVoiceModel) on the same seeded weights (cosyvoice_amd.synthetic.make_cv1_*) and the same global-RNG seeds.
No GPU, no HIP library: this configuration is CPU plumbing by definition."""
import os

import numpy as np
import pytest
import torch

from cosyvoice_amd import cosyvoice1 as C1
from cosyvoice_amd import synthetic as W

HERE = os.path.dirname(os.path.abspath(__file__))
CFG, HCFG = W.tiny_cv1()


def gold(name):
    d = np.load(os.path.join(HERE, "golden", name + ".npz"))
    return {k: torch.from_numpy(d[k]) for k in d.files}


t = lambda n: torch.tensor([n], dtype=torch.int32)
greedy = lambda scores, decoded, sampling: int(scores.argmax().item())


def build_llm(sampling):
    return C1.TransformerLM(W.make_cv1_llm(CFG), text_heads=CFG.text_heads, llm_heads=CFG.llm_heads, sampling=sampling)


def build_flow():
    return C1.MaskedDiffWithXvec(W.make_cv1_flow(CFG), enc_heads=CFG.flow_heads, est_heads=CFG.est_heads, input_frame_rate=CFG.input_frame_rate)


def build_hift():
    return C1.HiFTGenerator(W.make_hift(HCFG), sampling_rate=HCFG.sr, upsample_rates=HCFG.ups, upsample_kernel_sizes=HCFG.up_k, source_resblock_kernel_sizes=HCFG.src_k)


def test_transformer_lm_tokens_match_reference():
    g = gold("cv1_llm")
    kw = dict(text=g["text"], text_len=t(7), prompt_text=g["prompt_text"], prompt_text_len=t(4), prompt_speech_token=g["prompt_speech_token"],
              prompt_speech_token_len=t(9), embedding=g["embedding"])
    lm = build_llm(greedy)
    # text encoder (causal ConformerEncoder + affine) against the reference's encode()
    sd = lm.sd
    ids = torch.cat([g["prompt_text"], g["text"]], 1).reshape(-1).long()
    enc = C1._linear(C1._P(sd, "text_encoder_affine_layer."), lm.text_encoder.forward(torch.nn.functional.embedding(ids, sd["text_embedding.weight"])))
    torch.testing.assert_close(enc, g["text_encoded"], rtol=1e-4, atol=1e-4)
    assert list(lm.inference(max_token_text_ratio=6, min_token_text_ratio=2, **kw)) == g["tokens_greedy"].tolist()      # stepped forward_chunk + KV cache
    e0 = torch.zeros(1, 0, dtype=torch.int32)
    sft = dict(kw, prompt_text=e0, prompt_text_len=t(0), prompt_speech_token=e0, prompt_speech_token_len=t(0))
    assert list(lm.inference(max_token_text_ratio=5, min_token_text_ratio=2, **sft)) == g["tokens_sft"].tolist()
    # repetition-aware sampling on the global RNG: same seed -> the reference's tokens
    lm = build_llm(C1.ras_sampling)
    torch.manual_seed(7)
    got = list(lm.inference(max_token_text_ratio=6, min_token_text_ratio=2, **kw))
    assert got == g["tokens_ras"].tolist() and 14 <= len(got) <= 42


def test_flow_inference_with_flow_cache_matches_reference():
    g = gold("cv1_flow")
    flow = build_flow()
    cache = torch.zeros(1, 80, 0, 2)
    for name, n in (("a", 50), ("b", 30)):
        torch.manual_seed(40 + n)
        feat, cache = flow.inference(token=g["token_" + name], token_len=t(n), prompt_token=g["prompt_token"], prompt_token_len=t(12), prompt_feat=g["prompt_feat"],
                                     prompt_feat_len=t(25), embedding=g["embedding"], flow_cache=cache)
        assert feat.shape == g["feat_" + name].shape == (1, 80, int(n / 50 * 22050 / 256))
        torch.testing.assert_close(feat, g["feat_" + name], rtol=1e-3, atol=1e-3)
        torch.testing.assert_close(cache, g["cache_" + name], rtol=1e-4, atol=1e-4)


def test_hift_22k_matches_reference():
    g = gold("cv1_hift")
    h = build_hift()
    torch.testing.assert_close(h.f0_predictor(g["feat"]), g["f0"], rtol=1e-4, atol=1e-3)
    assert float((g["f0"] > 10).float().mean()) > 0.2           # voiced and unvoiced frames both occur
    torch.manual_seed(77)
    speech, source = h.inference(speech_feat=g["feat"])
    assert speech.shape == (1, 30 * 256)
    # the harmonic phase is a cumulative sum over 7680 samples: fp32 summation order shows at ~1e-4 in sin(phase)
    torch.testing.assert_close(source, g["source"], rtol=0, atol=2e-3)
    torch.testing.assert_close(speech, g["speech"], rtol=0, atol=5e-3)
    torch.manual_seed(78)
    speech2, source2 = h.inference(speech_feat=g["feat"], cache_source=g["cache_source"])
    torch.testing.assert_close(source2[:, :, :1024], g["cache_source"], rtol=0, atol=0)
    torch.testing.assert_close(speech2, g["speech2"], rtol=0, atol=5e-3)


@pytest.mark.parametrize("stream", [False, True])
def test_cosyvoice_model_tts_matches_reference(stream):
    """cli.model.CosyVoiceModel: offline, and streaming with the 100-token hop + 20-token overlap, flow cache, mel-overlap fade and HiFT cache."""
    g = gold("cv1_model")
    tokens = g["tokens"].tolist()

    class ScriptedLLM:
        def inference(self, **kw):
            yield from tokens

    m = C1.CosyVoiceModel(ScriptedLLM(), build_flow(), build_hift())
    torch.manual_seed(55)
    chunks = [o["tts_speech"] for o in m.tts(text=torch.zeros(1, 3, dtype=torch.int32), flow_embedding=g["embedding"], llm_embedding=g["embedding"],
                                            flow_prompt_speech_token=g["prompt_token"], prompt_speech_feat=g["prompt_feat"], stream=stream)]
    key = "stream" if stream else "offline"
    assert [c.shape[1] for c in chunks] == g[key + "_n"].tolist() and all(c.device.type == "cpu" for c in chunks)
    got = torch.cat(chunks, 1)
    want = g[key]
    torch.testing.assert_close(got[:, : want.shape[1]], want, rtol=0, atol=5e-3)
    assert len(chunks) == (3 if stream else 1)
    assert not (m.tts_speech_token_dict or m.llm_end_dict or m.mel_overlap_dict or m.flow_cache_dict or m.hift_cache_dict)


def test_load_takes_reference_state_dict_files(tmp_path):
    """CosyVoiceModel.load(llm.pt, flow.pt, hift.pt) (cli/model.py:65-73): reference key names, `generator.` prefix stripped from the vocoder's."""
    torch.save(W.make_cv1_llm(CFG), tmp_path / "llm.pt")
    torch.save(W.make_cv1_flow(CFG), tmp_path / "flow.pt")
    torch.save({"generator." + k: v for k, v in W.make_hift(HCFG).items()}, tmp_path / "hift.pt")
    m = C1.CosyVoiceModel()
    m.load(str(tmp_path / "llm.pt"), str(tmp_path / "flow.pt"), str(tmp_path / "hift.pt"), text_heads=CFG.text_heads, llm_heads=CFG.llm_heads,
           enc_heads=CFG.flow_heads, est_heads=CFG.est_heads,
           hift=dict(sampling_rate=HCFG.sr, upsample_rates=HCFG.ups, upsample_kernel_sizes=HCFG.up_k, source_resblock_kernel_sizes=HCFG.src_k))
    m.llm.sampling = greedy
    g = gold("cv1_llm")
    inf = m.llm.inference
    m.llm.inference = lambda **kw: inf(**dict(kw, max_token_text_ratio=6, min_token_text_ratio=6))
    torch.manual_seed(1)
    out = next(iter(m.tts(text=g["text"], flow_embedding=g["embedding"], llm_embedding=g["embedding"], stream=False)))["tts_speech"]   # inference_sft-shaped request
    assert out.shape == (1, int(42 / 50 * 22050 / 256) * 256) and torch.isfinite(out).all() and float(out.abs().max()) > 0


@pytest.mark.gpu
def test_cosyvoice_300m_on_the_gpu_matches_the_cpu_goldens():
    """the same CosyVoice-300M plumbing with its weights on cuda:0 (torch ops through PyTorch-ROCm; random draws stay on
    the host RNG, so the CPU goldens of the real reference apply unchanged)."""
    dev = "cuda"
    mv = lambda sd: {k: v.to(dev) for k, v in sd.items()}
    g = gold("cv1_llm")
    lm = C1.TransformerLM(mv(W.make_cv1_llm(CFG)), text_heads=CFG.text_heads, llm_heads=CFG.llm_heads, sampling=C1.ras_sampling)
    torch.manual_seed(7)
    got = list(lm.inference(text=g["text"], text_len=t(7), prompt_text=g["prompt_text"], prompt_text_len=t(4), prompt_speech_token=g["prompt_speech_token"],
                            prompt_speech_token_len=t(9), embedding=g["embedding"], max_token_text_ratio=6, min_token_text_ratio=2))
    assert got == g["tokens_ras"].tolist()
    g = gold("cv1_model")
    tokens = g["tokens"].tolist()

    class ScriptedLLM:
        def inference(self, **kw):
            yield from tokens

    flow = C1.MaskedDiffWithXvec(mv(W.make_cv1_flow(CFG)), enc_heads=CFG.flow_heads, est_heads=CFG.est_heads, input_frame_rate=CFG.input_frame_rate)
    hift = C1.HiFTGenerator(mv(W.make_hift(HCFG)), sampling_rate=HCFG.sr, upsample_rates=HCFG.ups, upsample_kernel_sizes=HCFG.up_k, source_resblock_kernel_sizes=HCFG.src_k)
    m = C1.CosyVoiceModel(ScriptedLLM(), flow, hift)
    torch.manual_seed(55)
    chunks = [o["tts_speech"] for o in m.tts(text=torch.zeros(1, 3, dtype=torch.int32), flow_embedding=g["embedding"], llm_embedding=g["embedding"],
                                            flow_prompt_speech_token=g["prompt_token"], prompt_speech_feat=g["prompt_feat"], stream=True)]
    assert [c.shape[1] for c in chunks] == g["stream_n"].tolist()
    torch.testing.assert_close(torch.cat(chunks, 1), g["stream"], rtol=0, atol=1e-2)
