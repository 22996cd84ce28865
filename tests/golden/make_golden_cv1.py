"""Golden vectors for the CosyVoice-300M plumbing (cosyvoice_amd/cosyvoice1.py), made by the REAL reference classes.

Run in the build container only:  python tests/golden/make_golden_cv1.py
TransformerLM (+ ConformerEncoder / TransformerEncoder), MaskedDiffWithXvec (+ InterpolateRegulator, ConditionalCFM, ConditionalDecoder over the
restated Matcha blocks of matcha_stub.py), HiFTGenerator at 22.05 kHz (+ ConvRNNF0Predictor) and cli.model.CosyVoiceModel are instantiated
at the dims of configs.tiny_cv1() and loaded (strict=True - which validates every key name and shape of the weight factory) with the
seeded state dicts of cosyvoice_amd.synthetic.make_cv1_llm / make_cv1_flow / make_hift; inputs and outputs are stored, weights are not
(the factory regenerates them).  Random draws come from the global torch RNG, seeded right before each call - the product consumes it
in the same order.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)
import ref_import  # noqa: E402

ref_import.install()
from cosyvoice_amd import synthetic as W  # noqa: E402

# `--k`: the same vectors at configs.tiny_cv1_k() (64-wide heads) for the kernel-backed path, written as cv1k_*.npz
KMODE = "--k" in sys.argv
CFG, HCFG = W.tiny_cv1_k() if KMODE else W.tiny_cv1()
TAG = "cv1k" if KMODE else "cv1"
N_MODEL_TOKENS = 150 if KMODE else 270      # cv1k: one streamed chunk + the final one (the emulator runs every Euler step of both)


def save(name, **arrs):
    out = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()}
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, "%d arrays, %.0f KB" % (len(out), os.path.getsize(os.path.join(HERE, name + ".npz")) / 1024))


ENC = dict(dropout_rate=0.1, positional_dropout_rate=0.1, attention_dropout_rate=0.0, normalize_before=True, pos_enc_layer_type="rel_pos_espnet",
           selfattention_layer_type="rel_selfattn")


def build_llm(sampling):
    from cosyvoice.llm.llm import TransformerLM
    from cosyvoice.transformer.encoder import ConformerEncoder, TransformerEncoder
    c = CFG
    text_encoder = ConformerEncoder(input_size=c.text_enc_in, output_size=c.llm_dim, attention_heads=c.text_heads, linear_units=c.text_ffn, num_blocks=c.text_blocks,
                                    input_layer="linear", use_cnn_module=False, macaron_style=False, use_dynamic_chunk=False, use_dynamic_left_chunk=False,
                                    static_chunk_size=1, **ENC)
    llm = TransformerEncoder(input_size=c.llm_dim, output_size=c.llm_dim, attention_heads=c.llm_heads, linear_units=c.llm_ffn, num_blocks=c.llm_blocks,
                             input_layer="linear_legacy", static_chunk_size=1, **ENC)
    m = TransformerLM(text_encoder_input_size=c.text_enc_in, llm_input_size=c.llm_dim, llm_output_size=c.llm_dim, text_token_size=c.text_vocab,
                      speech_token_size=c.speech_token_size, text_encoder=text_encoder, llm=llm, sampling=sampling, spk_embed_dim=c.spk_dim)
    m.load_state_dict(W.make_cv1_llm(c), strict=True)
    return m.eval()


def golden_llm():
    from cosyvoice.utils.common import ras_sampling
    greedy = lambda scores, decoded, sampling: int(scores.argmax().item())
    g = torch.Generator().manual_seed(3)
    text = torch.randint(0, CFG.text_vocab, (1, 7), generator=g, dtype=torch.int32)
    prompt_text = torch.randint(0, CFG.text_vocab, (1, 4), generator=g, dtype=torch.int32)
    prompt_speech = torch.randint(0, 40, (1, 9), generator=g, dtype=torch.int32)
    emb = torch.randn(1, 16, generator=g)
    t = lambda n: torch.tensor([n], dtype=torch.int32)
    out = {}
    for name, samp, seed in (("greedy", greedy, 0), ("ras", ras_sampling, 7)):
        m = build_llm(samp)
        torch.manual_seed(seed)
        out["tokens_" + name] = np.array(list(m.inference(text=text, text_len=t(7), prompt_text=prompt_text, prompt_text_len=t(4), prompt_speech_token=prompt_speech,
                                                           prompt_speech_token_len=t(9), embedding=emb, max_token_text_ratio=6, min_token_text_ratio=2)))
    # sft-style request: no prompt text / speech (frontend_sft)
    m = build_llm(greedy)
    e0 = torch.zeros(1, 0, dtype=torch.int32)
    out["tokens_sft"] = np.array(list(m.inference(text=text, text_len=t(7), prompt_text=e0, prompt_text_len=t(0), prompt_speech_token=e0,
                                                  prompt_speech_token_len=t(0), embedding=emb, max_token_text_ratio=5, min_token_text_ratio=2)))
    with torch.inference_mode():
        enc, _ = m.encode(m.text_embedding(torch.cat([prompt_text, text], 1)), t(11))
    save(TAG + "_llm", text=text, prompt_text=prompt_text, prompt_speech_token=prompt_speech, embedding=emb, text_encoded=enc[0], **out)


def build_flow():
    from omegaconf import DictConfig
    from cosyvoice.flow.decoder import ConditionalDecoder
    from cosyvoice.flow.flow import MaskedDiffWithXvec
    from cosyvoice.flow.flow_matching import ConditionalCFM
    from cosyvoice.flow.length_regulator import InterpolateRegulator
    from cosyvoice.transformer.encoder import ConformerEncoder
    c = CFG
    enc = ConformerEncoder(output_size=c.flow_dim, attention_heads=c.flow_heads, linear_units=c.flow_ffn, num_blocks=c.flow_blocks, input_layer="linear",
                           input_size=c.flow_dim, use_cnn_module=False, macaron_style=False, **dict(ENC, attention_dropout_rate=0.1))
    est = ConditionalDecoder(in_channels=4 * c.mel, out_channels=c.mel, channels=list(c.est_ch), dropout=0.0, attention_head_dim=c.est_head_dim,
                             n_blocks=c.est_blocks, num_mid_blocks=c.est_mid, num_heads=c.est_heads, act_fn="gelu")
    cfm = ConditionalCFM(in_channels=240, n_spks=1, spk_emb_dim=80,
                         cfm_params=DictConfig({"sigma_min": 1e-6, "solver": "euler", "t_scheduler": "cosine", "training_cfg_rate": 0.2,
                                                "inference_cfg_rate": 0.7, "reg_loss_type": "l1"}), estimator=est)
    flow = MaskedDiffWithXvec(input_size=c.flow_dim, output_size=c.mel, spk_embed_dim=c.spk_dim, vocab_size=c.speech_token_size,
                              input_frame_rate=c.input_frame_rate, encoder=enc,
                              length_regulator=InterpolateRegulator(channels=c.mel, sampling_ratios=[1] * c.regulator_layers), decoder=cfm)
    flow.load_state_dict(W.make_cv1_flow(c), strict=True)
    return flow.eval()


def golden_flow():
    flow = build_flow()
    g = torch.Generator().manual_seed(5)
    prompt_token = torch.randint(0, 40, (1, 12), generator=g, dtype=torch.int32)
    prompt_feat = torch.randn(1, 25, 80, generator=g) * 2 - 5
    emb = torch.randn(1, 16, generator=g)
    t = lambda n: torch.tensor([n], dtype=torch.int32)
    out = {}
    cache = torch.zeros(1, 80, 0, 2)
    for name, n in (("a", 50), ("b", 30)):                      # 50 tokens: head / middle / tail interpolation; 30: one piece; "b" runs on "a"'s flow cache
        token = torch.randint(0, 40, (1, n), generator=g, dtype=torch.int32)
        torch.manual_seed(40 + n)
        feat, cache = flow.inference(token=token, token_len=t(n), prompt_token=prompt_token, prompt_token_len=t(12), prompt_feat=prompt_feat,
                                     prompt_feat_len=t(25), embedding=emb, flow_cache=cache)
        out["token_" + name], out["feat_" + name], out["cache_" + name] = token, feat, cache
    save(TAG + "_flow", prompt_token=prompt_token, prompt_feat=prompt_feat, embedding=emb, **out)


def build_hift():
    from cosyvoice.hifigan.f0_predictor import ConvRNNF0Predictor
    from cosyvoice.hifigan.generator import HiFTGenerator
    c = HCFG
    h = HiFTGenerator(in_channels=c.mel, base_channels=c.base, nb_harmonics=c.harmonics, sampling_rate=c.sr, upsample_rates=list(c.ups),
                      upsample_kernel_sizes=list(c.up_k), source_resblock_kernel_sizes=list(c.src_k),
                      f0_predictor=ConvRNNF0Predictor(num_class=1, in_channels=c.mel, cond_channels=c.f0_ch))
    h.load_state_dict(W.make_hift(c), strict=True)
    return h.eval()


def golden_hift():
    h = build_hift()
    g = torch.Generator().manual_seed(9)
    feat = torch.randn(1, 80, 30, generator=g) * 2 - 5
    torch.manual_seed(77)
    speech, source = h.inference(speech_feat=feat)
    torch.manual_seed(78)
    cs = source[:, :, :1024].clone() * 0.5
    speech2, source2 = h.inference(speech_feat=feat, cache_source=cs)
    with torch.inference_mode():
        f0 = h.f0_predictor(feat)
    save(TAG + "_hift", feat=feat, f0=f0, speech=speech, source=source, cache_source=cs, speech2=speech2, source2=source2)
    return h


def golden_model():
    """cli.model.CosyVoiceModel (streaming with flow cache / mel overlap / HiFT cache, and offline) around the tiny flow + HiFT and a scripted LLM."""
    import cosyvoice.cli.model as M
    flow, hift = build_flow(), build_hift()
    g = torch.Generator().manual_seed(13)
    tokens = torch.randint(0, 40, (N_MODEL_TOKENS,), generator=g).tolist()  # hop 100 + overlap 20: two streamed chunks and a final one (270 tokens)
    prompt_token = torch.randint(0, 40, (1, 10), generator=g, dtype=torch.int32)
    prompt_feat = torch.randn(1, 17, 80, generator=g) * 2 - 5
    emb = torch.randn(1, 16, generator=g)

    class ScriptedLLM:
        def inference(self, **kw):
            for t in tokens:
                yield t

    M.time.sleep = lambda s: None
    out = {}
    for stream in (False, True):
        m = M.CosyVoiceModel(ScriptedLLM(), flow, hift)
        torch.manual_seed(55)
        with torch.inference_mode():
            chunks = [o["tts_speech"] for o in m.tts(text=torch.zeros(1, 3, dtype=torch.int32), flow_embedding=emb, llm_embedding=emb,
                                                    flow_prompt_speech_token=prompt_token, prompt_speech_feat=prompt_feat, stream=stream)]
        key = "stream" if stream else "offline"
        out[key + "_n"] = np.array([c.shape[1] for c in chunks])
        out[key] = torch.cat(chunks, 1)
    out["offline"] = out["offline"][:, :30000]                   # one-shot synthesis has no chunk seams: its head pins it (keeps the fixture small)
    save(TAG + "_model", tokens=np.array(tokens), prompt_token=prompt_token, prompt_feat=prompt_feat,
         embedding=emb, **out)


if __name__ == "__main__":
    which = [a for a in sys.argv[1:] if not a.startswith("--")] or (["llm", "flow", "model"] if KMODE else ["llm", "flow", "hift", "model"])
    for w in which:
        globals()["golden_" + w]()
