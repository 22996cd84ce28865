"""Parity at the REAL CosyVoice2-0.5B dimensions (BASELINE.json configs[1]).  The other parity tests run emulator-sized models, so the
tile shapes, GEMV widths and attention sizes the real model launches were only exercised by bench.py, which checks nothing.  On the
MI355X these tests build the full-size LLM / flow / HiFT (seeded random weights) and compare a bounded piece of each stage with the CPU
oracle (a few seconds of CPU work each); under the CPU emulator the same code runs on the tiny configuration.

Criteria are relative L2 errors.  They could not be calibrated on hardware when this file was written (GPU budget of the round spent),
so they are calibrated with the oracle at full size instead: the fp32 estimator moves by 1.3e-6 (rel. L2) under a 1e-6 input perturbation
(no amplification through the 14 stages), and the oracle's own bf16 mirror sits 6.2e-3 from its fp32 result (max |diff| 1.8e-2 on
outputs of std 0.69).  Bounds: fp32 mode 2e-4 (~100x the expected summation-order noise), bf16 mode 5e-2 (8x the mirror's distance);
a wrong tile index or a mis-sized launch gives O(1).  Greedy ids must match wherever the oracle's own top-2 margin is clear.  The
measured errors are printed (run with -s) to tighten the bounds next round."""
import pytest
import torch

from cosyvoice_amd.flow import CausalMaskedDiffWithXvec
from cosyvoice_amd.hift import HiFTGenerator
from cosyvoice_amd.llm import Qwen2LM
from oracle import flow as OF
from oracle import hift as OH
from oracle import llm as OL
from oracle import weights as W


def _cfgs(lib):
    return W.tiny() if lib.emulated else W.cv2()


def _rel(a, b):
    return ((a - b).norm() / b.norm().clamp_min(1e-12)).item()


def test_llm_fullsize(lib):
    lc = _cfgs(lib)[0]
    sd = W.make_llm(lc)
    u = W.synthetic_utterance(lc, _cfgs(lib)[1], n_prompt_tok=20, n_prompt_text=6, n_text=8, seed=5)
    lm = Qwen2LM(sd, lc, lib=lib, max_len=256, sampling="greedy", decode_chunk=4)
    n_steps = 6
    trace = {}
    want = OL.inference(sd, lc, u["text"], u["prompt_text"], u["llm_prompt_speech_token"], max_token_text_ratio=n_steps / 8, min_token_text_ratio=n_steps / 8,
                        trace=trace)
    lm_input = lm.build_lm_input(u["text"], u["prompt_text"], u["llm_prompt_speech_token"])
    torch.testing.assert_close(lm_input.cpu(), OL.build_lm_input(sd, lc, u["text"], u["prompt_text"], u["llm_prompt_speech_token"]), rtol=0, atol=0)
    lm.prefill(lm_input)
    sp = lm.make_sampling(n_steps, n_steps)
    for i in range(len(want)):
        toks, _ = lm.decode(1, sp)
        logp = lm.last_logits().log_softmax(-1)
        ref = trace["logp"][i]
        err = _rel(logp, ref)
        print("llm step %d: rel L2 of log-probs %.2e" % (i, err))
        assert err < 1e-3, (i, err)
        top2 = torch.topk(ref.masked_fill(torch.arange(ref.numel()) == lc.speech_token_size, -float("inf")), 2).values
        if (top2[0] - top2[1]).item() > 2e-2:                     # not a near-tie in the oracle itself -> the id must match
            assert toks == [want[i]], (i, toks, want[i])
        else:
            break                                                  # past a near-tie the two sequences may legitimately diverge


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_flow_estimator_fullsize(lib, precision):
    fc = _cfgs(lib)[1]
    sd = W.make_flow(fc)
    flow = CausalMaskedDiffWithXvec(sd, fc, lib=lib, precision=precision)
    g = torch.Generator().manual_seed(12)
    T = 41 if lib.emulated else 200                                # 200 frames: every GEMM tile shape of the U10 run, 4 query tiles
    x = torch.randn(2, 80, T, generator=g); mu = torch.randn(2, 80, T, generator=g); cond = torch.randn(2, 80, T, generator=g)
    spk = torch.randn(2, 80, generator=g); t = torch.tensor([0.25, 0.25]); mask = torch.ones(2, 1, T)
    for streaming in (False, True):
        out = flow.decoder.estimator(x, mask, mu, t, spk, cond, streaming=streaming).cpu()
        ref = OF.estimator(sd, fc, x, mask, mu, t, spk, cond, streaming)
        err = _rel(out, ref)
        print("estimator %s streaming=%s: rel L2 %.2e" % (precision, streaming, err))
        assert err < (2e-4 if precision == "fp32" else 5e-2), (precision, streaming, err)
    h, _ = flow.encoder(torch.randn(1, 24 if lib.emulated else 60, fc.dim, generator=g), torch.tensor([60]), streaming=False)
    # (encoder output only has to be finite here; its parity is covered at tiny size and through inference() below at full size)
    assert torch.isfinite(h).all()


def test_flow_inference_fullsize(lib):
    fc = _cfgs(lib)[1]
    sd = W.make_flow(fc)
    n_p, n_t = (7, 13) if lib.emulated else (20, 40)
    g = torch.Generator().manual_seed(6)
    tok = torch.randint(0, fc.vocab, (1, n_t), generator=g, dtype=torch.int32); ptok = torch.randint(0, fc.vocab, (1, n_p), generator=g, dtype=torch.int32)
    pfeat = torch.randn(1, 2 * n_p, 80, generator=g) * 2 - 5; emb = torch.randn(1, fc.spk_dim, generator=g)
    n = lambda k: torch.tensor([k], dtype=torch.int32)
    flow = CausalMaskedDiffWithXvec(sd, fc, lib=lib, n_timesteps=2)
    mel, _ = flow.inference(token=tok, token_len=n(n_t), prompt_token=ptok, prompt_token_len=n(n_p), prompt_feat=pfeat, prompt_feat_len=n(2 * n_p),
                            embedding=emb, streaming=False, finalize=True)
    ref = OF.inference(sd, fc, tok, ptok, pfeat, emb, streaming=False, finalize=True, n_timesteps=2)
    err = _rel(mel.cpu(), ref)
    print("flow.inference (2 Euler steps): rel L2 %.2e" % err)
    assert mel.shape == ref.shape and err < 5e-3, err


def test_hift_decode_fullsize(lib):
    hc = _cfgs(lib)[2]
    sd = W.make_hift(hc)
    hift = HiFTGenerator(sd, hc, lib=lib)
    gen = torch.Generator().manual_seed(4)
    m = 7 if lib.emulated else 30
    mel = torch.randn(1, 80, m, generator=gen) * 2 - 5
    s = torch.tanh(torch.randn(1, 1, 480 * m, generator=gen))
    out = hift.decode(mel, s).cpu()
    ref = OH.decode(sd, hc, mel, s)
    err = _rel(out, ref)
    print("hift.decode: rel L2 %.2e" % err)
    assert out.shape == ref.shape and err < 5e-3, err
