"""The opt-in fp8 batched decode (BASELINE.json configs[4] "fp8 MFMA LLM path"; cosyvoice_amd/csrc/llm_batch_kernels.h::skinny_fp8_kernel).  There is
no reference for this mode, so it is held to (1) an oracle that mirrors its definition exactly (oracle/llm_fp8.py: e4m3
weights with per-row scales, per-sequence activation scales, fp32 accumulation) and (2) the fp32 oracle at the tolerance the format allows (logits
within a few percent).  Tolerance against the mirror: 2e-3 of the output scale, not summation order - an activation that sits on an e4m3 rounding
boundary takes the neighbouring code when the scale differs in its last bit (the MI355X forms 1 / sx and x * inv with its own division / rounding
sequence; 2 of 96 outputs moved by 4e-5 on hardware), and greedy ids then follow the mirror only until the first near-tie.  Under the emulator (host
arithmetic = the mirror's) kernel and mirror agree to 2e-5 and the ids are equal."""
import ctypes as C

import pytest
import torch

from cosyvoice_amd import synthetic as W
from cosyvoice_amd import weights as Wt
from cosyvoice_amd.llm import Qwen2LM
from oracle import llm as OL
from oracle import llm_fp8 as OF8


@pytest.mark.parametrize("N,K,nb,mode,gamma,ksplit,rt", [(96, 128, 5, 0, True, 1, 1), (70, 256, 16, 0, False, 1, 1), (128, 128, 3, 1, True, 1, 2),
                                                         (64, 512, 8, 2, False, 4, 2), (63, 128, 2, 0, True, 1, 2)])
def test_skinny_fp8_kernel_matches_mirror(lib, N, K, nb, mode, gamma, ksplit, rt):
    g = torch.Generator().manual_seed(N + K + nb)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    x = torch.randn(nb, K, generator=g) * 1.7
    gam = (1 + 0.1 * torch.randn(K, generator=g)) if gamma else None
    bias = torch.randn(N, generator=g) * 0.1 if mode == 0 else None
    res = torch.randn(nb, N, generator=g) if mode == 0 else None
    w8, sw = Wt.quantize_fp8_rows(w)
    dev = lambda t: None if t is None else lib.hook(t.to(lib.device).contiguous())
    w8d, swd, xd, gd, bd, rd = dev(w8), dev(sw), dev(x), dev(gam), dev(bias), dev(res)
    n_out = N // 2 if mode == 1 else N
    y = lib.hook(torch.zeros((ksplit if mode == 2 else 1) * nb, n_out, device=lib.device))
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    lib.cv_skinny_fp8(p(w8d), p(swd), p(bd), p(xd), C.c_int64(K), p(y), C.c_int64(n_out), C.c_int32(N), C.c_int32(K), p(gd), C.c_float(1e-6), p(rd), C.c_int64(N),
                      C.c_int32(mode), C.c_int32(nb), C.c_int32(ksplit), C.c_int32(rt), None)
    got = y.cpu()
    tol = 2e-5 if lib.emulated else 2e-3
    lin = OF8.Fp8Linear(w)
    assert torch.equal(lin.wq, w8.view(torch.float8_e4m3fn).float()) and torch.equal(lin.sw, sw)       # the product's quantiser = the mirror's
    for b in range(nb):
        if mode == 2:
            step = K // ksplit
            for s in range(ksplit):
                q, sx = OF8.quant_act(x[b, s * step:(s + 1) * step])
                want = ((lin.wq[:, s * step:(s + 1) * step] @ q) * sx) * lin.sw
                torch.testing.assert_close(got[s * nb + b], want, rtol=tol, atol=tol * float(want.abs().max()))
            continue
        want = lin(x[b], gam, 1e-6)
        if mode == 1:
            want = (want[0::2] / (1 + torch.exp(-want[0::2]))) * want[1::2]
        else:
            want = want + bias + res[b]
        torch.testing.assert_close(got[b], want, rtol=tol, atol=tol * float(want.abs().max()))


def _req(cfg, seed, n_text, n_prompt_text, n_prompt_tok):
    u = W.synthetic_utterance(cfg, W.tiny()[1], n_prompt_tok=n_prompt_tok, n_prompt_text=n_prompt_text, n_text=n_text, seed=seed)
    return dict(text=u["text"], prompt_text=u["prompt_text"], prompt_speech_token=u["llm_prompt_speech_token"])


def test_fp8_batched_decode_tokens_equal_the_mirror(lib):
    cfg = W.tiny()[0]
    sd = W.make_llm(cfg)
    lm = Qwen2LM(sd, cfg, lib=lib, max_len=160, sampling="greedy", decode_chunk=6, batch_fp8=True)
    reqs = [_req(cfg, 300 + i, 3 + i % 2, 2, 6 + 9 * i) for i in range(3)]
    got = lm.inference_batch(reqs, max_token_text_ratio=4, min_token_text_ratio=2)
    agree32, agree8, total = 0, 0, 0
    for i, (r, g) in enumerate(zip(reqs, got)):
        trace = {}
        want = OF8.inference(sd, cfg, r["text"], r["prompt_text"], r["prompt_speech_token"], max_token_text_ratio=4, min_token_text_ratio=2, trace=trace)
        if lib.emulated:
            assert g == want, (g, want)
        assert len(g) == len(want)
        agree8 += sum(a == b for a, b in zip(g, want)); total += len(want)
        ref = OL.inference(sd, cfg, r["text"], r["prompt_text"], r["prompt_speech_token"], max_token_text_ratio=4, min_token_text_ratio=2)
        agree32 += sum(a == b for a, b in zip(g, ref))
    assert agree8 >= 0.5 * total, (agree8, total)
    # first decode step of every slot: the device's log-probabilities against the mirror's (robust to later near-ties)
    st = None
    lm.lib.cv_llm_batch_begin(lm._h, C.c_int32(len(reqs)), st)
    inputs = [lm.build_lm_input(r["text"], r["prompt_text"], r["prompt_speech_token"]) for r in reqs]
    lm._prefill_slots(list(range(len(reqs))), inputs, [lm.make_sampling(2, 8) for _ in reqs], st)
    buf, n_out, f = (C.c_int32 * len(reqs))(), (C.c_int32 * len(reqs))(), (C.c_int32 * len(reqs))()
    lm.lib.cv_llm_batch_decode(lm._h, C.c_int32(1), buf, n_out, f, st)
    V = cfg.speech_token_size + cfg.n_special
    for i, r in enumerate(reqs):
        host = (C.c_float * V)()
        lm.lib.cv_llm_batch_logits(lm._h, C.c_int32(i), host, st)
        trace = {}
        OF8.inference(sd, cfg, r["text"], r["prompt_text"], r["prompt_speech_token"], max_token_text_ratio=1, min_token_text_ratio=1, trace=trace)
        dev_logp = torch.tensor(list(host)).log_softmax(-1)
        # (the hidden state entering the head comes from two different fp32 prefills - device and mirror: one activation on a rounding boundary moves every logit)
        # Measured on the MI355X: 62 of 63 log-probabilities within 1e-2, the worst 2.1e-2 (a wrong scale or layout is off by > 0.3).
        d = (dev_logp - trace["logp"][0]).abs()
        assert d.max().item() < 6e-2 and d.mean().item() < 1e-2, (d.max().item(), d.mean().item())
    # the single-sequence path of the same object still runs on the bf16 weights: fp32-oracle tokens
    r = reqs[0]
    t = lambda n: torch.tensor([n], dtype=torch.int32)
    alone = list(lm.inference(text=r["text"], text_len=t(r["text"].shape[1]), prompt_text=r["prompt_text"], prompt_text_len=t(2), prompt_speech_token=r["prompt_speech_token"],
                              prompt_speech_token_len=t(r["prompt_speech_token"].shape[1]), max_token_text_ratio=4, min_token_text_ratio=2))
    assert alone == OL.inference(sd, cfg, r["text"], r["prompt_text"], r["prompt_speech_token"], max_token_text_ratio=4, min_token_text_ratio=2)
    assert agree32 >= 1                                              # fp8 logits are close enough to share tokens with the fp32 oracle, not equal to it


def test_fp8_logits_close_to_fp32_oracle():
    """What the format costs: first-step log-probabilities of the mirror against the fp32 oracle (tiny model, CPU only)."""
    cfg = W.tiny()[0]
    sd = W.make_llm(cfg)
    r = _req(cfg, 77, 4, 3, 12)
    t8, t32 = {}, {}
    OF8.inference(sd, cfg, r["text"], r["prompt_text"], r["prompt_speech_token"], max_token_text_ratio=2, min_token_text_ratio=2, trace=t8)
    OL.inference(sd, cfg, r["text"], r["prompt_text"], r["prompt_speech_token"], max_token_text_ratio=2, min_token_text_ratio=2, trace=t32)
    a, b = t8["logp"][0], t32["logp"][0]
    rel = ((a - b).norm() / b.norm()).item()
    assert rel < 0.08, rel
