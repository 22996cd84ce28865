"""SURVEY.md section 8 row a17 (flow part): CausalMaskedDiffWithDiT / DiT on the device vs the oracle and vs the goldens generated from the REAL
reference classes (tests/golden/dit_tiny.npz: cosyvoice.flow.flow.CausalMaskedDiffWithDiT, flow.DiT.dit.DiT, upsample_encoder.PreLookaheadLayer)."""
import os

import numpy as np
import pytest
import torch

from cosyvoice_amd.flow import CausalMaskedDiffWithDiT
from oracle import dit as OD
from oracle import flow as OF
from cosyvoice_amd import synthetic as W

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def tiny():
    cfg = W.tiny_cv3_flow()
    return cfg, W.make_flow_dit(cfg)


def _gold():
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(G, "dit_tiny.npz")).items()}


def test_dit_estimator_matches_reference_golden(lib, tiny):
    """B3 for CosyVoice3: decoder.estimator(x, mask, mu, t, spks, cond, streaming) at the reference's own export tolerance (rtol 1e-2 / atol 1e-4,
    bin/export_onnx.py:109) and 2e-4, both mask modes; then a longer sequence with per-row times against the oracle."""
    cfg, sd = tiny
    g = _gold()
    flow = CausalMaskedDiffWithDiT(sd, cfg, lib=lib)
    T = g["est_x"].shape[2]
    mask = torch.ones(2, 1, T)
    for streaming, key in ((False, "est_full"), (True, "est_stream")):
        out = flow.decoder.estimator(g["est_x"], mask, g["est_mu"], g["est_t"], g["est_spk"], g["est_cond"], streaming=streaming).cpu()
        torch.testing.assert_close(out, g[key], rtol=1e-2, atol=1e-4)
        torch.testing.assert_close(out, g[key], rtol=2e-4, atol=2e-4)
    gen = torch.Generator().manual_seed(4)
    T = 70
    x = torch.randn(2, 80, T, generator=gen); mu = torch.randn(2, 80, T, generator=gen); cond = torch.randn(2, 80, T, generator=gen)
    spk = torch.randn(2, 80, generator=gen); t = torch.tensor([0.15, 0.8])
    for streaming in (False, True):
        out = flow.decoder.estimator(x, torch.ones(2, 1, T), mu, t, spk, cond, streaming=streaming).cpu()
        torch.testing.assert_close(out, OD.estimator(sd, cfg, x, torch.ones(2, 1, T), mu, t, spk, cond, streaming), rtol=3e-4, atol=3e-4)


@pytest.mark.parametrize("streaming,finalize,key", [(False, True, "mel_full"), (True, False, "mel_stream")])
def test_dit_flow_inference_matches_reference_golden(lib, tiny, streaming, finalize, key):
    """B5 for CosyVoice3: CausalMaskedDiffWithDiT.inference (PreLookaheadLayer with / without context, repeat_interleave, prompt conditioning, CFG
    Euler solve on the device) against the real reference's output."""
    cfg, sd = tiny
    g = _gold()
    flow = CausalMaskedDiffWithDiT(sd, cfg, lib=lib)
    n = lambda k: torch.tensor([k], dtype=torch.int32)
    for rep in range(3 if lib.emulated else 3):                  # the third call replays the captured hipGraph of the Euler solve
        mel, _ = flow.inference(token=g["token"].int(), token_len=n(g["token"].shape[1]), prompt_token=g["prompt_token"].int(), prompt_token_len=n(g["prompt_token"].shape[1]),
                                prompt_feat=g["prompt_feat"], prompt_feat_len=n(g["prompt_feat"].shape[1]), embedding=g["embedding"], streaming=streaming, finalize=finalize)
        torch.testing.assert_close(mel.cpu(), g[key], rtol=1e-3, atol=1e-3)
    assert flow.encoder is None


def test_dit_bf16_mode_noise_level(lib, tiny):
    """bf16 mode (fp16=True of the reference API; BASELINE.json configs[4] names a reduced-precision path): the product's distance to the fp32 oracle
    must not exceed what the oracle's own bf16 mirror shows (same criterion as tests/test_flow.py::test_bf16_mode_estimator_noise_level)."""
    cfg, sd = tiny
    flow = CausalMaskedDiffWithDiT(sd, cfg, lib=lib, precision="bf16")
    gen = torch.Generator().manual_seed(9)
    T = 45
    x = torch.randn(2, 80, T, generator=gen); mu = torch.randn(2, 80, T, generator=gen); cond = torch.randn(2, 80, T, generator=gen)
    spk = torch.randn(2, 80, generator=gen); t = torch.tensor([0.5, 0.5]); mask = torch.ones(2, 1, T)
    out = flow.decoder.estimator(x, mask, mu, t, spk, cond, streaming=False).cpu()
    ref = OD.estimator(sd, cfg, x, mask, mu, t, spk, cond, False)
    with OF.bf16_act():
        mir = OD.estimator(sd, cfg, x, mask, mu, t, spk, cond, False)
    e_dev, e_mir = (out - ref).abs().mean().item(), (mir - ref).abs().mean().item()
    assert e_mir > 1e-4 and e_dev < 1.5 * e_mir + 1e-4, (e_dev, e_mir)


def test_dit_streaming_equals_one_shot_on_finished_chunks(lib, tiny):
    """The reference's own check (flow/flow.py:417-443, print-only there): with the chunk mask, the frames of a chunked call that lie in complete
    chunks equal the one-shot streaming result."""
    cfg, sd = tiny
    flow = CausalMaskedDiffWithDiT(sd, cfg, lib=lib)
    gen = torch.Generator().manual_seed(2)
    chunk = cfg.chunk
    n_p, n_t = chunk, 4 * chunk
    token = torch.randint(0, cfg.vocab, (1, n_t), generator=gen, dtype=torch.int32); ptok = torch.randint(0, cfg.vocab, (1, n_p), generator=gen, dtype=torch.int32)
    pfeat = torch.rand(1, 2 * n_p, 80, generator=gen); emb = torch.rand(1, cfg.spk_dim, generator=gen)
    n = lambda k: torch.tensor([k], dtype=torch.int32)
    common = dict(prompt_token=ptok, prompt_token_len=n(n_p), prompt_feat=pfeat, prompt_feat_len=n(2 * n_p), embedding=emb)
    full, _ = flow.inference(token=token, token_len=n(n_t), streaming=True, finalize=True, **common)
    full = full.cpu()
    la = cfg.pre_lookahead
    for i in range(0, n_t, chunk):
        fin = i + chunk + la >= n_t
        part, _ = flow.inference(token=token[:, : i + chunk + la], token_len=n(min(n_t, i + chunk + la)), streaming=True, finalize=fin, **common)
        part = part.cpu()[:, :, 2 * i:]
        torch.testing.assert_close(full[:, :, 2 * i: 2 * i + part.shape[2]], part, rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_dit_inference_batch_equals_single(lib, tiny, precision):
    """cv_flow_inference_batch on the DiT estimator: utterances of equal shape solved in one pass (estimator batch rows = 2 x utterances) give, each,
    exactly the mel `inference()` gives for it alone (flow/flow.py:246); then the single path and a longer request on the same handle."""
    import dataclasses
    cfg, sd = tiny
    cfg = dataclasses.replace(cfg, n_timesteps=2)
    flow = CausalMaskedDiffWithDiT(sd, cfg, lib=lib, precision=precision)
    g = torch.Generator().manual_seed(72)
    n = lambda k: torch.tensor([k], dtype=torch.int32)
    items = [dict(token=torch.randint(0, cfg.vocab, (1, 9), generator=g, dtype=torch.int32), prompt_token=torch.randint(0, cfg.vocab, (1, 5), generator=g, dtype=torch.int32),
                  prompt_feat=torch.randn(1, 10, cfg.mel, generator=g) * 2 - 5, embedding=torch.randn(1, cfg.spk_dim, generator=g)) for _ in range(3)]
    single = lambda it, tok: flow.inference(token=tok, token_len=n(tok.shape[1]), prompt_token=it["prompt_token"], prompt_token_len=n(5), prompt_feat=it["prompt_feat"],
                                            prompt_feat_len=n(10), embedding=it["embedding"], streaming=False, finalize=True)[0].cpu()
    alone = [single(it, it["token"]) for it in items]
    assert not torch.equal(alone[0], alone[1])
    for rep in range(3):                                          # the third call replays the captured graph of the batched solve
        for a, b in zip(alone, flow.inference_batch(items)):
            assert torch.equal(a, b.cpu())
    assert torch.equal(alone[2], flow.inference_batch(items[1:])[1].cpu())
    assert torch.equal(single(items[0], items[0]["token"]), alone[0])
    tok_long = torch.randint(0, cfg.vocab, (1, 27), generator=g, dtype=torch.int32)
    fresh = CausalMaskedDiffWithDiT(sd, cfg, lib=lib, precision=precision)
    assert torch.equal(single(items[0], tok_long), fresh.inference(token=tok_long, token_len=n(27), prompt_token=items[0]["prompt_token"], prompt_token_len=n(5),
                                                                  prompt_feat=items[0]["prompt_feat"], prompt_feat_len=n(10), embedding=items[0]["embedding"],
                                                                  streaming=False, finalize=True)[0].cpu())


def test_dit_inference_ragged_equals_single(lib, tiny):
    """The DiT estimator in a padded batch of different lengths (cv_flow_inference_ragged): causal position convs, per-row norms, attention with a key
    count per batch row - every utterance bit-identical to itself alone."""
    import dataclasses
    cfg, sd = tiny
    cfg = dataclasses.replace(cfg, n_timesteps=2)
    flow = CausalMaskedDiffWithDiT(sd, cfg, lib=lib, precision="bf16")
    g = torch.Generator().manual_seed(74)
    n = lambda k: torch.tensor([k], dtype=torch.int32)
    shapes = [(13, 5, 10), (8, 6, 12), (19, 4, 8)]
    items = [dict(token=torch.randint(0, cfg.vocab, (1, a), generator=g, dtype=torch.int32), prompt_token=torch.randint(0, cfg.vocab, (1, b), generator=g, dtype=torch.int32),
                  prompt_feat=torch.randn(1, c, cfg.mel, generator=g) * 2 - 5, embedding=torch.randn(1, cfg.spk_dim, generator=g)) for a, b, c in shapes]
    for streaming, finalize in ((False, True), (True, False)):
        alone = [flow.inference(token=it["token"], token_len=n(it["token"].shape[1]), prompt_token=it["prompt_token"], prompt_token_len=n(it["prompt_token"].shape[1]),
                                prompt_feat=it["prompt_feat"], prompt_feat_len=n(it["prompt_feat"].shape[1]), embedding=it["embedding"], streaming=streaming,
                                finalize=finalize)[0].cpu() for it in items]
        for rep in range(3):
            for a, b in zip(alone, flow.inference_batch(items, streaming=streaming, finalize=finalize)):
                assert a.shape == b.shape and torch.equal(a, b.cpu())
