"""SURVEY.md section 8 row a17 (vocoder part): CausalHiFTGenerator / CausalConvRNNF0Predictor on the device vs the goldens generated from the REAL
reference classes (tests/golden/causal_hift_tiny.npz) and vs the oracle restatement."""
import dataclasses
import os

import numpy as np
import pytest
import torch

from cosyvoice_amd.hift import CausalHiFTGenerator
from oracle import hift as OH
from cosyvoice_amd import synthetic as W

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def tiny():
    cfg = dataclasses.replace(W.tiny()[2], causal=True)
    return cfg, W.make_hift(cfg)


def _gold():
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(G, "causal_hift_tiny.npz")).items()}


def test_causal_hift_matches_reference_golden(lib, tiny):
    cfg, sd = tiny
    g = _gold()
    h = CausalHiFTGenerator(sd, cfg, lib=lib)
    # f0: fp32 on the device vs the reference's float64 predictor
    torch.testing.assert_close(h.f0(g["mel"], True).cpu(), g["f0"], rtol=1e-4, atol=2e-3)
    assert h.f0(g["mel"][:, :, :13], False).shape == (1, 10)
    # decoder pinned tightly by feeding it the reference's own source (the source itself integrates f0 into a phase of thousands of radians)
    torch.testing.assert_close(h.decode(g["mel"], g["source"], True).cpu(), g["speech"], rtol=2e-4, atol=2e-4)
    torch.testing.assert_close(h.decode(g["mel"][:, :, :10], g["source_c"], False).cpu(), g["speech_c"], rtol=2e-4, atol=2e-4)
    speech, source = h.inference(g["mel"], True, noise=g["noise"])
    assert speech.shape == g["speech"].shape and source.shape == g["source"].shape
    torch.testing.assert_close(source.cpu(), g["source"], rtol=0, atol=5e-3)
    speech_c, source_c = h.inference(g["mel"][:, :, :13], False, noise=g["noise"])
    assert speech_c.shape == g["speech_c"].shape == (1, 480 * 5) and source_c.shape == g["source_c"].shape
    torch.testing.assert_close(source_c.cpu(), g["source_c"], rtol=0, atol=5e-3)


def test_causal_hift_streaming_equals_one_shot(lib, tiny):
    """The reference's own invariance check (generator.py:729-746, print-only there): every sample a non-final chunk emits equals the one-shot
    waveform - here exactly (the fp32 f0 / phase of a frame do not depend on the chunk), with the default counter-RNG noise."""
    cfg, sd = tiny
    h = CausalHiFTGenerator(sd, cfg, lib=lib)
    gen = torch.Generator().manual_seed(5)
    m, chunk, ctx = 30, 9, 8
    mel = torch.rand(1, 80, m, generator=gen) * 6 - 7
    full, _ = h.inference(mel, True)
    full = full.cpu()
    assert full.shape == (1, 480 * m)
    for i in range(0, m, chunk):
        fin = i + chunk + ctx >= m
        part, _ = h.inference(mel[:, :, : i + chunk + ctx], fin)
        part = part.cpu()[:, i * 480:]
        assert part.shape[1] == (m - i if fin else chunk) * 480
        torch.testing.assert_close(full[:, i * 480: i * 480 + part.shape[1]], part, rtol=0, atol=1e-6)
        if fin:
            break
    # and against the oracle with the same explicit noise
    noise = torch.rand(1, 480 * m, 9, generator=gen)
    sp, src = h.inference(mel, True, noise=noise)
    _, src_o = OH.causal_inference(sd, cfg, mel, True, None, noise, f0_dtype=torch.float32)
    torch.testing.assert_close(src.cpu(), src_o, rtol=0, atol=5e-3)
    torch.testing.assert_close(h.decode(mel, src_o, True).cpu(), OH.causal_decode(sd, cfg, mel, src_o, True), rtol=2e-4, atol=2e-4)


def test_reduced_products_mode_of_the_causal_decoder(lib, tiny):
    """CosyVoice3Model(fp16=True) (the reference runs flow AND vocoder of that model under autocast, cli/model.py:426-447) builds its CausalHiFTGenerator with
    terms=3: the causal decoder's convolutions keep 16 significand bits per factor (fp32 accumulation; more than autocast's fp16 operands).  Against the golden of the
    REAL class the decoder stays inside the default mode's tolerance, against the default mode it is within 1e-4 rel-L2 (> 80 dB), f0 (double) is untouched, clones
    keep the mode, and a chunk still equals the one-shot prefix."""
    cfg, sd = tiny
    g = _gold()
    h6 = CausalHiFTGenerator(sd, cfg, lib=lib)
    h3 = CausalHiFTGenerator(sd, cfg, lib=lib, terms=3)
    assert h3.terms == 3 and h3.clone().terms == 3 and h6.terms == 6
    six = h6.decode(g["mel"], g["source"], True).cpu().clone()
    three = h3.decode(g["mel"], g["source"], True).cpu().clone()
    torch.testing.assert_close(three, g["speech"], rtol=2e-4, atol=2e-4)                     # the real class's waveform, the default mode's bound
    rel = float((three.double() - six.double()).norm() / six.double().norm())
    assert 0.0 < rel < 1e-4, rel                                                             # really another arithmetic, > 80 dB
    assert torch.equal(h3.f0(g["mel"], True).cpu(), h6.f0(g["mel"], True).cpu())            # the predictor is not part of the mode
    assert torch.equal(h3.clone().decode(g["mel"], g["source"], True).cpu(), three)
    part = h3.decode(g["mel"][:, :, :10], g["source_c"], False).cpu()
    torch.testing.assert_close(part, g["speech_c"], rtol=2e-4, atol=2e-4)
