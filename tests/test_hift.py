"""Stage B6 parity: HIP HiFT (f0 predictor, harmonic source, conv stack, iSTFT) vs the oracle and the reference goldens."""
import os

import numpy as np
import pytest
import torch

from cosyvoice_amd.hift import HiFTGenerator
from oracle import hift as OH
from cosyvoice_amd import synthetic as W

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def tiny():
    cfg = W.tiny()[2]
    return cfg, W.make_hift(cfg)


def test_matches_reference_golden(lib, tiny):
    """f0, source, decode and the cache_source path against vectors produced by the REAL reference (hift_tiny.npz)."""
    cfg, sd = tiny
    g = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(G, "hift_tiny.npz")).items()}
    hift = HiFTGenerator(sd, cfg, lib=lib)
    torch.testing.assert_close(hift.f0_predictor(g["mel"]).cpu(), g["f0"], rtol=1e-4, atol=1e-3)
    # decoder pinned tightly by feeding it the reference's own source
    torch.testing.assert_close(hift.decode(g["mel"], g["source"]).cpu(), g["speech"], rtol=2e-4, atol=2e-4)
    # full inference with the reference's noise draws: the source integrates f0 into a phase of thousands of radians, so fp32
    # round-off is amplified (see tests/test_oracle_golden.py) -> 2e-3 on the source
    speech, source = hift.inference(g["mel"], noise=g["noise"])
    torch.testing.assert_close(source.cpu(), g["source"], rtol=0, atol=2e-3)
    speech_c, source_c = hift.inference(g["mel"], cache_source=g["cache"], noise=g["noise"])
    torch.testing.assert_close(source_c.cpu()[:, :, :960], g["cache"], rtol=0, atol=0)
    torch.testing.assert_close(source_c.cpu(), g["source_c"], rtol=0, atol=2e-3)
    torch.testing.assert_close(hift.decode(g["mel"], g["source_c"]).cpu(), g["speech_c"], rtol=2e-4, atol=2e-4)


def test_decode_vs_oracle(lib, tiny):
    cfg, sd = tiny
    hift = HiFTGenerator(sd, cfg, lib=lib)
    gen = torch.Generator().manual_seed(4)
    m = 7
    mel = torch.randn(1, 80, m, generator=gen) * 2 - 5
    s = torch.tanh(torch.randn(1, 1, 480 * m, generator=gen))
    torch.testing.assert_close(hift.decode(mel, s).cpu(), OH.decode(sd, cfg, mel, s), rtol=2e-4, atol=2e-4)


def test_source_statistics_with_device_rng(lib, tiny):
    """Without the parity hook the SineGen2 noise comes from the in-kernel RNG: same f0, same deterministic part, noise ~ N(0,1)."""
    cfg, sd = tiny
    hift = HiFTGenerator(sd, cfg, lib=lib)
    gen = torch.Generator().manual_seed(6)
    mel = torch.randn(1, 80, 6, generator=gen) * 2 - 5
    sp1, s1 = hift.inference(mel)
    sp2, s2 = hift.inference(mel)
    _, s0 = hift.inference(mel, noise=torch.zeros(480 * 6, 9))
    assert not torch.equal(s1, s2)                                     # fresh draws per call
    assert (s1.cpu() - s0.cpu()).abs().max() < 0.6 and (s1.cpu() - s0.cpu()).abs().mean() > 1e-4
    assert torch.isfinite(sp1).all() and sp1.abs().max() <= cfg.audio_limit + 1e-6


def test_single_frame_and_f0_boundary(lib, tiny):
    """Shortest input (one mel frame = 480 samples: every conv is all padding) and the f0 predictor on its own."""
    cfg, sd = tiny
    hift = HiFTGenerator(sd, cfg, lib=lib)
    gen = torch.Generator().manual_seed(8)
    for m in (1, 3):
        mel = torch.randn(1, 80, m, generator=gen) * 2 - 5
        s = torch.tanh(torch.randn(1, 1, 480 * m, generator=gen))
        torch.testing.assert_close(hift.decode(mel, s).cpu(), OH.decode(sd, cfg, mel, s), rtol=2e-4, atol=2e-4)
        torch.testing.assert_close(hift.f0_predictor(mel).cpu(), OH.f0_predictor(sd, mel), rtol=1e-4, atol=1e-3)


def test_snake_once_and_two_sided_split_variants(lib, tiny, monkeypatch):
    """Round 3, the two changes to the ResBlock convolutions, each against its predecessor on the same input:
      * Snake applied once per value (elementwise pass / conv epilogue / second output) instead of in the prologue of every tap and N-tile of the consuming
        convolution (CV_HIFT_SNAKE_ONCE=0): the same fp32 values enter the same products -> bit-identical waveform;
      * both operands split into bf16 planes, six exact products per k on the bf16 matrix pipe (gemm_conv.h WX3) instead of the fp32 MFMA chain
        (CV_GEMM_WX3=0): fp32-rounding distance, far inside the oracle tolerance."""
    cfg, sd = tiny
    hift = HiFTGenerator(sd, cfg, lib=lib)
    gen = torch.Generator().manual_seed(12)
    m = 9
    mel = torch.randn(1, 80, m, generator=gen) * 2 - 5
    s = torch.tanh(torch.randn(1, 1, 480 * m, generator=gen))
    ref = OH.decode(sd, cfg, mel, s)
    base = hift.decode(mel, s).cpu()
    monkeypatch.setenv("CV_HIFT_SNAKE_ONCE", "0")
    pro = hift.decode(mel, s).cpu()
    assert torch.equal(base, pro)
    monkeypatch.delenv("CV_HIFT_SNAKE_ONCE")
    monkeypatch.setenv("CV_GEMM_WX3", "0")
    chain = hift.decode(mel, s).cpu()
    assert not torch.equal(base, chain)                                # the split path really ran
    e_split, e_chain = (base - ref).abs().max().item(), (chain - ref).abs().max().item()
    assert e_split < 2e-4 and e_chain < 2e-4 and (base - chain).abs().max().item() < 2e-5, (e_split, e_chain)


def test_three_plane_products_option(lib, tiny):
    """cv_hift_set_option "terms" = 3 (round 6, off by default): the decoder's convolutions keep three of the six plane products of the two-sided bf16 split
    (x1 w1 + x1 w2 + x2 w1: 16 mantissa bits per factor, gemm_conv.h w3_terms).  Stated tolerance: the waveform stays within 1e-4 rel-L2 / 70 dB of the six-term
    (fp32-exact class) one on the same source (measured at full size on the MI355X: 1.5e-5 / 96 dB, profiles/r6_hift.txt); the f0 predictor - whose output is
    integrated into a phase over the whole utterance - is NOT affected: bit-identical f0 and source; 6 restores the default bits."""
    import ctypes as C
    cfg, sd = tiny
    g = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(G, "hift_tiny.npz")).items()}
    hift = HiFTGenerator(sd, cfg, lib=lib)
    six, src6 = hift.inference(g["mel"], noise=g["noise"])
    six, src6 = six.cpu().clone(), src6.cpu().clone()
    lib.cv_hift_set_option(hift._h, b"terms", C.c_int32(3))
    three, src3 = hift.inference(g["mel"], noise=g["noise"])
    three, src3 = three.cpu().clone(), src3.cpu().clone()
    assert torch.equal(src3, src6)                                   # f0 predictor and source: exact either way
    err = (three.double() - six.double())
    rel = float(err.norm() / six.double().norm())
    assert 0.0 < rel < 1e-4 and 20 * np.log10(1.0 / rel) > 70.0, rel # really another arithmetic, and inside the stated tolerance
    lib.cv_hift_set_option(hift._h, b"terms", C.c_int32(6))
    again, _ = hift.inference(g["mel"], noise=g["noise"])
    assert torch.equal(again.cpu(), six)
    with pytest.raises(Exception):
        lib.cv_hift_set_option(hift._h, b"terms", C.c_int32(4))
