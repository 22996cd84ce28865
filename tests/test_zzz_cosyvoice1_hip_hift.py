"""Kernel-backed CosyVoice-300M, the 22.05 kHz HiFTGenerator (see tests/test_zzz_cosyvoice1_hip.py)."""
import torch

from cv1k_common import build_hift, gold


def test_hift_22k_matches_reference(lib):
    g = gold("cv1_hift")                                            # the vocoder has no attention: tiny_cv1's golden of the real class applies
    h = build_hift(lib)
    torch.testing.assert_close(h.f0_predictor(g["feat"]).cpu(), g["f0"], rtol=1e-4, atol=1e-3)
    torch.manual_seed(77)
    speech, source = h.inference(speech_feat=g["feat"])
    assert speech.shape == (1, 30 * 256)
    # the harmonic phase is a cumulative sum over 7680 samples: fp32 summation order shows at ~1e-4 in sin(phase) (here: exact per-frame closed form in double)
    torch.testing.assert_close(source.cpu(), g["source"], rtol=0, atol=2e-3)
    torch.testing.assert_close(speech.cpu(), g["speech"], rtol=0, atol=5e-3)
    torch.testing.assert_close(h.decode(g["feat"], g["source"]).cpu(), g["speech"], rtol=2e-4, atol=2e-4)      # the conv stack + iSTFT given the reference's source
    torch.manual_seed(78)
    speech2, source2 = h.inference(speech_feat=g["feat"], cache_source=g["cache_source"])
    torch.testing.assert_close(source2.cpu()[:, :, :1024], g["cache_source"], rtol=0, atol=0)
    torch.testing.assert_close(speech2.cpu(), g["speech2"], rtol=0, atol=5e-3)
    # the counter-RNG noise of the product path: same harmonic part, noise of the same scale
    hd = build_hift(lib, rng="device")
    torch.manual_seed(77)
    _, src_d = hd.inference(speech_feat=g["feat"])
    voiced = (g["f0"].repeat_interleave(256, dim=1) > 10).reshape(-1)
    assert float((src_d.cpu() - g["source"]).reshape(-1)[voiced].abs().max()) < 0.05 and torch.isfinite(src_d).all()
