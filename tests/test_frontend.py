"""on-device 24 kHz prompt log-mel (cosyvoice/cli/frontend.py:120-125) vs the oracle restatement of
matcha.utils.audio.mel_spectrogram (parity unpinned: the Matcha submodule and librosa are absent, see oracle/frontend.py)."""
import numpy as np
import pytest
import torch

from cosyvoice_amd.frontend import MelSpectrogram, mel_filterbank
from oracle import frontend as OFE


def test_mel_filterbank_anchors():
    """The two restatements of librosa.filters.mel agree, and the closed-form properties of the Slaney filterbank hold: the scale is linear
    below 1 kHz (200/3 Hz per mel) and logarithmic above (27 steps per factor 6.4), every filter has unit area in Hz, neighbours overlap
    so that the interior of the bank is a partition with slowly varying weight, nothing sits above fmax."""
    a = mel_filterbank(24000, 1920, 80, 0, 8000)
    b = OFE.librosa_mel(24000, 1920, 80, 0, 8000)
    np.testing.assert_allclose(a, b, rtol=1e-5, atol=1e-9)
    assert a.shape == (80, 961) and (a >= 0).all()
    assert abs(OFE.hz_to_mel(1000.0) - 15.0) < 1e-12 and abs(OFE.mel_to_hz(15.0 + 27.0) - 6400.0) < 1e-6
    df = 24000 / 1920
    np.testing.assert_allclose(a.sum(1) * df, np.ones(80), rtol=0.08)            # unit area, sampled on a 12.5 Hz grid
    assert a[:, int(8000 / df) + 1:].max() == 0.0
    a3 = mel_filterbank(24000, 1920, 80, 0, None)                                # cosyvoice3.yaml: fmax None -> Nyquist
    assert a3[-1, -2] > 0.0


@pytest.mark.parametrize("fmax", [8000, None])
def test_prompt_mel_matches_oracle(lib, fmax):
    g = torch.Generator().manual_seed(8)
    L = 480 * (5 if lib.emulated else 174)                                       # 174 frames = the 3.48 s prompt of the benchmark utterance
    t = torch.arange(L) / 24000.0
    y = (0.3 * torch.sin(2 * np.pi * 220.0 * t) + 0.1 * torch.sin(2 * np.pi * 3100.0 * t * (1 + 0.2 * t)) + 0.02 * torch.randn(L, generator=g)).unsqueeze(0).clamp(-1, 1)
    fe = MelSpectrogram(n_fft=1920, num_mels=80, sampling_rate=24000, hop_size=480, win_size=1920, fmin=0, fmax=fmax, center=False, lib=lib)
    got = fe(y).cpu()
    want = OFE.mel_spectrogram(y, fmax=fmax)
    assert got.shape == want.shape == (1, 80, L // 480)
    # fp32 direct DFT vs torch's fp32 FFT: both carry ~1e-6 of the frame energy as absolute error, which the log magnifies for bins near the
    # 1e-5 clamp.  Stated tolerance: 2e-3 absolute on the log-mel where the oracle's mel energy is above 1e-3, 5e-2 elsewhere.
    loud = want > np.log(1e-3)
    assert (got - want)[loud].abs().max().item() < 2e-3 and (got - want).abs().max().item() < 5e-2
    # front-end layout (frontend.py:122-123): [1, T, 80]
    feat = fe(y).squeeze(dim=0).transpose(0, 1).unsqueeze(dim=0)
    assert feat.shape == (1, L // 480, 80)
