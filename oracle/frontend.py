"""Oracle: the prompt log-mel front end (TEST INFRASTRUCTURE - see oracle/__init__.py).  PARITY UNPINNED: `matcha.utils.audio.mel_spectrogram`
(called at cosyvoice/cli/frontend.py:120-125, configured by cosyvoice2.yaml:150-158) lives in the un-vendored Matcha-TTS submodule and
`librosa.filters.mel` (librosa, unpinned in requirements.txt) is not installed; both are restated from their published algorithms (SURVEY.md
Appendix B): reflect pad (n_fft - hop) / 2, torch.stft(center=False, hann_window), sqrt(re^2 + im^2 + 1e-9), Slaney-scale area-normalised
triangular filters, log(clamp(., 1e-5)).  The filterbank here is an independent float64 restatement (loops, not the product's vectorised code);
its closed-form anchors are tested in tests/test_frontend.py."""
import math

import numpy as np
import torch


def hz_to_mel(f):
    f_sp = 200.0 / 3
    if f >= 1000.0:
        return 1000.0 / f_sp + math.log(f / 1000.0) / (math.log(6.4) / 27.0)
    return f / f_sp


def mel_to_hz(m):
    f_sp = 200.0 / 3
    if m >= 1000.0 / f_sp:
        return 1000.0 * math.exp((math.log(6.4) / 27.0) * (m - 1000.0 / f_sp))
    return f_sp * m


def librosa_mel(sr, n_fft, n_mels, fmin, fmax):
    fmax = sr / 2.0 if fmax is None else fmax
    lo, hi = hz_to_mel(fmin), hz_to_mel(fmax)
    pts = [mel_to_hz(lo + (hi - lo) * i / (n_mels + 1)) for i in range(n_mels + 2)]
    bins = n_fft // 2 + 1
    w = np.zeros((n_mels, bins))
    for m in range(n_mels):
        l, c, r = pts[m], pts[m + 1], pts[m + 2]
        for k in range(bins):
            f = k * sr / n_fft
            up, down = (f - l) / (c - l), (r - f) / (r - c)
            w[m, k] = max(0.0, min(up, down)) * 2.0 / (r - l)
    return w


def mel_spectrogram(y, n_fft=1920, num_mels=80, sampling_rate=24000, hop_size=480, win_size=1920, fmin=0, fmax=8000, center=False):
    """y [1, L] -> [1, num_mels, T]   (matcha.utils.audio.mel_spectrogram)."""
    basis = torch.from_numpy(librosa_mel(sampling_rate, n_fft, num_mels, fmin, fmax)).float()
    pad = (n_fft - hop_size) // 2
    y = torch.nn.functional.pad(y.unsqueeze(1), (pad, pad), mode="reflect").squeeze(1)
    spec = torch.view_as_real(torch.stft(y, n_fft, hop_length=hop_size, win_length=win_size, window=torch.hann_window(win_size), center=center,
                                         pad_mode="reflect", normalized=False, onesided=True, return_complex=True))
    spec = torch.sqrt(spec.pow(2).sum(-1) + 1e-9)
    return torch.log(torch.clamp(torch.matmul(basis, spec), min=1e-5))
