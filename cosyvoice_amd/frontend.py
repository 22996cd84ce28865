"""On-device prompt featurisation: the 24 kHz log-mel spectrogram `CosyVoiceFrontEnd._extract_speech_feat` computes
for every zero-shot / cross-lingual request (cosyvoice/cli/frontend.py:120-125) with `matcha.utils.audio.mel_spectrogram` configured by
cosyvoice2.yaml:150-158 (n_fft 1920, hop 480, win 1920, 80 mels, fmin 0, fmax 8000, center=False; cosyvoice3.yaml: fmax None).

`MelSpectrogram` is a drop-in for that `feat_extractor` callable: `mel = feat_extractor(speech[1, L])` -> `[1, 80, L // hop]` on the device.
Algorithm (matcha/utils/audio.py, restated in SURVEY.md Appendix B; the Matcha-TTS submodule is empty in the reference tree):
reflect-pad (n_fft - hop) / 2  ->  STFT(center=False, periodic Hann)  ->  sqrt(re^2 + im^2 + 1e-9)  ->  librosa mel basis (Slaney scale,
area-normalised)  ->  log(clamp(., 1e-5)).
Device mapping: the STFT is ONE strided implicit GEMM of the exact-fp32 MFMA kernel (an im2col window of a hop-strided 1-D signal is
contiguous: lda = hop, K = n_fft) against the windowed DFT basis, the mel projection a second GEMM with the log-clamp in its epilogue;
two elementwise kernels (cv_reflect_pad, cv_stft_magnitude) in between.  No FFT library, no CPU round trip."""
import ctypes as C
import math

import numpy as np
import torch

from . import ops
from ._lib import get_lib, stream_ptr


def _hz_to_mel(f):
    """librosa.hz_to_mel(htk=False): linear below 1 kHz, logarithmic above (Slaney's Auditory Toolbox)."""
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz, min_log_mel, logstep = 1000.0, 1000.0 / f_sp, math.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, mels)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz, min_log_mel, logstep = 1000.0, 1000.0 / f_sp, math.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filterbank(sr, n_fft, n_mels, fmin, fmax):
    """librosa.filters.mel(sr=, n_fft=, n_mels=, fmin=, fmax=) with its defaults (htk=False, norm='slaney'): triangular filters on the
    Slaney mel scale, each normalised to unit area.  Returns float32 [n_mels, n_fft // 2 + 1]."""
    fmax = sr / 2.0 if fmax is None else float(fmax)
    fft_f = np.linspace(0.0, sr / 2.0, n_fft // 2 + 1)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fft_f[None, :]
    lower = -ramps[:-2] / fdiff[:-1, None]
    upper = ramps[2:] / fdiff[1:, None]
    w = np.maximum(0.0, np.minimum(lower, upper))
    w *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return w.astype(np.float32)


class MelSpectrogram:
    def __init__(self, n_fft=1920, num_mels=80, sampling_rate=24000, hop_size=480, win_size=1920, fmin=0, fmax=8000, center=False, lib=None):
        assert not center, "the CosyVoice configs use center=False (explicit reflect padding)"
        assert win_size == n_fft and n_fft % 32 == 0 and (n_fft - hop_size) % 2 == 0
        self.lib = lib or get_lib()
        self.device = torch.device(self.lib.device)
        self.n_fft, self.hop, self.n_mels, self.bins = n_fft, hop_size, num_mels, n_fft // 2 + 1
        self.pad = (n_fft - hop_size) // 2
        n = np.arange(n_fft, dtype=np.float64)
        win = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / n_fft)                       # torch.hann_window(win_size): periodic
        ang = 2.0 * np.pi * np.outer(np.arange(self.bins, dtype=np.float64), n) / n_fft
        basis = np.concatenate([np.cos(ang) * win, -np.sin(ang) * win], 0).astype(np.float32)      # [2 bins][n_fft]: re rows, then im rows
        self._dft = self.lib.hook(torch.from_numpy(basis).to(self.device).contiguous())
        self.ldm = ops.round_up(self.bins, 32)
        mel = np.zeros((num_mels, self.ldm), dtype=np.float32)
        mel[:, :self.bins] = mel_filterbank(sampling_rate, n_fft, num_mels, fmin, fmax)
        self._mel = self.lib.hook(torch.from_numpy(mel).to(self.device).contiguous())

    @torch.inference_mode()
    def frames(self, y):
        """y [1, L] (or [L]) float waveform in [-1, 1] -> log-mel [T, n_mels] on the device (the `speech_feat[0]` layout of the front end)."""
        lib = self.lib
        y = lib.hook(y.reshape(-1).to(self.device, torch.float32).contiguous())
        L = y.numel()
        T = (L + 2 * self.pad - self.n_fft) // self.hop + 1
        if T <= 0 or L <= self.pad:
            raise ValueError("waveform too short for one frame (%d samples)" % L)
        st = stream_ptr(lib)
        yp = lib.hook(torch.empty(L + 2 * self.pad, dtype=torch.float32, device=self.device))
        lib.cv_reflect_pad(C.c_void_p(y.data_ptr()), C.c_void_p(yp.data_ptr()), C.c_int32(L), C.c_int32(self.pad), st)
        spec = lib.hook(torch.empty(T, 2 * self.bins, dtype=torch.float32, device=self.device))
        ops.gemm_conv(lib, yp, self._dft, self.n_fft, M=T, N=2 * self.bins, K=self.n_fft, lda=self.hop, a_len=yp.numel(), out=spec)
        mag = lib.hook(torch.empty(T, self.ldm, dtype=torch.float32, device=self.device))
        lib.cv_stft_magnitude(C.c_void_p(spec.data_ptr()), C.c_void_p(mag.data_ptr()), C.c_int32(T), C.c_int32(self.bins), C.c_int32(self.ldm), C.c_float(1e-9), st)
        out = lib.hook(torch.empty(T, self.n_mels, dtype=torch.float32, device=self.device))
        ops.gemm_conv(lib, mag, self._mel, self.ldm, M=T, N=self.n_mels, K=self.ldm, out=out, act="logclamp", act_p=1e-5)
        return out

    def __call__(self, y):
        """matcha's contract: [B = 1, L] -> [1, n_mels, T]."""
        return self.frames(y).t().unsqueeze(0)
