"""Oracle: the prompt log-mel front end (TEST INFRASTRUCTURE - see oracle/__init__.py).  PARITY UNPINNED: `matcha.utils.audio.mel_spectrogram`
(called at cosyvoice/cli/frontend.py:120-125, configured by cosyvoice2.yaml:150-158) lives in the un-vendored Matcha-TTS submodule and
`librosa.filters.mel` (librosa, unpinned in requirements.txt) is not installed; both are restated from their published algorithms (SURVEY.md
Appendix B): reflect pad (n_fft - hop) / 2, torch.stft(center=False, hann_window), sqrt(re^2 + im^2 + 1e-9), Slaney-scale area-normalised
triangular filters, log(clamp(., 1e-5)).  The filterbank here is an independent float64 restatement (loops, not the product's vectorised code);
its closed-form anchors are tested in tests/test_frontend.py.
CROSS-CHECKED (round 5, tests/test_frontend_pinned.py): the Slaney mel basis, the prompt mel, the whisper log-mel and the Kaldi fbank of this file agree with the separate
ports of the same published algorithms in Hugging Face `transformers` (audio_utils.mel_filter_bank / spectrogram, WhisperFeatureExtractor, SeamlessM4TFeatureExtractor -
installed here) to 1e-12 / 2e-4 / 1e-4 / 5e-4.  That is not parity with the reference's own dependencies (still absent), but it is an independent implementation."""
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


# ---- whisper log-mel and kaldi fbank (cosyvoice/cli/frontend.py:98, :109-113) -----------------------------------
# PARITY UNPINNED as well: neither `openai-whisper` nor `torchaudio` is installed here.  Restated from their published sources, step by step in
# the order those sources compute (no folding of the per-frame linear steps, torch.stft / torch.fft.rfft instead of a DFT matrix).
def whisper_log_mel(audio, n_mels=128):
    """whisper/audio.py log_mel_spectrogram: audio [1, L] (16 kHz) -> [1, n_mels, L // 160].  `mel_filters` there loads
    librosa.filters.mel(sr=16000, n_fft=400, n_mels=n_mels) from an .npz asset."""
    stft = torch.stft(audio, 400, 160, window=torch.hann_window(400), return_complex=True)
    magnitudes = stft[..., :-1].abs() ** 2
    filters = torch.from_numpy(librosa_mel(16000, 400, n_mels, 0.0, None)).float()
    log_spec = torch.clamp(filters @ magnitudes, min=1e-10).log10()
    log_spec = torch.maximum(log_spec, log_spec.max() - 8.0)
    return (log_spec + 4.0) / 4.0


def kaldi_fbank(waveform, num_mel_bins=80, sample_frequency=16000.0, frame_length=25.0, frame_shift=10.0, preemphasis=0.97, low_freq=20.0, high_freq=0.0):
    """torchaudio/compliance/kaldi.py fbank with dither=0 and every other argument at its default (snip_edges, remove_dc_offset, povey window,
    round_to_power_of_two, use_power, use_log_fbank, no energy, no vtln): waveform [1, L] -> [m, num_mel_bins]."""
    x = waveform[0]
    win, hop = int(sample_frequency * frame_length * 0.001), int(sample_frequency * frame_shift * 0.001)
    padded = 1 << (win - 1).bit_length()
    if x.numel() < win:
        return torch.empty(0, num_mel_bins)
    m = 1 + (x.numel() - win) // hop
    frames = x.as_strided((m, win), (hop, 1)).clone()
    frames = frames - frames.mean(dim=1, keepdim=True)                                      # remove_dc_offset
    prev = torch.nn.functional.pad(frames.unsqueeze(0), (1, 0), mode="replicate").squeeze(0)
    frames = frames - preemphasis * prev[:, :-1]
    frames = frames * torch.hann_window(win, periodic=False).pow(0.85).unsqueeze(0)        # povey
    frames = torch.nn.functional.pad(frames, (0, padded - win))
    power = torch.fft.rfft(frames).abs().pow(2.0)
    nyq = 0.5 * sample_frequency
    hi = high_freq + nyq if high_freq <= 0.0 else high_freq
    mel = lambda f: 1127.0 * math.log(1.0 + f / 700.0)
    mlo, mhi = mel(low_freq), mel(hi)
    delta = (mhi - mlo) / (num_mel_bins + 1)
    banks = torch.zeros(num_mel_bins, padded // 2 + 1)
    for b in range(num_mel_bins):
        left, center, right = mlo + b * delta, mlo + (b + 1) * delta, mlo + (b + 2) * delta
        for k in range(padded // 2):
            mk = mel(sample_frequency / padded * k)
            banks[b, k] = max(0.0, min((mk - left) / (center - left), (right - mk) / (right - center)))
    e = torch.mm(power, banks.T)
    return torch.max(e, torch.tensor(torch.finfo(torch.float).eps)).log()


def sinc_resample(waveform, orig_freq, new_freq, lowpass_filter_width=6, rolloff=0.99):
    """torchaudio.functional.resample(waveform[1, L], orig_freq, new_freq) with the transform's defaults (sinc_interp_hann), restated from
    torchaudio/functional/functional.py (_get_sinc_resample_kernel + _apply_sinc_resample_kernel).  PARITY UNPINNED (torchaudio is not installed)."""
    if orig_freq == new_freq:
        return waveform
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    base_freq = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base_freq)
    idx = torch.arange(-width, width + orig, dtype=torch.float64)[None, None] / orig
    t = torch.arange(0, -new, -1, dtype=torch.float64)[:, None, None] / new + idx
    t *= base_freq
    t = t.clamp_(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t *= math.pi
    scale = base_freq / orig
    kernels = torch.where(t == 0, torch.tensor(1.0, dtype=torch.float64), t.sin() / t)
    kernels = (kernels * window * scale).float()
    length = waveform.shape[-1]
    x = torch.nn.functional.pad(waveform, (width, width + orig))
    res = torch.nn.functional.conv1d(x[:, None], kernels, stride=orig)
    res = res.transpose(1, 2).reshape(waveform.shape[0], -1)
    return res[..., :math.ceil(new * length / orig)]
