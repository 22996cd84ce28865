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


# -----------------------------------------------------------------------------------------------------------------------------------
# the feature front ends of the two ONNX extractors (cosyvoice/cli/frontend.py:95-118)
# -----------------------------------------------------------------------------------------------------------------------------------
def _dft_basis(n_fft, n_in, kp, op=None):
    """[2 bins][kp] float32: rows = (cos | -sin) of an n_fft-point DFT over the first n_in samples, right-multiplied by the n_in x n_in operator
    `op` (float64; window, pre-emphasis, ...) that the reference applies to every frame before the FFT; columns >= n_in are zero."""
    bins = n_fft // 2 + 1
    ang = 2.0 * np.pi * np.outer(np.arange(bins, dtype=np.float64), np.arange(n_in, dtype=np.float64)) / n_fft
    f = np.concatenate([np.cos(ang), -np.sin(ang)], 0)
    if op is not None:
        f = f @ op
    out = np.zeros((2 * bins, kp), dtype=np.float32)
    out[:, :n_in] = f.astype(np.float32)
    return out


class _FramedSpectrum:
    """Shared device pipeline: frames of `win` samples every `hop` samples of a 1-D signal -> power spectrum of an n_fft-point DFT -> filterbank
    with log(max(., floor)).  The framing is the strided A operand of ONE exact-fp32 MFMA GEMM (lda = hop, K = win padded to 32), the filterbank
    a second one; no FFT library, nothing leaves the device."""

    def _setup(self, lib, n_fft, win, hop, basis_op, bank, floor):
        self.lib = lib or get_lib()
        self.device = torch.device(self.lib.device)
        self.n_fft, self.win, self.hop, self.bins, self.floor = n_fft, win, hop, n_fft // 2 + 1, floor
        self.kp = ops.round_up(win, 32)
        self._dft = self.lib.hook(torch.from_numpy(_dft_basis(n_fft, win, self.kp, basis_op)).to(self.device).contiguous())
        self.ldm = ops.round_up(self.bins, 32)
        self.n_out = bank.shape[0]
        b = np.zeros((self.n_out, self.ldm), dtype=np.float32)
        b[:, :bank.shape[1]] = bank
        self._bank = self.lib.hook(torch.from_numpy(b).to(self.device).contiguous())

    def _log_bank(self, sig, T):
        """sig: device fp32 signal holding at least (T - 1) * hop + kp samples -> [T, n_out] = log(max(bank @ |DFT(frame)|^2, floor))."""
        lib, st = self.lib, stream_ptr(self.lib)
        assert sig.numel() >= (T - 1) * self.hop + self.kp
        spec = lib.hook(torch.empty(T, 2 * self.bins, dtype=torch.float32, device=self.device))
        ops.gemm_conv(lib, sig, self._dft, self.kp, M=T, N=2 * self.bins, K=self.kp, lda=self.hop, a_len=sig.numel(), out=spec)
        pw = lib.hook(torch.empty(T, self.ldm, dtype=torch.float32, device=self.device))
        lib.cv_stft_power(C.c_void_p(spec.data_ptr()), C.c_void_p(pw.data_ptr()), C.c_int32(T), C.c_int32(self.bins), C.c_int32(self.ldm), st)
        out = lib.hook(torch.empty(T, self.n_out, dtype=torch.float32, device=self.device))
        ops.gemm_conv(lib, pw, self._bank, self.ldm, M=T, N=self.n_out, K=self.ldm, out=out, act="logclamp", act_p=self.floor)
        return out


class WhisperLogMel(_FramedSpectrum):
    """`whisper.log_mel_spectrogram(speech, n_mels=128)` as `_extract_speech_token` calls it (cli/frontend.py:98): 16 kHz, n_fft 400, hop 160,
    periodic Hann, torch.stft(center=True) = reflect padding by 200, the last frame dropped, |.|^2, librosa mel bank (Slaney, 128), log10 with a
    1e-10 floor, dynamic range clipped 8 below the utterance maximum, (x + 4) / 4.  Returns [1, n_mels, L // 160] on the device - the
    `feat` the speech-tokenizer session is fed."""

    def __init__(self, n_mels=128, lib=None):
        n = np.arange(400, dtype=np.float64)
        self.n_mels = n_mels
        self._setup(lib, 400, 400, 160, np.diag(0.5 - 0.5 * np.cos(2.0 * np.pi * n / 400)), mel_filterbank(16000, 400, n_mels, 0.0, None), 1e-10)

    @torch.inference_mode()
    def __call__(self, speech):
        lib, st = self.lib, stream_ptr(self.lib)
        y = lib.hook(speech.reshape(-1).to(self.device, torch.float32).contiguous())
        L = y.numel()
        T = L // self.hop                                            # 1 + L // hop frames of the centred STFT, minus the dropped last one
        if T <= 0 or L <= 200:
            raise ValueError("waveform too short (%d samples)" % L)
        yp = lib.hook(torch.zeros(L + 400 + (self.kp - self.win), dtype=torch.float32, device=self.device))
        lib.cv_reflect_pad(C.c_void_p(y.data_ptr()), C.c_void_p(yp.data_ptr()), C.c_int32(L), C.c_int32(200), st)
        ln = self._log_bank(yp, T)
        out = lib.hook(torch.empty(self.n_mels, T, dtype=torch.float32, device=self.device))
        lib.cv_whisper_lognorm(C.c_void_p(ln.data_ptr()), C.c_void_p(out.data_ptr()), C.c_int32(T), C.c_int32(self.n_mels), st)
        return out.unsqueeze(0)


def kaldi_mel_banks(num_bins, padded_window, sample_freq, low_freq, high_freq):
    """torchaudio.compliance.kaldi.get_mel_banks (vtln_warp 1.0): triangles on the HTK-style scale 1127 ln(1 + f / 700), evaluated in the mel
    domain, over the first padded_window / 2 FFT bins; one zero column appended for the Nyquist bin (kaldi.fbank pads it the same way)."""
    nfb = padded_window // 2
    nyq = 0.5 * sample_freq
    if high_freq <= 0.0:
        high_freq += nyq
    mel = lambda f: 1127.0 * np.log(1.0 + np.asarray(f, dtype=np.float64) / 700.0)
    lo, hi = mel(low_freq), mel(high_freq)
    delta = (hi - lo) / (num_bins + 1)
    b = np.arange(num_bins, dtype=np.float64)[:, None]
    left, center, right = lo + b * delta, lo + (b + 1.0) * delta, lo + (b + 2.0) * delta
    m = mel(sample_freq / padded_window * np.arange(nfb, dtype=np.float64))[None, :]
    w = np.maximum(0.0, np.minimum((m - left) / (center - left), (right - m) / (right - center)))
    return np.concatenate([w, np.zeros((num_bins, 1))], 1).astype(np.float32)


class KaldiFbank(_FramedSpectrum):
    """`torchaudio.compliance.kaldi.fbank(speech, num_mel_bins=80, dither=0, sample_frequency=16000)` (cli/frontend.py:109-112) with the
    remaining arguments at their defaults: 25 ms frames every 10 ms with snip_edges, DC removal, pre-emphasis 0.97 (first sample against
    itself), Povey window, zero padding to 512, power spectrum, 80 triangular mel bins from 20 Hz to Nyquist, log with a float32-epsilon floor.
    DC removal, pre-emphasis and window are linear maps of a frame: they are folded (in float64) into the DFT basis, so a frame costs one
    row of the STFT GEMM.  `__call__` -> [T, 80]; `cmn=True` subtracts the mean over frames (frontend.py:113), the CAM++ session's input."""

    def __init__(self, num_mel_bins=80, sample_frequency=16000, frame_length=25.0, frame_shift=10.0, preemphasis_coefficient=0.97,
                 low_freq=20.0, high_freq=0.0, lib=None):
        win, hop = int(sample_frequency * frame_length * 0.001), int(sample_frequency * frame_shift * 0.001)
        n_fft = 1 << (win - 1).bit_length()                         # round_to_power_of_two
        dc = np.eye(win) - np.full((win, win), 1.0 / win)            # remove_dc_offset
        pre = np.eye(win) - preemphasis_coefficient * np.eye(win, k=-1)
        pre[0, 0] -= preemphasis_coefficient                         # replicate padding: x[0] - c x[0]
        povey = (0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(win, dtype=np.float64) / (win - 1))) ** 0.85
        self.num_mel_bins = num_mel_bins
        self._setup(lib, n_fft, win, hop, np.diag(povey) @ pre @ dc, kaldi_mel_banks(num_mel_bins, n_fft, sample_frequency, low_freq, high_freq),
                    float(np.finfo(np.float32).eps))

    @torch.inference_mode()
    def __call__(self, speech, cmn=False):
        lib = self.lib
        y = speech.reshape(-1).to(self.device, torch.float32)
        L = y.numel()
        if L < self.win:
            return torch.empty(0, self.num_mel_bins, device=self.device)          # kaldi: fewer samples than one frame -> no frames
        T = 1 + (L - self.win) // self.hop
        sig = lib.hook(torch.zeros(L + (self.kp - self.win), dtype=torch.float32, device=self.device))
        sig[:L] = y
        out = self._log_bank(sig, T)
        if cmn:
            lib.cv_sub_col_mean(C.c_void_p(out.data_ptr()), C.c_int32(T), C.c_int32(self.num_mel_bins), stream_ptr(lib))
        return out


class PromptExtractors:
    """The three prompt extractors of `CosyVoiceFrontEnd` (cosyvoice/cli/frontend.py:95-125) with their signal processing on the device and the
    two ONNX networks left where the reference has them: `speech_tokenizer_session` / `campplus_session` are onnxruntime.InferenceSession objects
    (anything with `.get_inputs()` and `.run(None, feeds)`) the caller creates exactly as frontend.py:42-48 does.  The methods take the decoded
    waveform ([1, L] float in [-1, 1], 16 kHz for the first two, the flow's rate for the third) instead of a path - `load_wav` (torchaudio file
    decoding + resampling) is the caller's."""

    def __init__(self, feat_extractor, campplus_session, speech_tokenizer_session, lib=None):
        self.lib = lib or get_lib()
        self.device = torch.device(self.lib.device)
        self.feat_extractor, self.campplus_session, self.speech_tokenizer_session = feat_extractor, campplus_session, speech_tokenizer_session
        self.whisper_mel, self.fbank = WhisperLogMel(128, lib=self.lib), KaldiFbank(80, 16000, lib=self.lib)

    def _extract_speech_token(self, speech):
        assert speech.shape[1] / 16000 <= 30, 'do not support extract speech token for audio longer than 30s'
        feat = self.whisper_mel(speech)
        ins = self.speech_tokenizer_session.get_inputs()
        tok = self.speech_tokenizer_session.run(None, {ins[0].name: feat.cpu().numpy(), ins[1].name: np.array([feat.shape[2]], dtype=np.int32)})[0].flatten().tolist()
        speech_token = torch.tensor([tok], dtype=torch.int32).to(self.device)
        return speech_token, torch.tensor([speech_token.shape[1]], dtype=torch.int32).to(self.device)

    def _extract_spk_embedding(self, speech):
        feat = self.fbank(speech, cmn=True)
        emb = self.campplus_session.run(None, {self.campplus_session.get_inputs()[0].name: feat.unsqueeze(dim=0).cpu().numpy()})[0].flatten().tolist()
        return torch.tensor([emb]).to(self.device)

    def _extract_speech_feat(self, speech):
        speech_feat = self.feat_extractor(speech).squeeze(dim=0).transpose(0, 1).to(self.device).unsqueeze(dim=0)
        return speech_feat, torch.tensor([speech_feat.shape[1]], dtype=torch.int32).to(self.device)


# -----------------------------------------------------------------------------------------------------------------------------------
# load_wav's resampling (cosyvoice/utils/file_utils.py:44-50): torchaudio.transforms.Resample(orig_freq, new_freq) with its defaults
# -----------------------------------------------------------------------------------------------------------------------------------
def sinc_resample_kernel(orig_freq, new_freq, lowpass_filter_width=6, rolloff=0.99):
    """torchaudio.functional.resample's `sinc_interp_hann` filter bank (torchaudio/functional/functional.py::_get_sinc_resample_kernel, the
    transform's defaults): after dividing both rates by their gcd, new_freq windowed-sinc filters of 2 * width + orig_freq taps, one per output
    phase, cut off at rolloff x the lower Nyquist.  Returns (float32 [new][taps], width, orig, new) with the reduced rates."""
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    base = min(orig, new) * rolloff
    width = int(math.ceil(lowpass_filter_width * orig / base))
    idx = np.arange(-width, width + orig, dtype=np.float64)[None, :] / orig
    t = (np.arange(0, -new, -1, dtype=np.float64)[:, None] / new + idx) * base
    t = np.clip(t, -lowpass_filter_width, lowpass_filter_width)
    window = np.cos(t * math.pi / lowpass_filter_width / 2.0) ** 2
    t = t * math.pi
    with np.errstate(invalid="ignore", divide="ignore"):
        k = np.where(t == 0.0, 1.0, np.sin(t) / t)
    return (k * window * (base / orig)).astype(np.float32), width, orig, new


class Resample:
    """`torchaudio.transforms.Resample(orig_freq, new_freq)` as `load_wav` applies it to a prompt (file_utils.py:49): pad (width, width + orig),
    a strided convolution with the filter bank, the phases interleaved, cut to ceil(new * L / orig) samples.  On the device that is ONE implicit
    GEMM of the fp32 MFMA kernel: the window of output group n starts at n * orig in the padded signal (lda = orig), the filter bank is W
    [new][taps], and the row-major result [groups][new] IS the interleaved output.  `__call__(waveform[1, L]) -> [1, ceil(new * L / orig)]`."""

    def __init__(self, orig_freq, new_freq, lib=None):
        self.lib = lib or get_lib()
        self.device = torch.device(self.lib.device)
        self.orig_freq, self.new_freq = int(orig_freq), int(new_freq)
        k, self.width, self.orig, self.new = sinc_resample_kernel(orig_freq, new_freq)
        self.taps = k.shape[1]
        self.kp = ops.round_up(self.taps, 32)
        w = np.zeros((self.new, self.kp), dtype=np.float32)
        w[:, :self.taps] = k
        self._w = self.lib.hook(torch.from_numpy(w).to(self.device).contiguous())

    @torch.inference_mode()
    def __call__(self, waveform):
        if self.orig_freq == self.new_freq:
            return waveform.to(self.device, torch.float32)
        lib = self.lib
        y = waveform.reshape(-1).to(self.device, torch.float32)
        L = y.numel()
        groups = (L + self.orig - 1) // self.orig + 1               # conv1d output length over the padded signal: L // orig + 1 (+ a partly covered one)
        target = -(-self.new * L // self.orig)                      # ceil(new * L / orig)
        groups = max(groups, -(-target // self.new))
        sig = lib.hook(torch.zeros(self.width + (groups - 1) * self.orig + self.kp, dtype=torch.float32, device=self.device))
        sig[self.width:self.width + L] = y
        out = lib.hook(torch.empty(groups, self.new, dtype=torch.float32, device=self.device))
        ops.gemm_conv(lib, sig, self._w, self.kp, M=groups, N=self.new, K=self.kp, lda=self.orig, a_len=sig.numel(), out=out)
        return out.reshape(1, -1)[:, :target]


def load_wav(wav, target_sr, min_sr=16000, lib=None):
    """cosyvoice/utils/file_utils.py:44-50 for 16-bit PCM WAV files (the reference decodes with torchaudio + soundfile, absent here; other formats
    are the caller's to decode): channel mean, then `Resample(sample_rate, target_sr)` on the device when the rates differ."""
    import wave
    with wave.open(wav, "rb") as w:
        assert w.getsampwidth() == 2, "16-bit PCM WAV expected"
        sample_rate = w.getframerate()
        data = np.frombuffer(w.readframes(w.getnframes()), dtype=np.int16).astype(np.float32) / 32768.0
        if w.getnchannels() > 1:
            data = data.reshape(-1, w.getnchannels()).mean(1)
    speech = torch.from_numpy(data).unsqueeze(0)
    if sample_rate != target_sr:
        assert sample_rate >= min_sr, 'wav sample rate {} must be greater than {}'.format(sample_rate, target_sr)
        speech = Resample(sample_rate, target_sr, lib=lib)(speech)
    return speech


# -----------------------------------------------------------------------------------------------------------------------------------
# Request assembly: what `CosyVoiceFrontEnd.frontend_*` hand to `CosyVoice2Model.tts(**model_input)`
# -----------------------------------------------------------------------------------------------------------------------------------
_LLM_SPEECH_PROMPT = ("llm_prompt_speech_token", "llm_prompt_speech_token_len")
_TEXT_PROMPT = ("prompt_text", "prompt_text_len")
# what each prompt-driven mode REMOVES from the zero-shot request (cli/frontend.py:195-214): cross-lingual synthesis keeps no prompt inside the LLM at all, instruct2
# keeps the instruction as prompt text but no speech prompt
_MODE_DROPS = {"zero_shot": (), "cross_lingual": _TEXT_PROMPT + _LLM_SPEECH_PROMPT, "instruct2": _LLM_SPEECH_PROMPT}


def _ort_session(model, providers):
    """A path becomes an onnxruntime session configured as cli/frontend.py:42-48 does; a session-like object (`.get_inputs()`, `.run(None, feeds)`) is used as it is;
    None stays None (speakers then come from spk2info only)."""
    if model is None or not isinstance(model, (str, bytes)):
        return model
    try:
        import onnxruntime
    except ImportError as e:
        raise RuntimeError("cosyvoice_amd.frontend: %r needs onnxruntime for the speech tokenizer / CAM++ networks (they stay ONNX sessions, as in the reference); "
                           "pass a session object, or register speakers through spk2info" % model) from e
    opt = onnxruntime.SessionOptions()
    opt.graph_optimization_level = onnxruntime.GraphOptimizationLevel.ORT_ENABLE_ALL
    opt.intra_op_num_threads = 1
    return onnxruntime.InferenceSession(model, sess_options=opt, providers=providers)


class CosyVoiceFrontEnd:
    """Host mirror of `cosyvoice.cli.frontend.CosyVoiceFrontEnd` for everything between a tokenizer and `model.tts`: the three prompt extractors with their signal
    processing on the device (PromptExtractors above; the two networks stay the ONNX sessions the reference uses), the speaker cache `spk2info` (cli/frontend.py:49-52 -
    a cached speaker needs no network at all) and the request assembly of frontend_sft / _zero_shot / _cross_lingual / _instruct / _instruct2 / _vc
    (cli/frontend.py:157-224).  Same method names, arguments and model_input keys as the reference; pinned by tests/golden/frontend_requests.npz, which the REAL class
    produced (tests/golden/make_golden_frontend.py).

    Not here (out of scope): text normalisation.  `text_normalize` passes text through exactly where the reference does without a normaliser
    (generators, SSML-tagged text, text_frontend=False, '') and otherwise calls `text_normalizer(text, split)` if one was given.
    A prompt is a path to a 16-bit PCM WAV (load_wav above) or a `(waveform[1, L], sample_rate)` pair already in memory."""

    def __init__(self, get_tokenizer, feat_extractor=None, campplus_model=None, speech_tokenizer_model=None, spk2info="", allowed_special="all", lib=None,
                 text_normalizer=None):
        import os
        self.lib = lib or get_lib()
        self.device = torch.device(self.lib.device)
        self.tokenizer = get_tokenizer()
        self.feat_extractor = feat_extractor if feat_extractor is not None else MelSpectrogram(lib=self.lib)
        self.campplus_session = _ort_session(campplus_model, ["CPUExecutionProvider"])
        self.speech_tokenizer_session = _ort_session(speech_tokenizer_model, ["ROCMExecutionProvider", "CPUExecutionProvider"])
        self._extractors = PromptExtractors(self.feat_extractor, self.campplus_session, self.speech_tokenizer_session, lib=self.lib)
        if isinstance(spk2info, dict):
            self.spk2info = spk2info
        else:
            self.spk2info = torch.load(spk2info, map_location=self.device, weights_only=True) if spk2info and os.path.exists(spk2info) else {}
        self.allowed_special = allowed_special
        self.text_normalizer = text_normalizer

    # ---- extractors (cli/frontend.py:86-125)
    def _extract_text_token(self, text):
        from typing import Generator
        if isinstance(text, Generator):                          # streamed text: ids one by one, and a dummy length (frontend.py:87-90)
            return self._extract_text_token_generator(text), torch.tensor([0], dtype=torch.int32).to(self.device)
        ids = torch.tensor([self.tokenizer.encode(text, allowed_special=self.allowed_special)], dtype=torch.int32).to(self.device)
        return ids, torch.tensor([ids.shape[1]], dtype=torch.int32).to(self.device)

    def _extract_text_token_generator(self, text_generator):
        for piece in text_generator:
            ids, _ = self._extract_text_token(piece)
            for i in range(ids.shape[1]):
                yield ids[:, i:i + 1]

    def _speech(self, prompt_wav, rate):
        if isinstance(prompt_wav, (tuple, list)):
            wave_, sr = prompt_wav
            return wave_ if int(sr) == rate else Resample(int(sr), rate, lib=self.lib)(wave_)
        return load_wav(prompt_wav, rate, lib=self.lib)

    def _need(self, session, what):
        if session is None:
            raise RuntimeError("cosyvoice_amd.frontend: no %s session - give the constructor the .onnx path or a session object, or use a speaker cached in spk2info" % what)

    def _extract_speech_token(self, prompt_wav):
        self._need(self.speech_tokenizer_session, "speech tokenizer")
        return self._extractors._extract_speech_token(self._speech(prompt_wav, 16000))

    def _extract_spk_embedding(self, prompt_wav):
        self._need(self.campplus_session, "CAM++")
        return self._extractors._extract_spk_embedding(self._speech(prompt_wav, 16000))

    def _extract_speech_feat(self, prompt_wav):
        return self._extractors._extract_speech_feat(self._speech(prompt_wav, 24000))       # (24 kHz whatever the model: frontend.py:121)

    def text_normalize(self, text, split=True, text_frontend=True):
        from typing import Generator
        if isinstance(text, Generator):
            return [text]
        if ("<|" in text and "|>" in text) or text_frontend is False or text == "" or self.text_normalizer is None:
            return [text] if split is True else text
        return self.text_normalizer(text.strip(), split)

    # ---- requests (cli/frontend.py:157-224)
    def _prompt_fields(self, prompt_text, prompt_wav, resample_rate):
        """Everything a request takes from a prompt recording: text ids, mel, speech tokens (the same ids prompt the LLM and the flow), x-vector."""
        text, text_len = self._extract_text_token(prompt_text)
        feat, feat_len = self._extract_speech_feat(prompt_wav)
        tok, tok_len = self._extract_speech_token(prompt_wav)
        if resample_rate == 24000:                              # 25 Hz tokens, 50 Hz mel: exactly two frames per token (frontend.py:169-173)
            n = min(int(feat.shape[1] / 2), tok.shape[1])
            feat, tok = feat[:, :2 * n], tok[:, :n]
            feat_len[:], tok_len[:] = 2 * n, n
        emb = self._extract_spk_embedding(prompt_wav)
        return {"prompt_text": text, "prompt_text_len": text_len, "llm_prompt_speech_token": tok, "llm_prompt_speech_token_len": tok_len,
                "flow_prompt_speech_token": tok, "flow_prompt_speech_token_len": tok_len, "prompt_speech_feat": feat, "prompt_speech_feat_len": feat_len,
                "llm_embedding": emb, "flow_embedding": emb}

    def _prompted(self, mode, tts_text, prompt_text, prompt_wav, resample_rate, zero_shot_spk_id):
        # the text is tokenised first, as in the reference: with streamed text the generator is created before the prompt is read
        text, text_len = self._extract_text_token(tts_text)
        req = self._prompt_fields(prompt_text, prompt_wav, resample_rate) if zero_shot_spk_id == "" else dict(self.spk2info[zero_shot_spk_id])
        req["text"], req["text_len"] = text, text_len
        for k in _MODE_DROPS[mode]:
            del req[k]
        return req

    def frontend_sft(self, tts_text, spk_id):
        text, text_len = self._extract_text_token(tts_text)
        emb = self.spk2info[spk_id]["embedding"]
        return {"text": text, "text_len": text_len, "llm_embedding": emb, "flow_embedding": emb}

    def frontend_zero_shot(self, tts_text, prompt_text, prompt_wav, resample_rate, zero_shot_spk_id):
        return self._prompted("zero_shot", tts_text, prompt_text, prompt_wav, resample_rate, zero_shot_spk_id)

    def frontend_cross_lingual(self, tts_text, prompt_wav, resample_rate, zero_shot_spk_id):
        return self._prompted("cross_lingual", tts_text, "", prompt_wav, resample_rate, zero_shot_spk_id)

    def frontend_instruct(self, tts_text, spk_id, instruct_text):
        req = self.frontend_sft(tts_text, spk_id)
        del req["llm_embedding"]                                # the speaker vector would leak into an instructed LLM (frontend.py:203-204)
        req["prompt_text"], req["prompt_text_len"] = self._extract_text_token(instruct_text)
        return req

    def frontend_instruct2(self, tts_text, instruct_text, prompt_wav, resample_rate, zero_shot_spk_id):
        return self._prompted("instruct2", tts_text, instruct_text, prompt_wav, resample_rate, zero_shot_spk_id)

    def frontend_vc(self, source_speech_16k, prompt_wav, resample_rate):
        tok, tok_len = self._extract_speech_token(prompt_wav)
        feat, feat_len = self._extract_speech_feat(prompt_wav)
        emb = self._extract_spk_embedding(prompt_wav)
        src, src_len = self._extract_speech_token(source_speech_16k)
        return {"source_speech_token": src, "source_speech_token_len": src_len, "flow_prompt_speech_token": tok, "flow_prompt_speech_token_len": tok_len,
                "prompt_speech_feat": feat, "prompt_speech_feat_len": feat_len, "flow_embedding": emb}

    # ---- the speaker cache (cli/cosyvoice.py:65-78 keeps these three on the model class; they only touch the front end's spk2info)
    def list_available_spks(self):
        return list(self.spk2info.keys())

    def add_zero_shot_spk(self, prompt_text, prompt_wav, zero_shot_spk_id, resample_rate=24000):
        assert zero_shot_spk_id != "", "do not use empty zero_shot_spk_id"
        entry = self._prompt_fields(prompt_text, prompt_wav, resample_rate)
        self.spk2info[zero_shot_spk_id] = entry
        return True

    def save_spkinfo(self, path):
        torch.save(self.spk2info, path)
