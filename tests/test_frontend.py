"""on-device 24 kHz prompt log-mel (cosyvoice/cli/frontend.py:120-125) vs the oracle restatement of
matcha.utils.audio.mel_spectrogram (parity unpinned: the Matcha submodule and librosa are absent, see oracle/frontend.py)."""
import math
import os

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


# ---- feature front ends of the speech tokenizer and the CAM++ speaker network (cli/frontend.py:95-118) ----------
from cosyvoice_amd.frontend import KaldiFbank, PromptExtractors, WhisperLogMel, kaldi_mel_banks


def _speechlike(L, sr, seed):
    g = torch.Generator().manual_seed(seed)
    t = torch.arange(L) / float(sr)
    y = 0.25 * torch.sin(2 * np.pi * 180.0 * t) + 0.1 * torch.sin(2 * np.pi * 2300.0 * t * (1 + 0.3 * t)) + 0.02 * torch.randn(L, generator=g) + 0.01
    return y.unsqueeze(0).clamp(-1, 1)


def test_kaldi_mel_banks_anchors():
    """The vectorised bank of the product equals the oracle's loop restatement; closed-form anchors of Kaldi's scale: mel(700 (e - 1)) = 1127,
    the first triangle starts at low_freq = 20 Hz, the last one ends at Nyquist, the appended Nyquist column is zero, and neighbouring
    triangles sum to one between their centres (they are NOT area-normalised, unlike librosa's)."""
    a = kaldi_mel_banks(80, 512, 16000.0, 20.0, 0.0)
    assert a.shape == (80, 257) and (a >= 0).all() and a[:, 256].max() == 0.0
    mel = lambda f: 1127.0 * np.log(1.0 + f / 700.0)
    assert abs(mel(700.0 * (np.e - 1.0)) - 1127.0) < 1e-9
    f = 16000.0 / 512 * np.arange(257)
    assert a[0, f <= 20.0].max() == 0.0 and a[-1, 255] > 0.0
    centre = lambda b: 700.0 * (np.exp((mel(20.0) + (b + 1) * (mel(8000.0) - mel(20.0)) / 81) / 1127.0) - 1.0)
    inner = (f > centre(0)) & (f < centre(79))
    np.testing.assert_allclose(a[:, inner].sum(0), 1.0, atol=1e-5)
    # the oracle builds its bank inside kaldi_fbank; compare through a flat power spectrum instead: log of the row sums
    flat = OFE.kaldi_fbank(torch.zeros(1, 400))                      # silence: DC removal leaves zeros -> every bin sits on the epsilon floor
    np.testing.assert_allclose(flat.numpy(), np.full((1, 80), np.log(np.finfo(np.float32).eps)), rtol=0, atol=0)


def test_whisper_log_mel_matches_oracle(lib):
    L = 16000 * 3 // (10 if lib.emulated else 1) + 77                # not a multiple of the hop
    y = _speechlike(L, 16000, 5)
    got = WhisperLogMel(128, lib=lib)(y).cpu()
    want = OFE.whisper_log_mel(y, 128)
    assert got.shape == want.shape == (1, 128, L // 160)
    # fp32 DFT-as-GEMM vs torch's fp32 FFT: absolute error ~1e-6 of the frame energy per bin, magnified by the log for bins far below the
    # frame's peak.  Measured under the emulator: 4e-6 / 6e-6.  Stated tolerance on the (log10 + 4) / 4 scale: 1e-4 within 5 decades of the
    # utterance maximum, 1e-3 down to the 8-decade clip.
    top = want.max()
    loud = want > top - 5.0 / 4.0
    d = (got - want).abs()
    assert d[loud].max().item() < 1e-4 and d.max().item() < 1e-3
    assert abs(got.min().item() - (top.item() - 2.0)) < 1e-3 or got.min().item() > top.item() - 2.0       # the 8-decade clip, 8 / 4 = 2


def test_kaldi_fbank_matches_oracle(lib):
    L = 16000 * 3 // (10 if lib.emulated else 1) + 123
    y = _speechlike(L, 16000, 6)
    fb = KaldiFbank(80, 16000, lib=lib)
    got = fb(y).cpu()
    want = OFE.kaldi_fbank(y)
    assert got.shape == want.shape == (1 + (L - 400) // 160, 80)
    # same error model as above, natural-log scale; the folded (DC removal, pre-emphasis, window) basis is built in float64.  Measured under the
    # emulator: 8e-6 within 10 nepers of the maximum, 2.4e-4 over everything.
    loud = want > want.max() - 10.0
    d = (got - want).abs()
    assert d[loud].max().item() < 2e-4 and d.max().item() < 3e-3
    cm = fb(y, cmn=True).cpu()
    np.testing.assert_allclose(cm.numpy(), (got - got.mean(dim=0, keepdim=True)).numpy(), atol=2e-5)
    assert cm.mean(dim=0).abs().max().item() < 1e-5
    assert fb(y[:, :399]).shape == (0, 80)                           # shorter than one frame: no frames, like kaldi
    one = fb(y[:, :400]).cpu()
    np.testing.assert_allclose(one.numpy(), want[:1].numpy(), atol=3e-3)


class _FakeSession:
    """Stands in for onnxruntime.InferenceSession (no ONNX graphs offline): records its feeds, returns a fixed result."""
    class _In:
        def __init__(self, name):
            self.name = name

    def __init__(self, names, result):
        self._ins, self._result, self.feeds = [self._In(n) for n in names], result, None

    def get_inputs(self):
        return self._ins

    def run(self, outputs, feeds):
        assert outputs is None
        self.feeds = feeds
        return [self._result]


def test_prompt_extractors_feed_the_sessions_like_the_reference(lib):
    """cli/frontend.py:95-125: what reaches the two ONNX sessions (names, shapes, dtypes, values) and what comes back."""
    y16 = _speechlike(16000 // (4 if lib.emulated else 1), 16000, 7)
    y24 = _speechlike(24000 // (4 if lib.emulated else 1), 24000, 8)
    tok = _FakeSession(["feats", "feats_length"], np.array([[5, 17, 4000]], dtype=np.int64))
    spk = _FakeSession(["input"], np.linspace(-1, 1, 192, dtype=np.float32)[None])
    fe = PromptExtractors(MelSpectrogram(lib=lib), spk, tok, lib=lib)
    speech_token, speech_token_len = fe._extract_speech_token(y16)
    assert speech_token.dtype == torch.int32 and speech_token.tolist() == [[5, 17, 4000]] and speech_token_len.tolist() == [3]
    T = y16.shape[1] // 160
    assert tok.feeds["feats"].shape == (1, 128, T) and tok.feeds["feats"].dtype == np.float32
    assert tok.feeds["feats_length"].dtype == np.int32 and tok.feeds["feats_length"].tolist() == [T]
    np.testing.assert_allclose(tok.feeds["feats"], OFE.whisper_log_mel(y16, 128).numpy(), atol=1e-3)
    emb = fe._extract_spk_embedding(y16)
    assert emb.shape == (1, 192) and emb.dtype == torch.float32
    want = OFE.kaldi_fbank(y16)
    assert spk.feeds["input"].shape == (1,) + tuple(want.shape)
    np.testing.assert_allclose(spk.feeds["input"][0], (want - want.mean(dim=0, keepdim=True)).numpy(), atol=3e-3)
    feat, feat_len = fe._extract_speech_feat(y24)
    assert feat.shape == (1, y24.shape[1] // 480, 80) and feat_len.tolist() == [y24.shape[1] // 480]
    with pytest.raises(AssertionError):
        fe._extract_speech_token(torch.zeros(1, 16000 * 30 + 1))


# ---- load_wav's resampler (utils/file_utils.py:44-50 -> torchaudio.transforms.Resample) ------------------------------------------------
from cosyvoice_amd.frontend import Resample, load_wav, sinc_resample_kernel


@pytest.mark.parametrize("orig,new", [(16000, 24000), (24000, 16000), (44100, 16000), (22050, 24000)])
def test_resample_matches_oracle(lib, orig, new):
    L = (orig // 4 if lib.emulated and orig != 44100 else orig // 10 if lib.emulated else orig) + 37
    y = _speechlike(L, orig, 9)
    got = Resample(orig, new, lib=lib)(y).cpu()
    want = OFE.sinc_resample(y, orig, new)
    assert got.shape == want.shape == (1, -(-new * L // orig))
    torch.testing.assert_close(got, want, rtol=0, atol=2e-6)       # the same fp32 filter taps, summation order only


def test_resample_properties():
    """Closed-form anchors of the filter bank (host only): every output phase is a unit-gain low-pass (taps of a phase sum to ~1), the pass band keeps
    a tone's amplitude and frequency, the stop band (above the lower Nyquist) is suppressed, equal rates are the identity."""
    k, width, orig, new = sinc_resample_kernel(16000, 24000)
    assert (orig, new, k.shape) == (2, 3, (3, 2 * width + 2)) and width == math.ceil(6 * 2 / (2 * 0.99))
    np.testing.assert_allclose(k.sum(1), np.ones(3), atol=2e-3)
    t = torch.arange(16000) / 16000.0
    tone = torch.sin(2 * np.pi * 1000.0 * t).unsqueeze(0)
    up = OFE.sinc_resample(tone, 16000, 24000)
    ref = torch.sin(2 * np.pi * 1000.0 * torch.arange(24000) / 24000.0)
    assert (up[0, 200:-200] - ref[200:-200]).abs().max().item() < 1e-2       # a 6-zero-crossing Hann-windowed sinc: ~0.5 % pass-band ripple
    hi = torch.sin(2 * np.pi * 11000.0 * torch.arange(24000) / 24000.0).unsqueeze(0)       # above the 8 kHz Nyquist of the target
    assert OFE.sinc_resample(hi, 24000, 16000)[0, 200:-200].abs().max().item() < 2e-2
    assert OFE.sinc_resample(tone, 16000, 16000) is tone


def test_load_wav_reads_pcm16_and_resamples(lib, tmp_path):
    import wave
    sr = 22050
    x = (_speechlike(sr // 5, sr, 10)[0] * 32767).round().to(torch.int16)
    p = str(tmp_path / "p.wav")
    with wave.open(p, "wb") as w:
        w.setnchannels(2); w.setsampwidth(2); w.setframerate(sr)
        w.writeframes(torch.stack([x, x], 1).numpy().tobytes())                            # stereo, both channels equal -> the mean is the signal
    mono = (x.float() / 32768.0).unsqueeze(0)
    got = load_wav(p, 16000, lib=lib).cpu()
    torch.testing.assert_close(got, OFE.sinc_resample(mono, sr, 16000), rtol=0, atol=2e-6)
    same = load_wav(p, sr, lib=lib)
    assert torch.equal(same, mono)
    with pytest.raises(AssertionError):
        p8 = str(tmp_path / "p8.wav")
        with wave.open(p8, "wb") as w:
            w.setnchannels(1); w.setsampwidth(2); w.setframerate(8000); w.writeframes(x.numpy().tobytes())
        load_wav(p8, 16000, lib=lib)


def test_front_end_from_a_wav_file_to_a_request(lib, tmp_path):
    """cosyvoice_amd.frontend.CosyVoiceFrontEnd end to end (cli/frontend.py:157-184): a 16-bit WAV prompt at 22.05 kHz -> load_wav's resampler to 16 / 24 kHz on the
    device -> whisper log-mel / Kaldi fbank / 24 kHz prompt mel on the device -> the two (stand-in) ONNX sessions -> a zero-shot request with the forced 2:1 mel / token
    ratio; the same prompt registered in spk2info then serves a request without touching a network."""
    import wave
    from cosyvoice_amd.frontend import CosyVoiceFrontEnd

    class Tok:
        def encode(self, text, allowed_special="all"):
            return [ord(c) for c in text]
    sr = 22050
    x = (_speechlike(sr // 2, sr, 11)[0] * 32767).round().to(torch.int16)
    p = str(tmp_path / "prompt.wav")
    with wave.open(p, "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(sr); w.writeframes(x.numpy().tobytes())
    mono = (x.float() / 32768.0).unsqueeze(0)
    n_tok = 9                                                          # the tokenizer network "returns" 9 tokens; the 24 kHz mel has 0.5 s * 50 = 25 frames
    tok = _FakeSession(["feats", "feats_length"], np.arange(100, 100 + n_tok, dtype=np.int64)[None])
    spk = _FakeSession(["input"], np.linspace(-1, 1, 192, dtype=np.float32)[None])
    fe = CosyVoiceFrontEnd(Tok, campplus_model=spk, speech_tokenizer_model=tok, lib=lib)
    req = fe.frontend_zero_shot("hi there", "prompt", p, 24000, "")
    # (the oracle front ends take the waveform the DEVICE resampler produced - that one is held to oracle.sinc_resample by test_load_wav_reads_pcm16_and_resamples -
    # so that the tolerances below are the ones the extractors' own tests state)
    y16, y24 = load_wav(p, 16000, lib=lib).cpu(), load_wav(p, 24000, lib=lib).cpu()
    torch.testing.assert_close(y24, OFE.sinc_resample(mono, sr, 24000), rtol=0, atol=2e-6)
    np.testing.assert_allclose(tok.feeds["feats"], OFE.whisper_log_mel(y16, 128).numpy(), atol=2e-3)
    assert req["text"].tolist() == [[ord(c) for c in "hi there"]] and req["text_len"].tolist() == [8] and req["prompt_text_len"].tolist() == [6]
    assert req["llm_prompt_speech_token"].tolist() == [list(range(100, 100 + n_tok))] and req["flow_prompt_speech_token_len"].tolist() == [n_tok]
    want_mel = OFE.mel_spectrogram(y24)[0].transpose(0, 1)[: 2 * n_tok]
    assert req["prompt_speech_feat"].shape == (1, 2 * n_tok, 80) and req["prompt_speech_feat_len"].tolist() == [2 * n_tok]
    diff = (req["prompt_speech_feat"][0].cpu() - want_mel).abs()
    loud = want_mel > np.log(1e-3)                                     # the tolerance rule of test_prompt_mel_matches_oracle
    assert diff[loud].max().item() < 2e-3 and diff.max().item() < 5e-2
    assert req["llm_embedding"].shape == (1, 192) and torch.equal(req["llm_embedding"], req["flow_embedding"])
    # a (waveform, rate) pair instead of a path: the same request
    req2 = fe.frontend_zero_shot("hi there", "prompt", (mono, sr), 24000, "")
    assert all(torch.equal(req[k], req2[k]) for k in req)
    # cached speaker: no session is touched again
    fe.add_zero_shot_spk("prompt", p, "me")
    tok.feeds = spk.feeds = None
    req3 = fe.frontend_cross_lingual("hola", "unused.wav", 24000, "me")
    assert tok.feeds is None and spk.feeds is None and "prompt_text" not in req3 and "llm_prompt_speech_token" not in req3
    assert torch.equal(req3["prompt_speech_feat"], req["prompt_speech_feat"]) and req3["text"].tolist() == [[ord(c) for c in "hola"]]
