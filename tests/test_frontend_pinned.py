"""The front-end oracles (oracle/frontend.py) against INDEPENDENT implementations of the same published algorithms.

The reference computes its prompt features with third-party code that is absent from the reference tree and from this image: `matcha.utils.audio.mel_spectrogram` +
`librosa.filters.mel` (cosyvoice/cli/frontend.py:120-125), `whisper.log_mel_spectrogram` (:98) and `torchaudio.compliance.kaldi.fbank` (:109-113).  The oracle restates
them; nothing of the reference can pin that restatement.  What IS installed (offline wheelhouse) is Hugging Face `transformers`, whose `audio_utils` / feature extractors
are separate ports of the same three algorithms (librosa's Slaney mel basis, Whisper's log-mel, torchaudio's Kaldi fbank).  Agreement with them is not parity with the
reference's own dependencies - those stay unavailable - but it rules out a private misreading of the published algorithms shared by the oracle and the kernels."""
import numpy as np
import pytest
import torch

from oracle import frontend as OFE

tf_audio = pytest.importorskip("transformers.audio_utils")


@pytest.mark.parametrize("sr,n_fft,n_mels,fmin,fmax", [(24000, 1920, 80, 0.0, 8000.0),        # the CosyVoice2 prompt mel (cosyvoice2.yaml:150-158)
                                                       (22050, 1024, 80, 0.0, 8000.0),        # CosyVoice-300M's
                                                       (16000, 400, 128, 0.0, 8000.0)])       # whisper's mel_filters asset
def test_slaney_mel_basis_equals_the_transformers_port(sr, n_fft, n_mels, fmin, fmax):
    ours = OFE.librosa_mel(sr, n_fft, n_mels, fmin, fmax)
    theirs = tf_audio.mel_filter_bank(num_frequency_bins=n_fft // 2 + 1, num_mel_filters=n_mels, min_frequency=fmin, max_frequency=fmax, sampling_rate=sr,
                                      norm="slaney", mel_scale="slaney").T
    assert ours.shape == theirs.shape
    assert np.abs(ours - theirs).max() < 1e-12


def test_prompt_mel_equals_a_transformers_spectrogram():
    """matcha's mel_spectrogram = reflect pad (n_fft - hop) / 2, periodic Hann STFT without centring, magnitude, Slaney mel, log(clamp 1e-5) - rebuilt from
    transformers.audio_utils.spectrogram (its magnitude has no + 1e-9 under the root: hence the tolerance)."""
    g = torch.Generator().manual_seed(3)
    y = torch.randn(1, 24000 + 123, generator=g) * 0.2
    ours = OFE.mel_spectrogram(y)[0].numpy()
    pad = (1920 - 480) // 2
    yp = np.pad(y[0].numpy().astype(np.float64), (pad, pad), mode="reflect")
    filt = tf_audio.mel_filter_bank(num_frequency_bins=961, num_mel_filters=80, min_frequency=0.0, max_frequency=8000.0, sampling_rate=24000, norm="slaney", mel_scale="slaney")
    theirs = tf_audio.spectrogram(yp, tf_audio.window_function(1920, "hann"), frame_length=1920, hop_length=480, fft_length=1920, power=1.0, center=False,
                                  mel_filters=filt, mel_floor=1e-5, log_mel="log")
    assert ours.shape == theirs.shape
    assert np.abs(ours - theirs).max() < 2e-4


def test_whisper_log_mel_equals_the_transformers_extractor():
    transformers = pytest.importorskip("transformers")
    fe = transformers.WhisperFeatureExtractor(feature_size=128)
    g = torch.Generator().manual_seed(0)
    audio = torch.randn(1, 16000 * 3 + 37, generator=g) * 0.1
    ours = OFE.whisper_log_mel(audio)[0].numpy()
    theirs = fe._np_extract_fbank_features(audio.numpy(), "cpu")[0]
    assert ours.shape == theirs.shape
    assert np.abs(ours - theirs).max() < 1e-4


def test_kaldi_fbank_equals_the_transformers_port():
    """SeamlessM4TFeatureExtractor ports torchaudio.compliance.kaldi.fbank (povey window, pre-emphasis 0.97, DC removal, power spectrum over 512 points, 80 Kaldi-scale
    triangles from 20 Hz, log with the float32 epsilon floor).  It scales the waveform to 16-bit range first, so the oracle gets the scaled waveform."""
    transformers = pytest.importorskip("transformers")
    fk = transformers.SeamlessM4TFeatureExtractor(feature_size=80, num_mel_bins=80, sampling_rate=16000)
    g = torch.Generator().manual_seed(1)
    w = torch.randn(1, 16000 * 2 + 11, generator=g) * 0.1
    ours = OFE.kaldi_fbank(w * 32768.0).numpy()
    theirs = fk._extract_fbank_features(w[0].numpy())
    assert ours.shape == theirs.shape
    assert np.abs(ours - theirs).max() < 5e-4 and np.abs(ours).max() > 10
