"""Kernel-backed CosyVoice-300M, cli.model.CosyVoiceModel.tts(stream=True) (see tests/test_zzz_cosyvoice1_hip.py and cv1k_common.run_model_tts).
On the MI355X only: in the build container the same streamed request runs inside the REAL cosyvoice.cli.model.CosyVoiceModel (tests/test_dropin_reference_cv1.py,
same golden), and the emulator needs ~2.5 minutes per run of it."""
import pytest

from cv1k_common import run_model_tts


@pytest.mark.gpu
def test_cosyvoice_model_tts_streaming_matches_reference(hip_lib):
    run_model_tts(hip_lib, True)
