"""Kernel-backed CosyVoice-300M, cli.model.CosyVoiceModel.tts(stream=True) (see tests/test_zzz_cosyvoice1_hip.py and cv1k_common.run_model_tts)."""
from cv1k_common import run_model_tts


def test_cosyvoice_model_tts_streaming_matches_reference(lib):
    run_model_tts(lib, True)
