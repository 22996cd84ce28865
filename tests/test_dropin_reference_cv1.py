"""The CosyVoice-300M drop-in exercised INSIDE THE REAL REFERENCE CLASS: a real
`cosyvoice.cli.model.CosyVoiceModel` whose `flow` and `hift` attributes are the kernel-backed stages of cosyvoice_amd/cosyvoice1_hip.py runs its OWN streaming
`tts()` - llm_job thread, 100-token hop + 20-token overlap, flow cache, mel-overlap fade, HiFT mel / source / speech cache, the reference's `fade_in_out` - and
must give the waveform the same class gives around the real `MaskedDiffWithXvec` / `HiFTGenerator` (tests/golden/cv1k_model.npz, made by exactly that run).
Build-container only (needs /root/reference); the product objects run on the CPU emulator of the HIP execution model."""
import os
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))

from cv1k_common import build_flow, build_hift, gold  # noqa: E402

pytestmark = pytest.mark.skipif(not os.path.isdir(os.environ.get("COSYVOICE_REFERENCE", "/root/reference")),
                                reason="the reference tree is only present in the build container")


class _NoSleepTime:
    """Stands in for the `time` module INSIDE cosyvoice.cli.model only (its tts() polls with time.sleep(0.1)).  Assigning to `M.time.sleep` would patch the
    stdlib module for the whole process - and the pytest-xdist worker goes on to run other files (a sleep-based scheduler test failed that way)."""
    sleep = staticmethod(lambda s: None)

    def __getattr__(self, name):
        import time
        return getattr(time, name)


def test_kernel_backed_flow_and_hift_inside_the_real_cosyvoice_model(emu_lib):
    import ref_import
    ref_import.install()
    import cosyvoice.cli.model as M
    M.time = _NoSleepTime()                                        # the reference polls with sleep(0.1)
    g = gold("cv1k_model")
    tokens = g["tokens"].tolist()

    class ScriptedLLM:
        def inference(self, **kw):
            yield from tokens

    m = M.CosyVoiceModel(ScriptedLLM(), build_flow(emu_lib), build_hift(emu_lib))
    assert type(m).__module__ == "cosyvoice.cli.model"
    torch.manual_seed(55)
    with torch.inference_mode():
        chunks = [o["tts_speech"] for o in m.tts(text=torch.zeros(1, 3, dtype=torch.int32), flow_embedding=g["embedding"], llm_embedding=g["embedding"],
                                                flow_prompt_speech_token=g["prompt_token"], prompt_speech_feat=g["prompt_feat"], stream=True)]
    assert [c.shape[1] for c in chunks] == g["stream_n"].tolist() and len(chunks) == 2
    torch.testing.assert_close(torch.cat(chunks, 1), g["stream"], rtol=0, atol=5e-3)
