"""cosyvoice_amd.frontend.CosyVoiceFrontEnd (request assembly + speaker cache) against the model_input dicts the REAL
cosyvoice.cli.frontend.CosyVoiceFrontEnd assembled from the same extractor outputs (tests/golden/make_golden_frontend.py, tests/frontend_fakes.py): same keys, same
dtypes, same values on every path - sft, zero-shot (the forced 2:1 mel / token ratio at 24 kHz in its three length cases, none at 22.05 kHz, a cached speaker),
cross-lingual, instruct, instruct2, voice conversion, streamed text - and an untouched cache entry afterwards.  CPU only; the extractors' own arithmetic is covered by
tests/test_frontend.py / test_frontend_pinned.py."""
import json
import os

import numpy as np
import pytest
import torch

import frontend_fakes as FK
from cosyvoice_amd import frontend as FE

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class _NoLib:
    """The request assembly needs no kernel: a stand-in for the library handle (device = cpu) keeps this test free of the emulator build."""
    device = "cpu"
    emulated = True

    def hook(self, t):
        return t


def mirror(**kw):
    fe = FE.CosyVoiceFrontEnd(FK.FakeTokenizer, feat_extractor=lambda speech: None, lib=_NoLib(), **kw)
    return FK.install(fe)


def test_requests_match_the_real_front_end():
    gold = np.load(os.path.join(G, "frontend_requests.npz"))
    meta = json.load(open(os.path.join(G, "frontend_requests.json")))
    got = FK.cases(mirror(), lambda fe, text, wav, spk: fe.add_zero_shot_spk(text, wav, spk))
    assert sorted(got) == sorted(meta["keys"])
    for case, d in got.items():
        assert sorted(d) == meta["keys"][case], case
        for k, v in d.items():
            want = torch.from_numpy(gold[case + "/" + k])
            assert str(v.dtype).replace("torch.", "") == meta["dtypes"][case + "/" + k] and v.shape == want.shape, (case, k)
            assert torch.equal(v, want), (case, k)
    # the three 24 kHz length cases really are three cases
    assert got["zero_shot_24k_a"]["prompt_speech_feat"].shape[1] == 54 and got["zero_shot_24k_b"]["prompt_speech_feat"].shape[1] == 40
    assert got["zero_shot_24k_c"]["llm_prompt_speech_token"].shape[1] == 25 and got["zero_shot_22k_b"]["prompt_speech_feat"].shape[1] == 43
    assert meta["text_normalize"] == FK.normalize_cases(mirror())


def test_speaker_cache_round_trip_and_missing_networks(tmp_path):
    fe = mirror()
    fe.add_zero_shot_spk("a cached prompt", "b.wav", "cached")
    assert fe.list_available_spks() == ["cached"]
    fe.save_spkinfo(str(tmp_path / "spk2info.pt"))
    again = FE.CosyVoiceFrontEnd(FK.FakeTokenizer, feat_extractor=lambda speech: None, spk2info=str(tmp_path / "spk2info.pt"), lib=_NoLib())     # no extractor stand-ins: the cache alone
    a = again.frontend_zero_shot("other text", "ignored", "ignored.wav", 24000, "cached")
    b = fe.frontend_zero_shot("other text", "ignored", "ignored.wav", 24000, "cached")
    assert sorted(a) == sorted(b) and all(torch.equal(a[k], b[k]) for k in a)
    with pytest.raises(RuntimeError, match="speech tokenizer"):                 # an uncached speaker needs the networks, and says so
        again._extract_speech_token((torch.zeros(1, 16000), 16000))
    with pytest.raises(KeyError):
        again.frontend_sft("text", "nobody")
    with pytest.raises(AssertionError):
        again.add_zero_shot_spk("p", "a.wav", "")
    assert FE.CosyVoiceFrontEnd(FK.FakeTokenizer, feat_extractor=lambda s: None, spk2info=str(tmp_path / "absent.pt"), lib=_NoLib()).spk2info == {}
    norm = FE.CosyVoiceFrontEnd(FK.FakeTokenizer, feat_extractor=lambda s: None, lib=_NoLib(), text_normalizer=lambda text, split: [text.upper()] if split else text.upper())
    assert norm.text_normalize("  hello ", split=True) == ["HELLO"] and norm.text_normalize("<|en|>hi", split=False) == "<|en|>hi"
