"""Deterministic stand-ins for the parts of `CosyVoiceFrontEnd` that need files, a tokenizer vocabulary or the two ONNX networks - shared by
tests/golden/make_golden_frontend.py (which hangs them on the REAL cosyvoice.cli.frontend.CosyVoiceFrontEnd) and tests/test_frontend_requests.py (which hangs them on
cosyvoice_amd.frontend.CosyVoiceFrontEnd), so that both assemble requests from identical extractor outputs.  Test infrastructure."""
import zlib

import torch

# prompt "files": name -> (speech tokens the tokenizer network would return, mel frames the feature extractor would return)
WAVS = {"a.wav": (27, 54), "b.wav": (20, 43), "c.wav": (30, 50), "src.wav": (12, 24)}


class FakeTokenizer:
    def encode(self, text, allowed_special="all"):
        assert allowed_special == "all"
        return [ord(ch) % 251 + 1 for ch in text]


def _gen(name, salt):
    return torch.Generator().manual_seed(zlib.crc32((name + salt).encode()))


def speech_token(name):
    n = WAVS[name][0]
    tok = torch.randint(0, 6561, (1, n), generator=_gen(name, "tok"), dtype=torch.int32)
    return tok, torch.tensor([n], dtype=torch.int32)


def speech_feat(name):
    n = WAVS[name][1]
    return torch.randn(1, n, 80, generator=_gen(name, "feat")), torch.tensor([n], dtype=torch.int32)


def spk_embedding(name):
    return torch.randn(1, 192, generator=_gen(name, "emb"))


def install(fe):
    """Shadow the three extractors of a front-end object (the real class or the mirror) with the stand-ins above."""
    fe._extract_speech_token = speech_token
    fe._extract_speech_feat = speech_feat
    fe._extract_spk_embedding = spk_embedding
    return fe


def cases(fe, add_zero_shot_spk):
    """Every request-assembly path, in a fixed order: {case name: model_input dict}.  `add_zero_shot_spk(fe, prompt_text, prompt_wav, spk_id)` registers a speaker the
    way cosyvoice/cli/cosyvoice.py:69-75 does."""
    fe.spk2info = {"spkA": {"embedding": torch.randn(1, 192, generator=_gen("spkA", "emb"))}}
    add_zero_shot_spk(fe, "a cached prompt", "b.wav", "cached")
    out = {}
    out["sft"] = fe.frontend_sft("hello world", "spkA")
    for w in ("a.wav", "b.wav", "c.wav"):                                       # equal lengths / mel longer than 2 x tokens / tokens longer than mel / 2
        out["zero_shot_24k_" + w[0]] = fe.frontend_zero_shot("some text to say", "prompt words", w, 24000, "")
    out["zero_shot_22k_b"] = fe.frontend_zero_shot("some text to say", "prompt words", "b.wav", 22050, "")     # no forced 2:1 ratio at the CosyVoice-300M rate
    out["zero_shot_cached"] = fe.frontend_zero_shot("other text", "ignored", "ignored.wav", 24000, "cached")
    out["cross_lingual"] = fe.frontend_cross_lingual("bonjour", "c.wav", 24000, "")
    out["cross_lingual_cached"] = fe.frontend_cross_lingual("bonjour", "ignored.wav", 24000, "cached")
    out["instruct"] = fe.frontend_instruct("say it", "spkA", "speak slowly<|endofprompt|>")
    out["instruct2"] = fe.frontend_instruct2("say it", "be happy<|endofprompt|>", "a.wav", 24000, "")
    out["instruct2_cached"] = fe.frontend_instruct2("say it", "be happy<|endofprompt|>", "ignored.wav", 24000, "cached")
    out["vc"] = fe.frontend_vc("src.wav", "a.wav", 24000)
    # the cached speaker's own entry must not have been edited by the requests above (the reference copies it per request)
    out["spk2info_cached_after"] = dict(fe.spk2info["cached"])
    # streamed text: `text` becomes a generator of [1, 1] id tensors, `text_len` a dummy 0 (frontend.py:86-101)
    gen_in = (s for s in ("ab", "cde"))
    m = fe.frontend_zero_shot(gen_in, "prompt words", "a.wav", 24000, "")
    m["text"] = torch.cat(list(m["text"]), 1)
    out["zero_shot_text_generator"] = m
    return out


def normalize_cases(fe):
    """text_normalize paths that do not depend on an installed normaliser (frontend.py:127-133)."""
    g = (s for s in ("x",))
    r = fe.text_normalize(g, split=True)
    return {"generator": [r == [g]], "ssml_split": fe.text_normalize("<|en|>hello there", split=True), "ssml_nosplit": fe.text_normalize("<|en|>hello there", split=False),
            "off_split": fe.text_normalize("plain text 123", split=True, text_frontend=False), "off_nosplit": fe.text_normalize("plain text 123", split=False, text_frontend=False),
            "empty": fe.text_normalize("", split=True)}
