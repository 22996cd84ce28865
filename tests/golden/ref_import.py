"""Import the *real* reference (/root/reference, read-only) in this container to generate golden vectors.

Only tests/golden/make_golden.py uses this module, and only here (the GPU box has no /root/reference).
The reference needs a few third-party packages that are not installed; they are replaced by minimal stubs:
  torchaudio, onnxruntime, omegaconf  - imported at module import time, never called on the hot path
  x_transformers                      - not installed; RotaryEmbedding / apply_rotary_pos_emb restated in xtransformers_stub.py (published 2.x code)
  matcha.*                            - un-vendored submodule (third_party/Matcha-TTS is empty).  The classes the
                                        reference imports are RESTATED in matcha_stub.py from SURVEY.md Appendix B
                                        (upstream Matcha-TTS + diffusers 0.29 Attention); that restatement is the one
                                        part of the golden vectors that is not pinned by upstream code.
"""
import importlib.machinery
import os
import sys
import types

REFERENCE = os.environ.get("COSYVOICE_REFERENCE", "/root/reference")


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__path__ = []
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install():
    if not os.path.isdir(REFERENCE):
        raise RuntimeError("reference tree %s not present (golden vectors can only be regenerated in the build container)" % REFERENCE)
    if REFERENCE not in sys.path:
        sys.path.insert(0, REFERENCE)
    if "torchaudio" not in sys.modules:
        ta = _stub("torchaudio")
        _stub("torchaudio.compliance")
        ta.compliance = sys.modules["torchaudio.compliance"]
        _stub("torchaudio.compliance.kaldi")
        ta.compliance.kaldi = sys.modules["torchaudio.compliance.kaldi"]
    if "onnxruntime" not in sys.modules:
        _stub("onnxruntime")
    if "omegaconf" not in sys.modules:
        class DictConfig(dict):
            def __init__(self, content=None, **kw):
                super().__init__(content or {}, **kw)
            __getattr__ = dict.__getitem__
        _stub("omegaconf", DictConfig=DictConfig)
    if "x_transformers" not in sys.modules:
        here = os.path.dirname(os.path.abspath(__file__))
        if here not in sys.path:
            sys.path.insert(0, here)
        import xtransformers_stub
        _stub("x_transformers")
        _stub("x_transformers.x_transformers", RotaryEmbedding=xtransformers_stub.RotaryEmbedding, apply_rotary_pos_emb=xtransformers_stub.apply_rotary_pos_emb)
    if "matcha" not in sys.modules:
        here = os.path.dirname(os.path.abspath(__file__))
        if here not in sys.path:
            sys.path.insert(0, here)
        import matcha_stub
        _stub("matcha")
        _stub("matcha.models")
        _stub("matcha.models.components")
        _stub("matcha.models.components.flow_matching", BASECFM=matcha_stub.BASECFM)
        _stub("matcha.models.components.decoder", SinusoidalPosEmb=matcha_stub.SinusoidalPosEmb, Block1D=matcha_stub.Block1D,
              ResnetBlock1D=matcha_stub.ResnetBlock1D, Downsample1D=matcha_stub.Downsample1D,
              TimestepEmbedding=matcha_stub.TimestepEmbedding, Upsample1D=matcha_stub.Upsample1D)
        _stub("matcha.models.components.transformer", BasicTransformerBlock=matcha_stub.BasicTransformerBlock)
