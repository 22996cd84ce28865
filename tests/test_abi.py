"""C-ABI surface: every entry point declared in include/cosyvoice_amd.h is exported by the gfx950 library (loadable without a
GPU), and the product loader has no CPU fallback."""
import ctypes
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(REPO, "include", "cosyvoice_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cv_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_boundary():
    syms = declared_symbols()
    for must in ("cv_llm_prefill", "cv_llm_decode", "cv_flow_estimator", "cv_flow_encoder", "cv_flow_inference", "cv_hift_inference",
                 "cv_hift_decode", "cv_gemm_conv", "cv_attention", "cv_norm_rows", "cv_fade_in_out", "cv_last_error"):
        assert must in syms
    assert len(syms) >= 35


def test_hip_library_exports_every_declared_symbol():
    from cosyvoice_amd.build import build_hip
    lib = ctypes.CDLL(build_hip())                      # cross-compiled for gfx950; loading needs no GPU
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing
    lib.cv_is_emulated.restype = ctypes.c_int
    assert lib.cv_is_emulated() == 0
    lib.cv_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.cv_version()


def test_emulator_library_exports_the_same_surface(emu_lib):
    missing = [s for s in declared_symbols() if not hasattr(emu_lib.dll, s)]
    assert not missing, missing
    assert emu_lib.emulated


def test_no_cpu_fallback(emu_lib, tmp_path):
    from cosyvoice_amd._lib import CosyVoiceAmdError, Lib
    with pytest.raises(CosyVoiceAmdError, match="not found"):
        Lib(str(tmp_path / "libcosyvoice_amd.so"))
    with pytest.raises(CosyVoiceAmdError, match="emulator"):
        Lib(emu_lib.path)                               # the product loader refuses the emulator build


def test_errors_are_reported_not_aborted(emu_lib):
    from cosyvoice_amd._lib import CosyVoiceAmdError
    with pytest.raises(CosyVoiceAmdError, match="null|bad"):
        emu_lib.cv_flow_estimator(None, None, None, None, None, None, None, ctypes.c_int32(4), ctypes.c_int32(0), None, None)


def test_lm1_entry_points_check_their_arguments(emu_lib):
    """cv_lm1_* (round 4): a configuration outside what the fused decode step serves returns NULL with a message; stepping an unbound handle, a position beyond the bound
    cache, tables shorter than the cache and an unaligned input row are errors, not memory faults."""
    import torch
    from cosyvoice_amd._lib import ACT, CosyVoiceAmdError, Lm1Config, Lm1LayerWeights
    create = emu_lib.raw("cv_lm1_create", ctypes.c_void_p)
    d, ffn, n_out = 64, 128, 10
    keep = []

    def buf(*shape):
        t = torch.zeros(*shape, dtype=torch.float32)
        keep.append(t)
        return t.data_ptr()
    lw = (Lm1LayerWeights * 1)()
    for name, shape in (("ln1_g", (d,)), ("ln1_b", (d,)), ("w_qkv", (4 * d, d)), ("b_qkv", (4 * d,)), ("w_out", (d, d)), ("b_out", (d,)), ("ln2_g", (d,)), ("ln2_b", (d,)),
                        ("w1", (ffn, d)), ("b1", (ffn,)), ("w2", (d, ffn)), ("b2", (d,))):
        setattr(lw[0], name, buf(*shape))
    c = Lm1Config()
    c.n_layers, c.d, c.heads, c.ffn, c.d_in, c.n_out, c.act, c.xscale = 1, d, 1, ffn, d, n_out, ACT["relu"], 8.0
    for name, shape in (("embed_w", (d, d)), ("embed_b", (d,)), ("embed_g", (d,)), ("embed_beta", (d,)), ("after_g", (d,)), ("after_b", (d,)), ("dec_w", (n_out, d)), ("dec_b", (n_out,))):
        setattr(c, name, buf(*shape))
    bad = Lm1Config.from_buffer_copy(c)
    bad.heads = 2                                                   # d != heads * 64
    assert not create(ctypes.byref(bad), lw) and b"64-wide heads" in emu_lib.dll.cv_last_error()
    bad = Lm1Config.from_buffer_copy(c)
    bad.dec_w = None
    assert not create(ctypes.byref(bad), lw) and b"16B aligned" in emu_lib.dll.cv_last_error()
    h = ctypes.c_void_p(create(ctypes.byref(c), lw))
    assert h
    try:
        x, logits = torch.zeros(d + 4), torch.zeros(n_out)
        with pytest.raises(CosyVoiceAmdError, match="cv_lm1_bind first"):
            emu_lib.cv_lm1_step(h, ctypes.c_void_p(x.data_ptr()), ctypes.c_int32(0), ctypes.c_void_p(logits.data_ptr()), None)
        rows, tabs = torch.zeros(8, 4 * d), torch.zeros(2 * 8 - 1, d)
        pr, pt = (ctypes.c_void_p * 1)(rows.data_ptr()), (ctypes.c_void_p * 1)(tabs.data_ptr())
        with pytest.raises(CosyVoiceAmdError, match="n_tab >= cap"):
            emu_lib.cv_lm1_bind(h, pr, pt, ctypes.c_int32(4), ctypes.c_int32(8), None)
        emu_lib.cv_lm1_bind(h, pr, pt, ctypes.c_int32(8), ctypes.c_int32(8), None)
        with pytest.raises(CosyVoiceAmdError, match="beyond the bound cache"):
            emu_lib.cv_lm1_step(h, ctypes.c_void_p(x.data_ptr()), ctypes.c_int32(8), ctypes.c_void_p(logits.data_ptr()), None)
        with pytest.raises(CosyVoiceAmdError, match="unaligned"):
            emu_lib.cv_lm1_step(h, ctypes.c_void_p(x.data_ptr() + 4), ctypes.c_int32(0), ctypes.c_void_p(logits.data_ptr()), None)
        emu_lib.cv_lm1_step(h, ctypes.c_void_p(x.data_ptr()), ctypes.c_int32(0), ctypes.c_void_p(logits.data_ptr()), None)      # zero weights: zero logits, one step counted
        assert float(logits.abs().max()) == 0.0 and emu_lib.raw("cv_lm1_stat", ctypes.c_int64)(h, b"steps") == 1
        assert emu_lib.raw("cv_lm1_stat", ctypes.c_int64)(h, b"no such counter") == -1
        with pytest.raises(CosyVoiceAmdError, match="unknown option"):
            emu_lib.cv_lm1_set_option(h, b"nope", ctypes.c_int32(1))
    finally:
        emu_lib.raw("cv_lm1_destroy", None)(h)
