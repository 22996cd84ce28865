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
