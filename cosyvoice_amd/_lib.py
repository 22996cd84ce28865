"""ctypes binding of libcosyvoice_amd.so (the gfx950 HIP library).

The product path calls `get_lib()`: it loads ONLY cosyvoice_amd/libcosyvoice_amd.so and raises if the
library is missing or if it identifies itself as the CPU emulator build (tests/emu) — there is no CPU
fallback.  Tests that run kernels under the emulator construct `Lib(path, allow_emulated=True)` explicitly.
"""
import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "libcosyvoice_amd.so")

c_f32p = C.POINTER(C.c_float)


class GemmConvArgs(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("a_batch", C.c_int64), ("a_len", C.c_int64), ("lda", C.c_int32), ("a_off0", C.c_int32),
        ("tap_step", C.c_int32), ("taps", C.c_int32), ("K", C.c_int32),
        ("pro", C.c_int32), ("pro_p", C.c_float), ("pro_alpha", C.c_void_p),
        ("W", C.c_void_p), ("w_dtype", C.c_int32), ("Kp", C.c_int32), ("ldw", C.c_int64), ("w_batch", C.c_int64),
        ("bias", C.c_void_p),
        ("C", C.c_void_p), ("c_batch", C.c_int64), ("c_len", C.c_int64), ("ldc", C.c_int32), ("c_off", C.c_int64),
        ("M", C.c_int32), ("N", C.c_int32), ("batch", C.c_int32),
        ("act", C.c_int32), ("act_p", C.c_float),
        ("res", C.c_void_p), ("res_batch", C.c_int64),
        ("out_scale", C.c_float),
        ("row_scale", C.c_void_p), ("row_scale_batch", C.c_int64),
        ("accumulate", C.c_int32), ("a_bf16", C.c_int32), ("W3", C.c_void_p),
    ]


class AttnArgs(C.Structure):
    _fields_ = [
        ("q", C.c_void_p), ("q_batch", C.c_int64), ("q_row", C.c_int32), ("q_head", C.c_int32),
        ("k", C.c_void_p), ("k_batch", C.c_int64), ("k_row", C.c_int32), ("k_head", C.c_int32),
        ("v", C.c_void_p), ("v_batch", C.c_int64), ("v_row", C.c_int32), ("v_head", C.c_int32),
        ("o", C.c_void_p), ("o_batch", C.c_int64), ("o_row", C.c_int32), ("o_head", C.c_int32),
        ("B", C.c_int32), ("H", C.c_int32), ("kv_group", C.c_int32), ("Tq", C.c_int32), ("Tk", C.c_int32),
        ("scale", C.c_float), ("mask_mode", C.c_int32), ("chunk", C.c_int32),
        ("rel_bd", C.c_void_p), ("bd_batch", C.c_int64), ("bd_head", C.c_int64), ("bd_row", C.c_int32), ("bf16", C.c_int32),
        ("klen", C.c_void_p),
    ]


class Lm1LayerWeights(C.Structure):
    """cv_lm1_layer_weights (include/cosyvoice_amd.h)"""
    _fields_ = [(n, C.c_void_p) for n in ("ln1_g", "ln1_b", "w_qkv", "b_qkv", "w_out", "b_out", "ln2_g", "ln2_b", "w1", "b1", "w2", "b2")]


class Lm1LayerBf16(C.Structure):
    """cv_lm1_layer_bf16 (include/cosyvoice_amd.h)"""
    _fields_ = [(n, C.c_void_p) for n in ("w_qkv", "w_out", "w1", "w2")]


class Lm1Config(C.Structure):
    """cv_lm1_config (include/cosyvoice_amd.h)"""
    _fields_ = [("n_layers", C.c_int32), ("d", C.c_int32), ("heads", C.c_int32), ("ffn", C.c_int32), ("d_in", C.c_int32), ("n_out", C.c_int32),
                ("act", C.c_int32), ("xscale", C.c_float)] + [(n, C.c_void_p) for n in ("embed_w", "embed_b", "embed_g", "embed_beta", "after_g", "after_b", "dec_w", "dec_b")]


CV_F32, CV_BF16, CV_I32, CV_U8 = 0, 1, 2, 3
ACT = dict(none=0, silu=1, gelu_erf=2, elu=3, leaky=4, tanh=5, mish=6, abs=7, snake=8, logclamp=9, gelu_tanh=10, relu=11)
MASK = dict(none=0, causal=1, chunk=2)


class CosyVoiceAmdError(RuntimeError):
    pass


class Lib:
    """Thin handle on the shared library; every C entry point returns 0 or raises with cv_last_error()."""

    def __init__(self, path=DEFAULT_LIB, allow_emulated=False):
        if not os.path.exists(path):
            raise CosyVoiceAmdError(
                "HIP extension %s not found: build it with `python -m cosyvoice_amd.build` "
                "(or __graft_entry__.build()). There is no CPU fallback." % path)
        self.path = path
        self.dll = C.CDLL(path)
        self.dll.cv_last_error.restype = C.c_char_p
        self.dll.cv_version.restype = C.c_char_p
        self.emulated = bool(self.dll.cv_is_emulated())
        self.experiments = bool(self.dll.cv_has_experiments())     # built with CV_BUILD_EXPERIMENTS: the measured no-go variants exist and their options are accepted
        self.tensor_hook = None      # tests install a guard-page allocator here (tests/guard.py); identity in production
        if self.emulated and not allow_emulated:
            raise CosyVoiceAmdError("%s is the CPU-emulator test build; refusing to use it as the product path" % path)

    def hook(self, t):
        return t if self.tensor_hook is None or t is None else self.tensor_hook(t)

    def check(self, rc):
        if rc != 0:
            raise CosyVoiceAmdError(self.dll.cv_last_error().decode())

    def __getattr__(self, name):
        fn = getattr(self.dll, name)
        fn.restype = C.c_int

        def call(*args):
            self.check(fn(*args))
        return call

    def raw(self, name, restype=C.c_int):
        fn = getattr(self.dll, name)
        fn.restype = restype
        return fn

    @property
    def device(self):
        return "cpu" if self.emulated else "cuda"


_lock = threading.Lock()
_default = None


def get_lib():
    global _default
    with _lock:
        if _default is None:
            _default = Lib(DEFAULT_LIB, allow_emulated=False)
        return _default


def ptr(t):
    """data pointer of a torch tensor (or None)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr(lib):
    if lib.emulated:
        return None
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
