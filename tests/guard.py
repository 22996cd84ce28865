"""Guard-page tensors for emulator runs (TEST INFRASTRUCTURE): a tensor whose storage ends flush against an
inaccessible page, so a kernel that reads or writes past the end of one of its operands crashes the test."""
import ctypes

import torch

_keep = []


def guard(lib, t):
    """Copy `t` into guard-page backed memory owned by the emulator library. No-op on the real GPU."""
    if t is None or not getattr(lib, "emulated", False) or not torch.is_tensor(t):
        return t
    t = t.contiguous()
    n = t.numel() * t.element_size()
    if n == 0:
        return t
    fn = lib.dll.emu_guard_alloc
    fn.restype = ctypes.c_void_p
    fn.argtypes = [ctypes.c_size_t]
    p = fn(n)
    buf = (ctypes.c_char * n).from_address(p)
    if t.dtype == torch.bfloat16:
        g = torch.frombuffer(buf, dtype=torch.int16).view(torch.bfloat16).view(t.shape)
    else:
        g = torch.frombuffer(buf, dtype=t.dtype).view(t.shape)
    g.copy_(t)
    _keep.append(buf)
    return g
