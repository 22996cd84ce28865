"""torch-tensor front-ends of the operator-level C ABI (cv_gemm_conv / cv_norm_rows / cv_attention).

These are used by the parity tests and by the weight repacker; the stage-level entry points
(cv_llm_*, cv_flow_*, cv_hift_*) launch the same kernels from C++ without coming back to Python.
"""
import ctypes as C
import torch

from ._lib import ACT, MASK, CV_BF16, CV_F32, AttnArgs, GemmConvArgs, ptr, stream_ptr


def round_up(x, m):
    return (x + m - 1) // m * m


def pack_weight(w, dtype=torch.bfloat16):
    """[N, taps, K] (or [N, K]) float tensor -> contiguous [N, taps*Kp] with Kp = round_up(K, 32), zero padded."""
    if w.dim() == 2:
        w = w.unsqueeze(1)
    n, taps, k = w.shape
    kp = round_up(k, 32)
    out = torch.zeros(n, taps, kp, dtype=dtype, device=w.device)
    out[:, :, :k] = w.to(dtype)
    return out.reshape(n, taps * kp).contiguous(), kp


def gemm_conv(lib, A, Wp, Kp, *, M, N, K, taps=1, lda=None, a_off0=0, tap_step=0, a_len=None, a_batch=0,
              bias=None, out=None, ldc=None, c_off=0, c_len=None, c_batch=0, batch=1,
              pro="none", pro_p=0.0, pro_alpha=None, act="none", act_p=0.0, res=None, res_batch=0,
              out_scale=1.0, row_scale=None, row_scale_batch=0, accumulate=False, ldw=0, w_batch=0, a_bf16=False, w3=None):
    lda = K if lda is None else lda
    ldc = N if ldc is None else ldc
    if a_len is None:
        a_len = A.numel() if batch == 1 else a_batch
    if out is None:
        out = lib.hook(torch.zeros(batch, M, N, dtype=torch.float32, device=A.device))
    if c_len is None:
        c_len = out.numel() if batch == 1 else c_batch
    g = GemmConvArgs()
    g.A = A.data_ptr(); g.a_batch = a_batch; g.a_len = a_len; g.lda = lda; g.a_off0 = a_off0
    g.tap_step = tap_step; g.taps = taps; g.K = K
    g.pro = ACT[pro]; g.pro_p = pro_p; g.pro_alpha = pro_alpha.data_ptr() if pro_alpha is not None else None
    g.W = Wp.data_ptr(); g.w_dtype = CV_BF16 if Wp.dtype == torch.bfloat16 else CV_F32; g.Kp = Kp; g.ldw = ldw; g.w_batch = w_batch
    g.bias = bias.data_ptr() if bias is not None else None
    g.C = out.data_ptr(); g.c_batch = c_batch; g.c_len = c_len; g.ldc = ldc; g.c_off = c_off
    g.M = M; g.N = N; g.batch = batch
    g.act = ACT[act]; g.act_p = act_p
    g.res = res.data_ptr() if res is not None else None; g.res_batch = res_batch
    g.out_scale = out_scale
    g.row_scale = row_scale.data_ptr() if row_scale is not None else None; g.row_scale_batch = row_scale_batch
    g.accumulate = int(accumulate)
    g.a_bf16 = int(a_bf16)
    g.W3 = w3.data_ptr() if w3 is not None else None          # fp32 weights pre-split into three bf16 planes (weights.split3_planes): two-sided split GEMM
    lib.cv_gemm_conv(C.byref(g), stream_ptr(lib))
    return out


def norm_rows(lib, x, gamma=None, beta=None, eps=1e-5, rms=False, act="none", scale=1.0, row_scale=None,
              col_add=None, rows_per_batch=0):
    x = x.contiguous()
    rows, c = x.numel() // x.shape[-1], x.shape[-1]
    y = lib.hook(torch.empty_like(x))
    lib.cv_norm_rows(ptr(x), ptr(y), C.c_int64(rows), C.c_int32(c), ptr(gamma), ptr(beta), C.c_float(eps),
                     C.c_int32(int(rms)), C.c_int32(ACT[act]), C.c_float(scale), ptr(row_scale), ptr(col_add),
                     C.c_int64(rows_per_batch), stream_ptr(lib))
    return y


def attention(lib, q, k, v, *, scale, mask="none", chunk=0, kv_group=1, rel_bd=None, bf16=False, klen=None):
    """q [B,Tq,H,64], k/v [B,Tk,Hkv,64] (any strides that are multiples of 4 floats) -> o [B,Tq,H,64].  klen: optional int32 [B] on the device, batch
    row b attends keys < klen[b] only."""
    B, Tq, H, D = q.shape
    Tk = k.shape[1]
    assert D == 64
    o = lib.hook(torch.empty(B, Tq, H, 64, dtype=torch.float32, device=q.device))
    a = AttnArgs()
    for name, t in (("q", q), ("k", k), ("v", v), ("o", o)):
        setattr(a, name, t.data_ptr())
        setattr(a, name + "_batch", t.stride(0)); setattr(a, name + "_row", t.stride(1)); setattr(a, name + "_head", t.stride(2))
        assert t.stride(3) == 1
    a.B = B; a.H = H; a.kv_group = kv_group; a.Tq = Tq; a.Tk = Tk
    a.scale = scale; a.mask_mode = MASK[mask]; a.chunk = chunk
    if rel_bd is not None:
        assert rel_bd.shape == (B, H, Tq, 2 * Tq - 1) and rel_bd.is_contiguous()
        a.rel_bd = rel_bd.data_ptr(); a.bd_batch = rel_bd.stride(0); a.bd_head = rel_bd.stride(1); a.bd_row = rel_bd.stride(2)
    a.bf16 = int(bf16)
    a.klen = klen.data_ptr() if klen is not None else None
    lib.cv_attention(C.byref(a), stream_ptr(lib))
    return o
