"""CosyVoice-300M (first-generation CosyVoice) on the hand-written gfx950 kernels - SURVEY.md section 8 row f4, second half.

`cosyvoice1.py` is the torch-eager plumbing of BASELINE.json configs[0] (CPU, no HIP library).  This module is the same model with every tensor
operation of the three stages behind the C ABI of libcosyvoice_amd.so; the host code below only sequences launches and keeps the request state:

  TransformerLM.inference      llm/llm.py:162-223        ConformerEncoder text encoder + 14-block TransformerEncoder stepped with a KV cache:
                                                          cv_gemm_conv (fp32-exact GEMMs; q + pos_bias_u | q + pos_bias_v | k | v from ONE GEMM over stacked
                                                          weights, written straight into the layer's cache rows), cv_norm_rows, cv_attention with the
                                                          relative-position term (matrix_bd of attention.py:318 as a batched GEMM over heads against a
                                                          per-layer table of linear_pos(pe) that is computed once, not once per step like the reference)
  MaskedDiffWithXvec.inference flow/flow.py:102-146      ConformerEncoder, InterpolateRegulator (cv_interp_rows, conv + cv_group_norm), ConditionalCFM with
                                                          the flow cache, the U-Net ConditionalDecoder flow/decoder.py:88-291: Block1D = conv GEMM +
                                                          cv_group_norm (Mish and the time projection fused), stride-2 Downsample1D and polyphase
                                                          ConvTranspose Upsample1D as cv_gemm_conv index forms, BasicTransformerBlock on cv_attention;
                                                          classifier-free guidance + Euler update in cv_cfg_euler
  HiFTGenerator.inference      hifigan/generator.py:378-569 at 22.05 kHz: cv_hift_f0, the type-1 SineGen source (cv_sinegen1_source), cv_hift_decode

Activations are fp32, channel-last [time][channel] (batch rows stacked); weights are the reference's state-dict tensors repacked once at load time
(fp32, K padded to 32).  Random draws: the sampler and the CFM noise use the host torch RNG in the reference's order (so the goldens of the real
classes apply, tests/test_zzz_cosyvoice1_hip*.py); the SineGen noise does too when `rng="host"` (parity) and comes from the kernel's counter RNG otherwise.
There is no CPU fallback: without the HIP library `get_lib()` raises.
"""
import ctypes as C
import math
import threading

import numpy as np
import torch

from . import cosyvoice1 as C1
from ._lib import ACT, CV_F32, MASK, AttnArgs, Lm1Config, Lm1LayerBf16, Lm1LayerWeights, get_lib, stream_ptr
from .hift import HiFTGenerator as _KernelHiFT
from .ops import gemm_conv, norm_rows, pack_weight
from .weights import split3_planes

F32 = torch.float32


class _Mat:
    """A GEMM weight operand: fp32 [N][taps * Kp] on the device (+ bias)."""
    __slots__ = ("w", "kp", "n", "k", "taps", "b", "w3")

    def __init__(self, w, kp, n, k, taps, b, w3=None):
        self.w, self.kp, self.n, self.k, self.taps, self.b, self.w3 = w, kp, n, k, taps, b, w3


class LaunchTape:
    """A recorded sequence of library launches with the buffers they touch.  The host side of this model is python: issuing a launch costs ~20 us of
    interpreter time (argument structs, output allocation, views) against 3-10 us of kernel time, so a sequence that repeats - the estimator of every Euler
    step after the first, the LM decode step - is recorded once and then replayed as bare ctypes calls on the SAME buffers (~2 us per launch); what changes
    between replays is patched into the recorded argument structs (`structs`).  The role a hipGraph plays for the CosyVoice2 path, kept on the host side so
    that the CPU emulator exercises it too."""

    def __init__(self, lib):
        self.lib, self.calls, self.keep, self.stream = lib, [], [], None

    def replay(self, stream=None):
        """stream: a c_void_p to launch on instead of the recorded stream (every entry point takes the stream as its LAST argument) - what a capture needs."""
        for fn, args in self.calls:
            if stream is not None:
                args = args[:-1] + (stream,)
            if fn(*args) != 0:
                self.lib.check(1)

    def capture(self):
        """The tape as a hipGraph (torch.cuda.CUDAGraph around one replay on the capture stream; thread-local capture mode: the LM thread of a streaming
        request keeps making synchronous calls).  Opt-in (Kernels.use_graphs) until its first MI355X run: returns None when the capture fails."""
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                self.replay(stream=C.c_void_p(torch.cuda.current_stream().cuda_stream))
            return g
        except Exception as e:                                  # noqa: BLE001 - reported, and the eager replay keeps serving
            self.capture_error = repr(e)
            return None

    @property
    def structs(self):
        """the argument struct of every call that takes one (cv_gemm_conv, cv_attention), None for the others - in call order"""
        return [getattr(a[0], "_obj", None) if a else None for _, a in self.calls]


class _Recorder:
    """Stands in for the Lib handle while a LaunchTape is recorded: every cv_* call is executed AND appended, every tensor that passes through hook() - which is
    every buffer the launch helpers allocate - is kept alive with the tape."""

    def __init__(self, lib, tape):
        self._lib, self._tape = lib, tape

    def hook(self, t):
        r = self._lib.hook(t)
        if r is not None:
            self._tape.keep.append(r)
        return r

    def __getattr__(self, name):
        if not name.startswith("cv_"):
            return getattr(self._lib, name)
        raw = self._lib.raw(name)

        def call(*args):
            self._lib.check(raw(*args))
            self._tape.calls.append((raw, args))
        return call


class Kernels:
    """Launch helpers over the operator-level C ABI for channel-last fp32 activations."""

    def __init__(self, lib=None, split3=False):
        """split3: every weight matrix also as three bf16 planes (weights.split3_planes), so that the GEMMs run both operands split on the bf16 matrix pipe
        (csrc/gemm_conv.h WX3: six exact plane products per k, fp32 accuracy) instead of the fp32 MFMA chain.  Same results to fp32 rounding; which one is
        faster at these shapes is a measurement for the first MI355X run of this path (HiFT: neutral), so it is an option."""
        self.lib = lib or get_lib()
        self.dev = torch.device(self.lib.device)
        self.split3 = bool(split3)
        self._gn_ws = self.lib.hook(torch.zeros(64 * 64 * 2 * 8, dtype=torch.float64, device=self.dev))      # cv_group_norm partial sums: B * G * 64 doubles
        self.use_tapes = True                                   # False: every launch sequenced from scratch (A/B and test knob)
        self.use_graphs = False                                 # True (MI355X only): a fixed tape - the estimator of one solve - is replayed as a hipGraph (LaunchTape.capture)
        self.graph_replays = 0

    def record(self):
        """`with K.record() as tape:` - the launches issued inside are executed and recorded (LaunchTape)."""
        import contextlib

        @contextlib.contextmanager
        def cm():
            real = self.lib
            tape = LaunchTape(real)
            sp = stream_ptr(real)
            tape.stream = None if sp is None else sp.value
            self.lib = _Recorder(real, tape)
            try:
                yield tape
            finally:
                self.lib = real
        return cm()

    def same_stream(self, tape):
        sp = stream_ptr(self.lib)
        return tape.stream == (None if sp is None else sp.value)

    # ---- memory ----
    def new(self, *shape):
        if self.lib.emulated:                                   # test builds: an output element that a kernel fails to write (or an operand read before it is written) shows as NaN
            return self.lib.hook(torch.full(shape, float("nan"), dtype=F32, device=self.dev))
        return self.lib.hook(torch.empty(*shape, dtype=F32, device=self.dev))

    def zeros(self, *shape):
        return self.lib.hook(torch.zeros(*shape, dtype=F32, device=self.dev))

    def put(self, t, dtype=F32):
        return self.lib.hook(t.detach().to(self.dev, dtype).contiguous())

    def mat(self, w, bias=None):
        """w: [N, K] or [N, taps, K] (tap-major, channel-minor: the order a channel-last activation window has in memory)."""
        w = w.detach().float()
        n, taps, k = (w.shape[0], 1, w.shape[1]) if w.dim() == 2 else w.shape
        wp, kp = pack_weight(w.to(self.dev), F32)
        w3 = self.lib.hook(split3_planes(wp)) if self.split3 else None
        return _Mat(self.lib.hook(wp), kp, n, k, taps, None if bias is None else self.put(bias), w3)

    def conv_mat(self, w, bias=None):
        """torch Conv1d weight [C_out, C_in, k] -> taps form."""
        return self.mat(w.permute(0, 2, 1), bias)

    # ---- launches ----
    def linear(self, x, m, M, *, lda=None, act="none", res=None, out=None, ldc=None, out_scale=1.0):
        """out[M, N] = act(x[M, K] W^T + b) * out_scale (+ res).  x / out / res may be row-pitched views (lda / ldc; res is indexed like out)."""
        lda = m.k if lda is None else lda
        ldc = m.n if ldc is None else ldc
        if out is None:
            out = self.new(M, m.n)
        gemm_conv(self.lib, x, m.w, m.kp, M=M, N=m.n, K=m.k, lda=lda, a_len=(M - 1) * lda + m.k, bias=m.b, out=out, ldc=ldc, c_len=(M - 1) * ldc + m.n,
                  act=act, res=res, out_scale=out_scale, w3=m.w3)
        return out

    def conv(self, x, m, B, T, *, pad, dil=1, act="none", res=None):
        """Conv1d(stride 1) over [B][T][C_in] -> [B][T_out][N]; zero padding is the range check of the flat index."""
        t_out = T + 2 * pad - dil * (m.taps - 1)
        out = self.new(B, t_out, m.n)
        gemm_conv(self.lib, x, m.w, m.kp, M=t_out, N=m.n, K=m.k, taps=m.taps, lda=m.k, a_off0=-pad * m.k, tap_step=dil * m.k, a_len=T * m.k, a_batch=T * m.k,
                  bias=m.b, out=out, c_len=t_out * m.n, c_batch=t_out * m.n, batch=B, act=act, res=res, res_batch=t_out * m.n, w3=m.w3)
        return out

    def conv_stride(self, x, m, B, T, c_in, *, k, stride, pad):
        """Strided Conv1d: the im2col window of a channel-last sequence is contiguous, so it is a Linear over rows of pitch stride * C_in (m.k = k * C_in)."""
        t_out = (T + 2 * pad - k) // stride + 1
        out = self.new(B, t_out, m.n)
        gemm_conv(self.lib, x, m.w, m.kp, M=t_out, N=m.n, K=m.k, lda=stride * c_in, a_off0=-pad * c_in, a_len=T * c_in, a_batch=T * c_in, bias=m.b, out=out,
                  c_len=t_out * m.n, c_batch=t_out * m.n, batch=B, w3=m.w3)
        return out, t_out

    def conv_transpose(self, x, m, B, T, c_in, c_out, *, k, stride, pad):
        """ConvTranspose1d in polyphase form (weights packed by `tconv_mat`): GEMM row j produces output rows j * stride - pad ... + stride - 1."""
        t_out = (T - 1) * stride - 2 * pad + k
        out = self.new(B, t_out, c_out)
        gemm_conv(self.lib, x, m.w, m.kp, M=T + m.taps - 1, N=stride * c_out, K=c_in, taps=m.taps, lda=c_in, a_off0=0, tap_step=-c_in, a_len=T * c_in, a_batch=T * c_in,
                  bias=m.b, out=out, ldc=stride * c_out, c_off=-pad * c_out, c_len=t_out * c_out, c_batch=t_out * c_out, batch=B, w3=m.w3)
        return out, t_out

    def tconv_mat(self, w, bias, stride):
        """torch ConvTranspose1d weight [C_in, C_out, k] -> [stride * C_out][q taps][C_in] (phase r, tap j holds kernel element r + stride * j)."""
        w = w.detach().float()
        cin, cout, k = w.shape
        q = (k + stride - 1) // stride
        wp = torch.zeros(stride, cout, q, cin)
        for r in range(stride):
            for j in range(q):
                if r + stride * j < k:
                    wp[r, :, j, :] = w[:, :, r + stride * j].t()
        return self.mat(wp.reshape(stride * cout, q, cin), bias.detach().float().repeat(stride))

    def layer_norm(self, x, g, b, eps, act="none", scale=1.0):
        return norm_rows(self.lib, x, g, b, eps=eps, act=act, scale=scale)

    def group_norm(self, x, B, T, Cc, G, g, b, act="none", col_add=None, eps=1e-5):
        assert B * G * 64 * 2 <= self._gn_ws.numel()
        y = self.new(B, T, Cc)
        self.lib.cv_group_norm(_p(x), _p(y), C.c_int32(B), C.c_int32(T), C.c_int32(Cc), C.c_int32(G), _p(g), _p(b), C.c_float(eps), C.c_int32(ACT[act]),
                               _p(col_add), C.c_int64(0), _p(self._gn_ws), stream_ptr(self.lib))
        return y

    def attention(self, q, k, v, *, Tq, Tk, H, B=1, scale, causal=False, rel_bd=None, bd_row=0):
        """q/k/v: 4-D views [B, T, H, 64] (any row / head / batch pitch that is a multiple of 4 floats) -> o [B, Tq, H, 64] contiguous."""
        o = self.new(B, Tq, H, 64)
        a = AttnArgs()
        for name, t in (("q", q), ("k", k), ("v", v), ("o", o)):
            assert t.stride(3) == 1 and t.shape[3] == 64
            setattr(a, name, t.data_ptr())
            setattr(a, name + "_batch", t.stride(0)); setattr(a, name + "_row", t.stride(1)); setattr(a, name + "_head", t.stride(2))
        a.B, a.H, a.kv_group, a.Tq, a.Tk = B, H, 1, Tq, Tk
        a.scale, a.mask_mode, a.chunk = scale, MASK["causal" if causal else "none"], 0
        if rel_bd is not None:
            a.rel_bd, a.bd_batch, a.bd_head, a.bd_row = rel_bd.data_ptr(), 0, Tq * bd_row, bd_row
        a.bf16, a.klen = 0, None
        self.lib.cv_attention(C.byref(a), stream_ptr(self.lib))
        return o

    def gather(self, table, ids):
        """rows of an fp32 [V, D] device table for int ids (host list / tensor) -> [n, D]."""
        ids = self.put(torch.as_tensor(ids).reshape(-1), torch.int32)
        out = self.new(ids.numel(), table.shape[1])
        self.lib.cv_gather_rows(_p(table), C.c_int32(CV_F32), C.c_int64(table.shape[0]), C.c_int32(table.shape[1]), _p(ids), C.c_int32(ids.numel()), _p(out),
                                C.c_float(1.0), stream_ptr(self.lib))
        return out


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


# ------------------------------------------------------------------------------------------------------------------------------------
# espnet encoders with relative positions (transformer/encoder.py, encoder_layer.py, attention.py:249-330, embedding.py:201-302)
# ------------------------------------------------------------------------------------------------------------------------------------
class _KVState:
    """Per-layer rows [cap, 4 d] = (q + u | q + v | k | v) of everything forwarded so far (forward_chunk's att_cache, kept in place)."""

    _serials = iter(range(1, 1 << 62))

    def __init__(self, kern, n_layers, d, cap):
        self.kern, self.d, self.len, self.cap = kern, d, 0, cap
        self.rows = [kern.zeros(cap, 4 * d) for _ in range(n_layers)]
        self.ident = (next(_KVState._serials), 0)               # (which state, which allocation of its rows): what a bound decode step (cv_lm1_bind) is valid for

    def reserve(self, n):
        if n <= self.cap:
            return
        cap = max(n, 2 * self.cap)
        for i, old in enumerate(self.rows):
            new = self.kern.zeros(cap, 4 * self.d)
            new[: self.len].copy_(old[: self.len])
            self.rows[i] = new
        self.cap = cap
        self.ident = (self.ident[0], self.ident[1] + 1)


class _FusedStep:
    """A cv_lm1 handle with the buffers it points into (weights are the encoder's own tensors; `keep` holds the argument structs and the decoder matrix)."""

    def __init__(self, lib, h, logits, keep):
        self.lib, self.h, self.logits, self.keep, self.bound = lib, h, logits, keep, None

    def stat(self, name):
        return int(self.lib.raw("cv_lm1_stat", C.c_int64)(self.h, name.encode()))

    def __del__(self):
        try:
            self.lib.raw("cv_lm1_destroy", None)(self.h)
        except Exception:                                       # noqa: BLE001 - interpreter shutdown
            pass


class EspnetEncoder(C1.EspnetEncoder):
    """cosyvoice1.EspnetEncoder with every operation on the kernels.  Head size must be 64 (cv_attention): 1024 / 16 and 512 / 8 in CosyVoice-300M."""

    def __init__(self, sd, prefix, heads, kind, causal=False, kern=None):
        super().__init__(sd, prefix, heads, kind, causal)
        self.k = kern or Kernels()
        assert self.d == heads * 64, "cv_attention serves 64-wide heads"
        p, K = self.p, self.k
        self.embed = K.mat(p("embed.out.0.weight"), p("embed.out.0.bias"))
        self.embed_ln = (K.put(p("embed.out.1.weight")), K.put(p("embed.out.1.bias")))
        self.after = (K.put(p("after_norm.weight")), K.put(p("after_norm.bias")))
        n_att, n_ff = ("norm_mha.", "norm_ff.") if kind == "conformer" else ("norm1.", "norm2.")
        self.layers = []
        for i in range(self.n_layers):
            L = p.sub("encoders.%d." % i)
            a = L.sub("self_attn.")
            u, v = a("pos_bias_u").reshape(-1), a("pos_bias_v").reshape(-1)
            wq, bq = a("linear_q.weight"), a("linear_q.bias")
            w4 = torch.cat([wq, wq, a("linear_k.weight"), a("linear_v.weight")], 0)
            b4 = torch.cat([bq + u, bq + v, a("linear_k.bias"), a("linear_v.bias")], 0)
            self.layers.append(dict(
                ln1=(K.put(L(n_att + "weight")), K.put(L(n_att + "bias"))), ln2=(K.put(L(n_ff + "weight")), K.put(L(n_ff + "bias"))),
                qkv=K.mat(w4, b4), pos=K.mat(a("linear_pos.weight")), out=K.mat(a("linear_out.weight"), a("linear_out.bias")),
                w1=K.mat(L("feed_forward.w_1.weight"), L("feed_forward.w_1.bias")), w2=K.mat(L("feed_forward.w_2.weight"), L("feed_forward.w_2.bias"))))
        self._pos_n, self._pos_tab = 0, None

    # linear_pos(pe) for every relative position of a window of `n` keys, per layer: row m of the table is relative position n_tab - 1 - m
    def _pos_tables(self, n_keys):
        if n_keys > self._pos_n:
            n = max(64, self._pos_n)
            while n < n_keys:
                n *= 2
            pe = self.k.put(C1.EspnetEncoder._pos(self, n, torch.zeros(1)))
            self._pos_tab = [self.k.linear(pe, L["pos"], 2 * n - 1) for L in self.layers]
            self._pos_n = n
        return self._pos_n, self._pos_tab

    def _embed(self, xs, M):
        y = self.k.linear(xs, self.embed, M)
        return self.k.layer_norm(y, *self.embed_ln, 1e-5, act="relu" if self.kind == "transformer" else "none", scale=math.sqrt(self.d))

    def _layer(self, i, x, t1, rows, t0, causal, bd_cols=0):
        """x [t1, d]; rows: the layer's [cap, 4 d] buffer holding t0 earlier positions.  bd_cols: row pitch of matrix_bd (0: what this call needs; forward_chunk
        passes what the cache's capacity will ever need, so that the recorded decode step keeps its geometry while the context grows)."""
        K, L, d, H = self.k, self.layers[i], self.d, self.heads
        n_keys = t0 + t1
        K.linear(K.layer_norm(x, *L["ln1"], 1e-12), L["qkv"], t1, out=rows[t0:], ldc=4 * d)
        n_tab, tabs = self._pos_tables(n_keys)
        P = t1 - 1 + n_keys                                    # columns rel_shift can reach: bd[i][c] is relative position n_keys - 1 - c
        Pp = (max(P, bd_cols) + 3) // 4 * 4
        bd = K.new(H, t1, Pp)
        pp = tabs[i][n_tab - n_keys:]
        gemm_conv(K.lib, rows[t0:, d:], pp, 64, M=t1, N=P, K=64, lda=4 * d, a_batch=64, a_len=(t1 - 1) * 4 * d + 64, ldw=d, w_batch=64,
                  out=bd, ldc=Pp, c_batch=t1 * Pp, c_len=t1 * Pp, batch=H)
        heads = lambda t: t.unflatten(1, (H, 64)).unsqueeze(0)
        o = K.attention(heads(rows[t0:n_keys, 0:d]), heads(rows[:n_keys, 2 * d:3 * d]), heads(rows[:n_keys, 3 * d:4 * d]), Tq=t1, Tk=n_keys, H=H, scale=0.125,
                        causal=causal and t1 > 1, rel_bd=bd, bd_row=Pp)
        x = K.linear(o, L["out"], t1, res=x)
        y = K.linear(K.layer_norm(x, *L["ln2"], 1e-12), L["w1"], t1, act="silu" if self.kind == "conformer" else "relu")
        return K.linear(y, L["w2"], t1, res=x)

    def forward(self, xs):
        """xs [T, d_in] on the device -> [T, d]."""
        T = xs.shape[0]
        x = self._embed(xs, T)
        for i in range(self.n_layers):
            x = self._layer(i, x, T, self.k.new(T, 4 * self.d), 0, self.causal)
        return self.k.layer_norm(x, *self.after, 1e-5)

    def forward_chunk(self, xs, state):
        """BaseEncoder.forward_chunk with the whole history kept: xs [t1, d_in] (a view is fine), state = _KVState or None -> (ys [t1, d], state).
        One-row calls (the decode step) after the first are replays of a recorded step with the position-dependent arguments patched (LaunchTape); the ys of such
        a call is the recorded step's output buffer - valid until the next call on this state."""
        t1 = xs.shape[0]
        if state is None:
            state = _KVState(self.k, self.n_layers, self.d, max(256, 2 * t1))
        state.reserve(state.len + t1)
        K, t0 = self.k, state.len
        self._pos_tables(t0 + t1)                               # (a table that has to grow does so here, outside any recording)
        step = t1 == 1 and t0 >= 1 and K.use_tapes
        plan = getattr(state, "plan", None)
        if step and plan is not None and plan["cap"] == state.cap and plan["pos_n"] == self._pos_n and K.same_stream(plan["tape"]):
            self._patch_step(plan, xs, state, t0)
            plan["tape"].replay()
            state.len += 1
            return plan["y"], state
        state.plan = None
        if step:
            with K.record() as tape:
                y = self._forward_rows(xs, t1, state, t0)
            state.plan = dict(tape=tape, y=y, cap=state.cap, pos_n=self._pos_n, structs=tape.structs)
            assert len(tape.calls) == 3 + 8 * self.n_layers, len(tape.calls)
        else:
            y = self._forward_rows(xs, t1, state, t0)
        state.len += t1
        return y, state

    # ---- the decode step as ONE library call (csrc/lm1.hip: 3 + 5 launches per layer inside a hipGraph instead of 3 + 8 per layer replayed from here) -----------
    def make_step(self, decoder, bf16=False):
        """cv_lm1 handle over this encoder's weights + the LM's decoder matrix (`decoder`: _Mat [n_out, d]).  None when the shapes are outside what the fused
        step serves (it is an acceleration of forward_chunk + decoder, not another model): d > 1024, feed-forward > 4096.  bf16: the step reads bf16 copies of its
        matrices (cv_lm1_use_bf16: the model's fp16 mode; the copies are made once per encoder and shared by every handle)."""
        K, L0 = self.k, self.layers[0]
        if self.kind != "transformer" or self.d > 1024 or L0["w1"].n > 4096 or self.embed.k > 4096 or self.embed.k % 4 or K.split3:
            return None
        # cv_lm1_create's own conditions (csrc/lm1.hip), mirrored so that a model outside them falls back to the launch-per-operator step instead of raising:
        # 64-wide heads, feed-forward % 4, at most 32 layers, every bias present
        tensors = [t for L in self.layers for t in (L["ln1"][0], L["ln1"][1], L["qkv"].w, L["qkv"].b, L["out"].w, L["out"].b, L["ln2"][0], L["ln2"][1], L["w1"].w, L["w1"].b, L["w2"].w, L["w2"].b)]
        tensors += [self.embed.w, self.embed.b, self.embed_ln[0], self.embed_ln[1], self.after[0], self.after[1], decoder.w, decoder.b]
        if self.d != self.heads * 64 or L0["w1"].n % 4 or self.n_layers > 32 or any(t is None or t.data_ptr() % 16 for t in tensors):
            return None
        lw = (Lm1LayerWeights * self.n_layers)()
        for i, L in enumerate(self.layers):
            for name, t in (("ln1_g", L["ln1"][0]), ("ln1_b", L["ln1"][1]), ("w_qkv", L["qkv"].w), ("b_qkv", L["qkv"].b), ("w_out", L["out"].w), ("b_out", L["out"].b),
                            ("ln2_g", L["ln2"][0]), ("ln2_b", L["ln2"][1]), ("w1", L["w1"].w), ("b1", L["w1"].b), ("w2", L["w2"].w), ("b2", L["w2"].b)):
                setattr(lw[i], name, t.data_ptr())
        c = Lm1Config()
        c.n_layers, c.d, c.heads, c.ffn, c.d_in, c.n_out = self.n_layers, self.d, self.heads, L0["w1"].n, self.embed.k, decoder.n
        c.act, c.xscale = ACT["relu"], math.sqrt(self.d)
        for name, t in (("embed_w", self.embed.w), ("embed_b", self.embed.b), ("embed_g", self.embed_ln[0]), ("embed_beta", self.embed_ln[1]),
                        ("after_g", self.after[0]), ("after_b", self.after[1]), ("dec_w", decoder.w), ("dec_b", decoder.b)):
            setattr(c, name, t.data_ptr())
        real = getattr(K.lib, "_lib", K.lib)
        h = real.raw("cv_lm1_create", C.c_void_p)(C.byref(c), lw)
        if not h:                                               # refused by the library for a reason the gate above does not know: the operator-per-launch step serves the model
            return None
        step = _FusedStep(real, C.c_void_p(h), K.new(decoder.n), (c, lw, decoder))
        if bf16:
            if getattr(self, "_w16", None) is None:
                half = lambda m: K.put(m.w, torch.bfloat16)
                self._w16 = ([tuple(half(L[n]) for n in ("qkv", "out", "w1", "w2")) for L in self.layers], half(self.embed), half(decoder))
            l16 = (Lm1LayerBf16 * self.n_layers)()
            for i, ws in enumerate(self._w16[0]):
                l16[i].w_qkv, l16[i].w_out, l16[i].w1, l16[i].w2 = (t.data_ptr() for t in ws)
            real.cv_lm1_use_bf16(step.h, l16, C.c_void_p(self._w16[1].data_ptr()), C.c_void_p(self._w16[2].data_ptr()))
            step.keep = step.keep + (l16, self._w16)
        return step

    def fused_step(self, step, xs, state):
        """One decode row through `step` (make_step): cache row state.len of every layer is written, logits land in step.logits (valid until the next call)."""
        K, t0 = self.k, state.len
        state.reserve(t0 + 1)
        self._pos_tables(state.cap)                             # the tables cover every position the cache can hold: n_tab >= cap
        key = (state.ident, self._pos_n)
        if step.bound != key:
            n = self.n_layers
            rows = (C.c_void_p * n)(*[r.data_ptr() for r in state.rows])
            tabs = (C.c_void_p * n)(*[t.data_ptr() for t in self._pos_tab])
            step.lib.cv_lm1_bind(step.h, rows, tabs, C.c_int32(self._pos_n), C.c_int32(state.cap), stream_ptr(step.lib))
            step.bound = key
        step.lib.cv_lm1_step(step.h, C.c_void_p(xs.data_ptr()), C.c_int32(t0), C.c_void_p(step.logits.data_ptr()), stream_ptr(step.lib))
        state.len += 1
        state.plan = None
        return step.logits

    def loop_begin(self, step, x_row, state, sp, emb_table, uniforms=None):
        """Open the device-resident decode loop (cv_lm1_decode_begin): `x_row` [1, d_in] is the next input row, written at position state.len; `sp` a SamplingC with the
        request's bounds; the cache is grown ONCE for every position the loop can write (no rebinding while it runs)."""
        t0 = state.len
        state.reserve(t0 + int(sp.max_len) + 1)
        self._pos_tables(state.cap)
        key = (state.ident, self._pos_n)
        if step.bound != key:
            n = self.n_layers
            rows = (C.c_void_p * n)(*[r.data_ptr() for r in state.rows])
            tabs = (C.c_void_p * n)(*[t.data_ptr() for t in self._pos_tab])
            step.lib.cv_lm1_bind(step.h, rows, tabs, C.c_int32(self._pos_n), C.c_int32(state.cap), stream_ptr(step.lib))
            step.bound = key
        u = None if uniforms is None else uniforms.detach().to(torch.float32).cpu().contiguous()
        step.lib.cv_lm1_decode_begin(step.h, C.c_void_p(x_row.data_ptr()), C.c_int32(t0), C.byref(sp), C.c_void_p(emb_table.data_ptr()), C.c_int32(int(sp.max_len)),
                                     C.c_void_p(u.data_ptr()) if u is not None else None, C.c_int32(0 if u is None else u.numel()), stream_ptr(step.lib))
        state.plan = None

    def loop_steps(self, step, state, n):
        """n steps of the open loop; returns (tokens emitted by these steps, finished)."""
        buf, n_out, fin = (C.c_int32 * n)(), C.c_int32(0), C.c_int32(0)
        step.lib.cv_lm1_decode(step.h, C.c_int32(n), buf, C.byref(n_out), C.byref(fin), stream_ptr(step.lib))
        state.len += n_out.value                                 # (rows written past the last emitted token belong to nobody)
        return [int(buf[k]) for k in range(n_out.value)], bool(fin.value)

    def _forward_rows(self, xs, t1, state, t0):
        x = self._embed(xs, t1)
        for i in range(self.n_layers):
            x = self._layer(i, x, t1, state.rows[i], t0, True, bd_cols=t1 - 1 + state.cap)
        return self.k.layer_norm(x, *self.after, 1e-5)

    def _patch_step(self, plan, xs, state, t0):
        """The recorded one-row step at position t0: launches in order [embed GEMM, embed LN] + per layer [LN, qkv4 GEMM, matrix_bd GEMM, attention, out GEMM, LN,
        w1 GEMM, w2 GEMM] + [after_norm].  What depends on the position: where the new row (q + u | q + v | k | v) goes and is read from, the first table row of
        the relative positions, and the key count."""
        st, d, n_keys = plan["structs"], self.d, t0 + 1
        st[0].A = xs.data_ptr()
        row = 4 * (t0 * 4 * d)                                  # byte offset of row t0 in a layer's [cap][4 d] buffer
        tab0 = 4 * (self._pos_n - n_keys) * d
        for i in range(self.n_layers):
            base, rows = 2 + 8 * i, state.rows[i].data_ptr()
            qkv, bd, at = st[base + 1], st[base + 2], st[base + 3]
            qkv.C = rows + row
            bd.A, bd.W, bd.N = rows + row + 4 * d, self._pos_tab[i].data_ptr() + tab0, n_keys
            at.q, at.Tk = rows + row, n_keys


class TransformerLM(C1.TransformerLM):
    """cosyvoice.llm.llm.TransformerLM.inference (llm/llm.py:162-223) on the kernels; sampling decisions on the host like the reference's python sampler."""

    def __init__(self, sd, text_heads=16, llm_heads=16, sampling=C1.ras_sampling, lib=None, split3=False, seed=0, decode_chunk=64, weight_dtype=None):
        """sampling: a callable (scores, decoded, sampling) -> id - the reference's python sampler, run on the HOST once per token like the reference's loop (llm/llm.py:196-223:
        the parity hook; its draws come from torch's global RNG) - or one of the strings "greedy" / "ras": the DEVICE sampler of csrc/llm_kernels.h (arg-max, or
        repetition-aware sampling top_p 0.8 / top_k 25 / win 10 / tau_r 0.1 with the library's counter RNG keyed by `seed` + request count), which keeps the whole decode
        loop on the device (cv_lm1_decode: tokens come back every `decode_chunk` steps).  Same decisions as the python sampler on the same probabilities; the draw stream
        is the device's own (as for Qwen2LM).
        weight_dtype = torch.bfloat16: the model's fp16 mode (the reference: `CosyVoice(model_dir, fp16=True)` -> cli/model.py:60-63 `self.llm.half()`), here W16A32 -
        every matrix and embedding table (tensors of two or more dimensions) is ROUNDED to bf16; prefill multiplies the rounded values as fp32, the decode step streams
        them as bf16 (cv_lm1_use_bf16: half the bytes per token).  Activations, biases, norms, cache, logits stay fp32: the tokens are those of the torch-eager port
        (cosyvoice1.py) over `self.sd`, the rounded state dict."""
        assert weight_dtype in (None, torch.float32, torch.bfloat16)
        self.w16 = weight_dtype == torch.bfloat16
        if self.w16:
            sd = {k: (v.to(torch.bfloat16).float() if torch.is_floating_point(v) and v.dim() >= 2 else v) for k, v in sd.items()}
        self.k = K = Kernels(lib, split3)
        self.seed, self.decode_chunk, self._request, self._uniforms = int(seed), int(decode_chunk), 0, None
        self.sd = sd
        self.text_encoder = EspnetEncoder(sd, "text_encoder.", text_heads, "conformer", causal=True, kern=K)
        self.llm = EspnetEncoder(sd, "llm.", llm_heads, "transformer", kern=K)
        self.speech_token_size = sd["llm_decoder.weight"].shape[0] - 1
        self.llm_input_size = sd["llm_embedding.weight"].shape[1]
        self.sos, self.task_id, self.eos_token = 0, 1, self.speech_token_size
        self.sampling = sampling
        self.text_emb, self.llm_emb, self.speech_emb = K.put(sd["text_embedding.weight"]), K.put(sd["llm_embedding.weight"]), K.put(sd["speech_embedding.weight"])
        self.affine = K.mat(sd["text_encoder_affine_layer.weight"], sd["text_encoder_affine_layer.bias"])
        self.spk_affine = K.mat(sd["spk_embed_affine_layer.weight"], sd["spk_embed_affine_layer.bias"])
        self.decoder = K.mat(sd["llm_decoder.weight"], sd["llm_decoder.bias"])
        # the decode step behind one C entry point (cv_lm1_step); fused_step = False keeps the launch-per-operator tape (A/B and test knob)
        self._host_logits = None
        self.step = self.llm.make_step(self.decoder, bf16=self.w16)
        self.fused_step = self.step is not None
        self.lock = threading.Lock()                             # one Kernels object (its recorder, its workspaces) per stage: requests on one stage object are serialised
        # The device-resident loop keeps its state (sampler state, sampled tokens, next input row, the bound KV rows) INSIDE a cv_lm1 handle, and the stage lock is
        # released between two chunks of a request: a request therefore takes a handle of its own for as long as its loop is open (with one shared handle a
        # second request's loop_begin - or a host-sampler step's rebind - in that gap made the first request decode on the other's sampler state and cache).  Handles
        # are cheap (they point at the encoder's weight tensors and own a few rows of workspace) and go back to this pool when the generator ends or is closed;
        # `self.step` stays the handle of the host-sampler path, which rebinds on every call.
        self._loop_steps, self._pool_lock, self._step_options = [], threading.Lock(), {}

    def set_step_option(self, name, value):
        """cv_lm1_set_option on every decode-step handle of this stage: the host-sampler handle, the pooled loop handles, and the ones made later."""
        self._step_options[name] = int(value)
        with self._pool_lock:
            for st in [self.step] + self._loop_steps:
                if st is not None:
                    st.lib.cv_lm1_set_option(st.h, name.encode(), C.c_int32(int(value)))

    def step_stat(self, name):
        """A counter (cv_lm1_stat) summed over the stage's handles that are at rest (the host-sampler handle + the pooled loop handles)."""
        with self._pool_lock:
            return sum(st.stat(name) for st in [self.step] + self._loop_steps if st is not None)

    def _to_host(self, logits):
        """The step's logits on the host (called under the stage lock).  On the GPU: into one pinned buffer, asynchronously, then one stream synchronisation - a pageable
        `.cpu()` stages through the runtime's own bounce buffer and costs ~3x as much per token."""
        if logits.device.type != "cuda":
            return logits.cpu()
        if self._host_logits is None or self._host_logits.numel() != logits.numel():
            self._host_logits = torch.empty(logits.numel(), dtype=F32, pin_memory=True)
        self._host_logits.copy_(logits, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        return self._host_logits

    def encode_text(self, ids):
        """text_encoder + text_encoder_affine_layer (llm.py:84-91) for one unpadded id sequence -> [n, D] on the device."""
        x = self.text_encoder.forward(self.k.gather(self.text_emb, ids))
        return self.k.linear(x, self.affine, x.shape[0])

    @torch.inference_mode()
    def inference(self, text, text_len, prompt_text, prompt_text_len, prompt_speech_token, prompt_speech_token_len, embedding,
                  sampling=25, max_token_text_ratio=20, min_token_text_ratio=2, uuid=""):
        # The stage lock is taken PER STEP / PER CHUNK, never across a yield: a request's state is its own (_KVState, the recorded step with its buffers,
        # and - device loop - a cv_lm1 handle of its own from the pool), so another request may prefill or step between two of this one's tokens, and a consumer that
        # stops reading blocks nobody.
        K, D = self.k, self.llm_input_size
        with self.lock:
            ids = torch.cat([prompt_text, text], dim=1).reshape(-1).to(torch.int32)
            n_text, n_prompt = int(text.shape[1]), int(prompt_speech_token.shape[1])
            has_spk = embedding.shape[0] != 0
            L = 1 + int(has_spk) + ids.numel() + 1 + n_prompt
            lm_input = K.new(L, D)                                  # [sos | speaker | encoded text | task id | prompt speech tokens]
            r = 0
            lm_input[r:r + 1].copy_(self.llm_emb[self.sos:self.sos + 1]); r += 1
            if has_spk:
                e = torch.nn.functional.normalize(embedding.float().cpu(), dim=1)     # 192 numbers: normalised on the host
                K.linear(K.put(e), self.spk_affine, 1, out=lm_input[r:]); r += 1
            enc = self.encode_text(ids)
            lm_input[r:r + ids.numel()].copy_(enc); r += ids.numel()
            lm_input[r:r + 1].copy_(self.llm_emb[self.task_id:self.task_id + 1]); r += 1
            if n_prompt:
                lm_input[r:r + n_prompt].copy_(K.gather(self.speech_emb, prompt_speech_token))
        min_len, max_len = int(n_text * min_token_text_ratio), int(n_text * max_token_text_ratio)
        if isinstance(self.sampling, str) and self.fused_step and max_len > 0 and L >= 2:
            # the loop on the device: prefill all rows but the last through the chunk path, then the last row is the loop's first input
            from .llm import SamplingC
            with self._pool_lock:
                step = self._loop_steps.pop() if self._loop_steps else None
            try:
                with self.lock:
                    if step is None:
                        step = self.llm.make_step(self.decoder, bf16=self.w16)      # (make_step succeeded for self.step: same arguments)
                        for name, value in self._step_options.items():
                            step.lib.cv_lm1_set_option(step.h, name.encode(), C.c_int32(value))
                    self._request += 1
                    _, state = self.llm.forward_chunk(lm_input[:L - 1], None)
                    sp = SamplingC(1 if self.sampling == "ras" else 0, self.eos_token, 1, min_len, max_len, 0.8, 25, 10, 0.1, self.seed + self._request,
                                   1 if self._uniforms is not None else 0)
                    self.llm.loop_begin(step, lm_input[L - 1:L].contiguous(), state, sp, self.speech_emb, self._uniforms)
                done, n_yield = False, 0
                while not done and n_yield < max_len:
                    with self.lock:
                        toks, done = self.llm.loop_steps(step, state, min(self.decode_chunk, max_len - n_yield))
                    for t in toks:
                        yield t
                    n_yield += len(toks)
            finally:                                                 # also on GeneratorExit: an abandoned request gives its handle back
                if step is not None:
                    step.bound = None                                # the next request binds its own cache rows
                    with self._pool_lock:
                        self._loop_steps.append(step)
            return
        out_tokens, state, x = [], None, lm_input
        for i in range(max_len):
            with self.lock:
                if self.fused_step and state is not None and x.shape[0] == 1 and x.is_contiguous():
                    logits = self.llm.fused_step(self.step, x, state)
                else:
                    y, state = self.llm.forward_chunk(x, state)
                    logits = K.linear(y[-1:], self.decoder, 1)
                logp = self._to_host(logits.reshape(-1)).log_softmax(dim=-1)    # the sampler draws from the host RNG, like the reference's python sampler
            if i < min_len:
                logp[self.speech_token_size] = -float("inf")
            top = self.sampling(logp, out_tokens, sampling)
            if top == self.eos_token:
                break
            yield top
            out_tokens.append(top)
            x = self.speech_emb[top:top + 1]


# ------------------------------------------------------------------------------------------------------------------------------------
# flow: MaskedDiffWithXvec (flow/flow.py:25-146)
# ------------------------------------------------------------------------------------------------------------------------------------
class ConditionalDecoder:
    """flow/decoder.py:88-291 (the non-causal U-Net estimator) for the classifier-free-guidance pair: activations [2][T][C] channel-last."""

    def __init__(self, sd, prefix, heads, kern):
        self.k = K = kern
        p = self.p = C1._P(sd, prefix)
        self.heads = heads
        self.in_channels = p("time_mlp.linear_1.weight").shape[1]
        self.t1, self.t2 = K.mat(p("time_mlp.linear_1.weight"), p("time_mlp.linear_1.bias")), K.mat(p("time_mlp.linear_2.weight"), p("time_mlp.linear_2.bias"))
        self.resnets = []                                       # every ResnetBlock1D in execution order (their time projections are batched over the Euler steps)
        self.down = [self._stage(p.sub("down_blocks.%d." % i), "down") for i in range(p.count("down_blocks."))]
        self.mid = [self._stage(p.sub("mid_blocks.%d." % i), "mid") for i in range(p.count("mid_blocks."))]
        self.up = [self._stage(p.sub("up_blocks.%d." % i), "up") for i in range(p.count("up_blocks."))]
        self.final = self._block(p.sub("final_block."))
        self.proj = K.conv_mat(p("final_proj.weight"), p("final_proj.bias"))

    def _block(self, p):
        return dict(conv=self.k.conv_mat(p("block.0.weight"), p("block.0.bias")), g=self.k.put(p("block.1.weight")), b=self.k.put(p("block.1.bias")))

    def _stage(self, p, kind):
        K, r = self.k, p.sub("0.")
        res = dict(b1=self._block(r.sub("block1.")), b2=self._block(r.sub("block2.")), mlp=K.mat(r("mlp.1.weight"), r("mlp.1.bias")),
                   rc=K.conv_mat(r("res_conv.weight"), r("res_conv.bias")), idx=len(self.resnets))
        self.resnets.append(res)
        blocks = []
        for j in range(p.count("1.")):
            t = p.sub("1.%d." % j)
            inner = t("attn1.to_q.weight").shape[0]
            assert inner == self.heads * 64, "cv_attention serves 64-wide heads"
            blocks.append(dict(ln1=(K.put(t("norm1.weight")), K.put(t("norm1.bias"))), ln3=(K.put(t("norm3.weight")), K.put(t("norm3.bias"))),
                               qkv=K.mat(torch.cat([t("attn1.to_q.weight"), t("attn1.to_k.weight"), t("attn1.to_v.weight")], 0)),
                               out=K.mat(t("attn1.to_out.0.weight"), t("attn1.to_out.0.bias")), inner=inner,
                               ff1=K.mat(t("ff.net.0.proj.weight"), t("ff.net.0.proj.bias")), ff2=K.mat(t("ff.net.2.weight"), t("ff.net.2.bias"))))
        st = dict(res=res, blocks=blocks, kind=kind)
        if kind == "mid":
            return st
        if p.get("2.conv.weight") is not None:                   # Downsample1D: Conv1d(k 3, stride 2, pad 1) / Upsample1D: ConvTranspose1d(k 4, stride 2, pad 1)
            w, b = p("2.conv.weight"), p("2.conv.bias")
            st["resample"] = K.mat(w.permute(0, 2, 1).reshape(w.shape[0], -1), b) if kind == "down" else K.tconv_mat(w, b, 2)
            st["ch"] = (w.shape[1], w.shape[0]) if kind == "down" else (w.shape[0], w.shape[1])
        else:
            st["plain"] = K.conv_mat(p("2.weight"), p("2.bias"))
        return st

    def prepare(self, t_host):
        """time_mlp(SinusoidalPosEmb(t)) for every Euler step at once, then every ResnetBlock1D's Linear(Mish(t_emb)): [n_steps, C] per block."""
        K, n = self.k, len(t_host)
        emb = K.new(n, self.in_channels)
        tv = K.put(torch.tensor(t_host, dtype=F32))             # a NAMED tensor: `_p(K.put(..))` hands the launch the address of a temporary that is freed before the call (found in round 6)
        K.lib.cv_time_sinusoid(_p(tv), _p(emb), C.c_int32(n), C.c_int32(self.in_channels), stream_ptr(K.lib))
        temb = K.linear(K.linear(emb, self.t1, n, act="silu"), self.t2, n, act="mish")       # t_emb only ever enters through Mish (matcha ResnetBlock1D.mlp)
        # one row per Euler step, the blocks' projections side by side: the estimator reads its step's row from a FIXED buffer (`tcur`, see __call__), so that the
        # launches of one step can be recorded and replayed for the others
        offs, total = [], 0
        for r in self.resnets:
            offs.append(total)
            total += (r["mlp"].n + 3) // 4 * 4
        tall = K.zeros(n, total)
        for r, off in zip(self.resnets, offs):
            K.linear(temb, r["mlp"], n, out=tall[:, off:], ldc=total)
        return tall, offs

    def _run_block(self, blk, x, T, col_add=None):
        K = self.k
        return K.group_norm(K.conv(x, blk["conv"], 2, T, pad=1), 2, T, blk["conv"].n, 8, blk["g"], blk["b"], act="mish", col_add=col_add)

    def _run_stage(self, st, x, T, tcur, offs):
        K, r = self.k, st["res"]
        h = self._run_block(r["b1"], x, T, col_add=tcur[offs[r["idx"]]:])
        h = self._run_block(r["b2"], h, T)
        x = K.linear(x, r["rc"], 2 * T, res=h)                                                # res_conv (1 x 1) + block2's output
        Cc = r["rc"].n
        for b in st["blocks"]:
            inner, H = b["inner"], self.heads
            qkv = K.linear(K.layer_norm(x, *b["ln1"], 1e-5), b["qkv"], 2 * T)
            v4 = lambda j: qkv.view(2, T, 3 * inner)[:, :, j * inner:(j + 1) * inner].unflatten(2, (H, 64))
            o = K.attention(v4(0), v4(1), v4(2), Tq=T, Tk=T, H=H, B=2, scale=0.125)
            x = K.linear(o, b["out"], 2 * T, res=x)
            y = K.linear(K.layer_norm(x, *b["ln3"], 1e-5), b["ff1"], 2 * T, act="gelu_erf")
            x = K.linear(y, b["ff2"], 2 * T, res=x)
        return x, Cc

    def __call__(self, h, T, tcur, offs):
        """h: packed input [2][T][in_channels] -> [2][T][mel] (the mask of an unpadded batch-1 request is all ones).  tcur: this step's row of prepare()'s
        matrix (a fixed buffer the caller refreshes per step), offs: where each ResnetBlock1D's projection starts in it."""
        K, x, hiddens = self.k, h, []
        for st in self.down:
            x, Cc = self._run_stage(st, x, T, tcur, offs)
            hiddens.append((x, T, Cc))
            if "resample" in st:
                x, T = K.conv_stride(x, st["resample"], 2, T, Cc, k=3, stride=2, pad=1)
            else:
                x = K.conv(x, st["plain"], 2, T, pad=1)
        for st in self.mid:
            x, Cc = self._run_stage(st, x, T, tcur, offs)
        for st in self.up:
            skip, Ts, Cs = hiddens.pop()
            cat = K.new(2, Ts, Cc + Cs)                                                       # x[:, :, :Ts] ++ skip along channels
            K.lib.cv_concat_cols(_p(x), C.c_int32(Cc), C.c_int64(T * Cc), _p(skip), C.c_int32(Cs), C.c_int64(Ts * Cs), _p(cat), C.c_int32(Ts), C.c_int32(2), stream_ptr(K.lib))
            x, Cc = self._run_stage(st, cat, Ts, tcur, offs)
            T = Ts
            if "resample" in st:
                x, T = K.conv_transpose(x, st["resample"], 2, T, st["ch"][0], st["ch"][1], k=4, stride=2, pad=1)
                Cc = st["ch"][1]
            else:
                x = K.conv(x, st["plain"], 2, T, pad=1)
        x = self._run_block(self.final, x, T)
        return K.linear(x, self.proj, 2 * T), T


class EstimatorHandle:
    """The same U-Net inside ONE library handle (csrc/flow.hip, cfg.estimator == 2: `unet1_forward`), round 6: the launches of an estimator evaluation are sequenced in C++
    and the ten evaluations of a solve replay as one hipGraph, the transformer blocks run on the kernels of the CosyVoice2 estimator.  precision "fp32": fp32 weights,
    every product at fp32 accuracy - the launch-per-operator class above, sequenced natively; "bf16": Linear / Conv1d operands rounded to bf16 where they are staged, the
    fused transformer-block kernels of flow_fused.h / flow_big.h / flow_band.h and the bf16 flash attention (the analogue of the reference's `fp16=True` for this model:
    cli/cosyvoice.py:27-56 loads the flow encoder as fp16 TorchScript and the estimator as an fp16 TensorRT engine)."""

    def __init__(self, sd, prefix, heads, lib, precision="fp32", cfg_rate=0.7, _tensors=None):
        from .flow import FlowConfigC
        from .llm import register_tensors
        from . import weights as Wt
        assert precision in ("fp32", "bf16")
        self.lib, self.precision, self.dev = lib, precision, torch.device(lib.device)
        dtype = torch.bfloat16 if precision == "bf16" else F32
        if _tensors is None:
            packed, dims = Wt.pack_unet1(sd, prefix, heads, self.dev, dtype)
            _tensors = ({k: lib.hook(v) for k, v in packed.items()}, dims)
        self._packed = _tensors                                   # (device tensors, dims): what a second handle over the same weights takes
        tensors, dims = _tensors
        self.mel = 80
        c = FlowConfigC(0, 0, 1, 0, 0, 0, 0, self.mel, dims["C"], heads, dims["n_blocks"], dims["n_mid"], 0, 0, cfg_rate, 2)
        self._h = C.c_void_p()
        lib.cv_flow_create(C.byref(self._h), C.byref(c))
        register_tensors(lib, "cv_flow_set_tensor", self._h, tensors)
        lib.cv_flow_finalize(self._h)
        lib.cv_flow_set_option(self._h, b"bf16_mfma", C.c_int32(int(precision == "bf16")))
        # one request at a time and launches of 5 us and more: the host stays ahead of the GPU, and replaying the ~6000-node graph of a solve costs more than issuing it
        # (MI355X, T = 1011: 40.5 ms eager / 43.0 as a graph in bf16 mode, 80 / 85 in fp32 - profiles/r6_cv1_flow_handle.txt); option "use_graph" = 1 turns it on
        lib.cv_flow_set_option(self._h, b"use_graph", C.c_int32(0))

    def set_option(self, name, value):
        self.lib.cv_flow_set_option(self._h, name.encode(), C.c_int32(int(value)))

    def stat(self, name):
        v = C.c_int64(0)
        self.lib.cv_flow_get_stat(self._h, name.encode(), C.byref(v))
        return v.value

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self.lib.raw("cv_flow_destroy", None)(self._h)
                self._h = None
        except Exception:
            pass

    def forward(self, x, mask, mu, t, spks, cond):
        """ConditionalDecoder.forward (flow/decoder.py:204-291) in the reference's layouts: x, mu, cond [2, 80, T], mask [2, 1, T] (all ones), t [2], spks [2, 80] -> [2, 80, T]."""
        T = x.shape[2]
        assert bool((mask == 1).all()), "batch-1 requests: the mask is all ones"
        args = [self.lib.hook(a.to(self.dev, F32).contiguous()) for a in (x, mask, mu, t, spks, cond)]
        out = self.lib.hook(torch.empty(2, self.mel, T, dtype=F32, device=self.dev))
        self.lib.cv_flow_estimator(self._h, *[_p(a) for a in args], C.c_int32(T), C.c_int32(0), _p(out), stream_ptr(self.lib))
        return out

    def solve(self, x, mu, spk, cond, T, n_steps):
        """x [T, 80] channel-last (in: z, out: the mel after n_steps Euler steps), mu / cond [T, 80], spk [1, 80]: ConditionalCFM.solve_euler (flow_matching.py:71-124)."""
        self.lib.cv_flow_solve(self._h, _p(x), _p(mu), _p(spk), _p(cond), C.c_int32(T), C.c_int32(n_steps), stream_ptr(self.lib))
        return x


class MaskedDiffWithXvec(C1.MaskedDiffWithXvec):
    def __init__(self, sd, enc_heads=8, est_heads=8, input_frame_rate=50, n_timesteps=10, inference_cfg_rate=0.7, lib=None, split3=False, estimator="handle",
                 precision="fp32"):
        """estimator: "handle" = the U-Net inside one library handle, the Euler solve as one hipGraph (EstimatorHandle, round 6); "operators" = one launch per operator
        from this file (ConditionalDecoder: the first form, kept as the A/B reference of the handle).  precision ("handle" only): "fp32" | "bf16", see EstimatorHandle."""
        assert estimator in ("handle", "operators")
        self.k = K = Kernels(lib, split3)
        self.sd, self.input_frame_rate, self.n_timesteps, self.cfg_rate = sd, input_frame_rate, n_timesteps, inference_cfg_rate
        self.encoder = EspnetEncoder(sd, "encoder.", enc_heads, "conformer", kern=K)
        self.estimator = ConditionalDecoder(sd, "decoder.estimator.", est_heads, K) if estimator == "operators" else \
            EstimatorHandle(sd, "decoder.estimator.", est_heads, K.lib, precision, inference_cfg_rate)
        self.precision = precision if estimator == "handle" else "fp32"
        self.output_size = sd["encoder_proj.weight"].shape[0]
        self.input_emb = K.put(sd["input_embedding.weight"])
        self.spk_affine = K.mat(sd["spk_embed_affine_layer.weight"], sd["spk_embed_affine_layer.bias"])
        self.enc_proj = K.mat(sd["encoder_proj.weight"], sd["encoder_proj.bias"])
        p = C1._P(sd, "length_regulator.model.")
        last = max(int(k[len(p.prefix):].split(".")[0]) for k in sd if k.startswith(p.prefix))
        self.reg = [(K.conv_mat(p("%d.weight" % i), p("%d.bias" % i)), K.put(p("%d.weight" % (i + 1))), K.put(p("%d.bias" % (i + 1)))) for i in range(0, last, 3)]
        self.reg_out = K.conv_mat(p("%d.weight" % last), p("%d.bias" % last))
        self.lock = threading.Lock()

    def _interp(self, h, a, b, out, row, size):
        """out[row : row + size] = F.interpolate(h[a:b] over time, size) (length_regulator.py:52-70), channel-last rows."""
        K, Cc = self.k, h.shape[1]
        K.lib.cv_interp_rows(_p(h[a:]), _p(out[row:]), C.c_int32(Cc), C.c_int32(b - a), C.c_int32(size), C.c_int32(Cc), stream_ptr(K.lib))

    def _regulate(self, x, T):
        K = self.k
        for conv, g, b in self.reg:
            x = K.group_norm(K.conv(x, conv, 1, T, pad=1), 1, T, conv.n, 1, g, b, act="mish")
        return K.linear(x, self.reg_out, T)

    @torch.inference_mode()
    def inference(self, token, token_len, prompt_token, prompt_token_len, prompt_feat, prompt_feat_len, embedding, flow_cache):
        with self.lock:                                          # concurrent requests of one CosyVoiceModel share this stage object
            return self._inference(token, prompt_token, prompt_feat, embedding, flow_cache)

    def _inference(self, token, prompt_token, prompt_feat, embedding, flow_cache):
        assert token.shape[0] == 1
        K, mel = self.k, self.output_size
        e = torch.nn.functional.normalize(embedding.float().cpu(), dim=1)
        spk = K.linear(K.put(e), self.spk_affine, 1)                                           # [1, 80]
        n1, n2 = int(prompt_token.shape[1]), int(token.shape[1])
        ids = torch.cat([prompt_token.reshape(-1), token.reshape(-1)]).long().clamp(min=0)
        enc = self.encoder.forward(K.gather(self.input_emb, ids))
        h = K.linear(enc, self.enc_proj, n1 + n2)                                              # [n1 + n2, 80]
        mel_len1, mel_len2 = int(prompt_feat.shape[1]), int(n2 / self.input_frame_rate * 22050 / 256)
        T = mel_len1 + mel_len2
        x = K.new(T, mel)
        if n1 != 0:
            self._interp(h, 0, n1, x, 0, mel_len1)
        edge = int(20 / self.input_frame_rate * 22050 / 256)
        if n2 > 40:                                            # head / middle / tail stretched separately: the 20-token overlap of consecutive chunks maps to the same frames
            self._interp(h, n1, n1 + 20, x, mel_len1, edge)
            self._interp(h, n1 + 20, n1 + n2 - 20, x, mel_len1 + edge, mel_len2 - 2 * edge)
            self._interp(h, n1 + n2 - 20, n1 + n2, x, T - edge, edge)
        else:
            self._interp(h, n1, n1 + n2, x, mel_len1, mel_len2)
        mu = self._regulate(x, T)                                                              # [T, 80]
        cond = K.zeros(T, mel)
        if mel_len1:
            cond[:mel_len1].copy_(prompt_feat.reshape(mel_len1, mel).to(K.dev, F32))
        feat, flow_cache = self._cfm(mu, spk, cond, T, mel_len1, flow_cache)
        out = K.new(mel, mel_len2)                                                             # [T][80] rows mel_len1.. -> the reference's [1, 80, mel_len2]
        K.lib.cv_transpose(_p(feat[mel_len1:]), _p(out), C.c_int32(mel_len2), C.c_int32(mel), stream_ptr(K.lib))
        return out.unsqueeze(0), flow_cache

    def _cfm(self, mu, spk, cond, T, prompt_len, cache):
        """ConditionalCFM.forward + solve_euler (flow/flow_matching.py:36-124) with the prompt / overlap flow cache of CosyVoice-300M."""
        K, mel = self.k, self.output_size
        z = torch.randn(1, mel, T)                               # drawn on the host like the reference's `.to(mu.device)`: same stream on every device
        cache = torch.zeros(1, mel, 0, 2) if cache is None else cache.cpu()
        n_cache = cache.shape[2]
        if n_cache != 0:
            z[:, :, :n_cache] = cache[:, :, :, 0]
            mu[:n_cache].copy_(cache[0, :, :, 1].t().to(K.dev))
        mu_cf = mu.cpu().t().unsqueeze(0)
        keep = lambda a: torch.cat([a[:, :, :prompt_len], a[:, :, -34:]], dim=2)
        new_cache = torch.stack([keep(z), keep(mu_cf)], dim=-1)
        t_span = torch.linspace(0, 1, self.n_timesteps + 1, dtype=F32)
        t_span = 1 - torch.cos(t_span * 0.5 * torch.pi)
        ts, dts = [], []
        t, dt = t_span[0], t_span[1] - t_span[0]                  # the reference's fp32 running sums (flow_matching.py:94-123)
        for step in range(1, len(t_span)):
            ts.append(float(t)); dts.append(float(dt))
            t = t + dt
            if step < len(t_span) - 1:
                dt = t_span[step + 1] - t
        if isinstance(self.estimator, EstimatorHandle):          # the same schedule, recurrences and update on the device (csrc/flow.hip::solve_euler)
            return self.estimator.solve(K.put(z[0].t()), mu, spk, cond, T, self.n_timesteps), new_cache
        tall, offs = self.estimator.prepare(ts)
        tcur = K.new(tall.shape[1])
        x = K.put(z[0].t())                                                                    # [T, 80]
        h = K.new(2, T, 4 * mel)
        tape = graph = None
        for step in range(len(ts)):
            tcur.copy_(tall[step])
            K.lib.cv_pack_cfg_input(_p(x), _p(mu), _p(spk), _p(cond), _p(h), C.c_int32(T), C.c_int32(mel), stream_ptr(K.lib))
            if tape is not None:                                 # the estimator's ~700 launches: recorded at the first step, replayed on the same buffers after
                if graph is None and K.use_graphs and not K.lib.emulated and step == 1:
                    graph = tape.capture()
                if graph is not None:
                    graph.replay()
                    K.graph_replays += 1
                else:
                    tape.replay()
            elif K.use_tapes:
                with K.record() as tape:
                    d, _ = self.estimator(h, T, tcur, offs)
            else:
                d, _ = self.estimator(h, T, tcur, offs)
            K.lib.cv_cfg_euler(_p(x), _p(d), C.c_int64(T * mel), C.c_float(dts[step]), C.c_float(self.cfg_rate), stream_ptr(K.lib))
        return x, new_cache


# ------------------------------------------------------------------------------------------------------------------------------------
# HiFTGenerator at 22.05 kHz (hifigan/generator.py:378-569): the shared vocoder handle (f0 predictor, conv stack, iSTFT) + the type-1 SineGen source
# ------------------------------------------------------------------------------------------------------------------------------------
class HiFTGenerator(_KernelHiFT):
    def __init__(self, state_dict, cfg, lib=None, seed=1986, rng="device", _tensors=None):
        """rng: "host" draws the SineGen phases AND noise from the global torch RNG in the reference's order (parity with the goldens of the real class);
        "device": phases from the host RNG (9 numbers), noise from the kernel's counter RNG keyed by seed + call count."""
        super().__init__(state_dict, cfg, lib=lib, seed=seed, _tensors=_tensors)
        assert cfg.sr == 22050 and not cfg.causal, "the 24 kHz generators (SineGen2) are cosyvoice_amd.hift.HiFTGenerator / CausalHiFTGenerator"
        self.rng = rng
        self._ws = None
        self.lock = threading.Lock()                             # the handle's workspaces serve one call at a time

    @torch.inference_mode()
    def inference(self, speech_feat, cache_source=None):
        with self.lock:
            return self._inference(speech_feat, cache_source)

    def _inference(self, speech_feat, cache_source):
        lib, cfg, m = self.lib, self.cfg, speech_feat.shape[2]
        L, H1 = m * self.upsample_scale, cfg.harmonics + 1
        f0 = self.f0_predictor(speech_feat)                                                    # [1, m] on the device
        phase = -np.pi + torch.rand(1, H1, 1) * (2 * np.pi)                                     # Uniform(-pi, pi).sample() of SineGen.forward
        phase[:, 0, :] = 0
        noise = None
        if self.rng == "host":
            noise = lib.hook(torch.randn(1, H1, L).to(self.device).contiguous())
            torch.randn(1, 1, L)                                                               # the reference draws (and discards) the noise branch: keep the RNG in step
        self._calls += 1
        if self._ws is None or self._ws.numel() < m * H1 * 2:
            self._ws = lib.hook(torch.zeros(max(256, m) * H1 * 2, dtype=torch.float64, device=self.device))
        s = lib.hook(torch.empty(1, 1, L, dtype=F32, device=self.device))
        ph = lib.hook(phase.reshape(-1).to(self.device, F32).contiguous())
        lib.cv_sinegen1_source(_p(f0), C.c_int32(m), C.c_int32(self.upsample_scale), C.c_int32(cfg.harmonics), C.c_float(cfg.sr), _p(ph), _p(noise),
                               C.c_uint64(self.seed + self._calls), _p(self._tensors["source.w"]), _p(self._tensors["source.b"]), C.c_float(cfg.nsf_alpha),
                               C.c_float(cfg.nsf_sigma), C.c_float(cfg.voiced_thr), _p(s), _p(self._ws), stream_ptr(lib))
        if cache_source is not None and cache_source.shape[2] != 0:
            s[:, :, :cache_source.shape[2]] = cache_source.to(self.device)
        return self.decode(speech_feat, s), s


# ------------------------------------------------------------------------------------------------------------------------------------
class CosyVoiceModel(C1.CosyVoiceModel):
    """cli.model.CosyVoiceModel (cli/model.py:27-242) over the kernel-backed stages: same `load`, `tts`, `token2wav`, per-uuid state."""

    def load(self, llm_model, flow_model, hift_model, hift_cfg=None, lib=None, fp16=False, **kw):
        """fp16: the reference's switch for this model (cli/cosyvoice.py:27-56 `CosyVoice(model_dir, fp16=True)`: cli/model.py:60-63 halves the LM and the flow, the
        estimator runs as an fp16 TensorRT engine).  Here: the LM's matrices as bf16 with fp32 activations (TransformerLM weight_dtype), the flow estimator in bf16 mode
        (EstimatorHandle precision "bf16"); the vocoder stays fp32 like the reference's."""
        from .configs import cv1
        ld = lambda f: {k: v.float() for k, v in torch.load(f, map_location="cpu", weights_only=True).items()}
        self.fp16 = bool(fp16)
        self.llm = TransformerLM(ld(llm_model), lib=lib, weight_dtype=torch.bfloat16 if fp16 else None, **{k: kw[k] for k in ("text_heads", "llm_heads") if k in kw})
        self.flow = MaskedDiffWithXvec(ld(flow_model), lib=lib, precision="bf16" if fp16 else "fp32", **{k: kw[k] for k in ("enc_heads", "est_heads", "input_frame_rate") if k in kw})
        self.hift = HiFTGenerator({k.replace("generator.", ""): v for k, v in ld(hift_model).items()}, hift_cfg or cv1()[1], lib=lib, **kw.get("hift", {}))
