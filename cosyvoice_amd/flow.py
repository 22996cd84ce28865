"""Host mirror of cosyvoice.flow.flow.CausalMaskedDiffWithXvec for inference (boundaries B3/B4/B5).

`inference(token, token_len, prompt_token, prompt_token_len, prompt_feat, prompt_feat_len, embedding, streaming, finalize)`
returns `(mel[1,80,2*n_new] fp32 on the device, None)` like the reference (flow/flow.py:235-281).  `self.decoder.estimator`
and `self.encoder` are callables with the reference signatures (the objects the reference swaps for TensorRT /
TorchScript, cli/model.py:83-92,277-279) and `self.decoder.solve_euler`-style stepping happens on the device.
"""
import ctypes as C

import torch

from . import weights as Wt
from ._lib import get_lib, stream_ptr
from .llm import register_tensors


class FlowConfigC(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("vocab", "dim", "enc_heads", "ffn", "enc_blocks", "up_blocks", "spk_dim", "mel", "est_ch",
                                         "est_heads", "est_blocks", "est_mid", "pre_lookahead", "chunk")] + [("cfg_rate", C.c_float), ("estimator", C.c_int32)]


def cfm_rand_noise():
    """CausalConditionalCFM.__init__: set_all_random_seed(0); torch.randn([1, 80, 50*300])  (flow_matching.py:199-200).
    A local generator with seed 0 yields the same stream without the reference's global-RNG side effect."""
    g = torch.Generator().manual_seed(0)
    return torch.randn([1, 80, 50 * 300], generator=g)


class _Estimator:
    """flow.decoder.estimator: (x[2,80,T], mask[2,1,T], mu[2,80,T], t[2], spks[2,80], cond[2,80,T], streaming) -> [2,80,T] (boundary B3)."""

    def __init__(self, flow):
        self.flow = flow

    def __call__(self, x, mask, mu, t, spks, cond, streaming=False):
        """mask: all ones (batch-1 inference, flow/flow.py:270), or the reference's PADDED mask - per row, ones for the valid frames then zeros
        (flow/decoder.py:405-494 multiplies every block by it and builds the attention bias from it).  A padded row comes out as the reference computes
        it: its valid frames as for that row alone at its own length, zeros behind (cv_flow_estimator_masked)."""
        f = self.flow
        T = x.shape[2]
        m2 = mask.reshape(2, T).to("cpu", torch.float32)
        lens = m2.sum(dim=1).to(torch.int64)
        prefix = (torch.arange(T).unsqueeze(0) < lens.unsqueeze(1)).to(torch.float32)
        if not torch.equal(m2, prefix) or int(lens.min()) < 1:
            raise NotImplementedError("the MI355X estimator computes attention masks from indices: a mask row must be ones for its valid frames followed by zeros "
                                      "(the padded-batch mask of flow/flow.py:236-281), at least one valid frame")
        args = [f.lib.hook(a.to(f.device, torch.float32).contiguous()) for a in (x, mask, mu, t, spks, cond)]
        out = f.lib.hook(torch.empty(2, f.cfg.mel, T, dtype=torch.float32, device=f.device))
        ptr = [C.c_void_p(a.data_ptr()) for a in args]
        if int(lens.min()) == T:
            f.lib.cv_flow_estimator(f._h, *ptr, C.c_int32(T), C.c_int32(int(streaming)), C.c_void_p(out.data_ptr()), stream_ptr(f.lib))
        else:
            kl = (C.c_int32 * 2)(int(lens[0]), int(lens[1]))
            f.lib.cv_flow_estimator_masked(f._h, ptr[0], ptr[1], kl, *ptr[2:], C.c_int32(T), C.c_int32(int(streaming)), C.c_void_p(out.data_ptr()), stream_ptr(f.lib))
        return out


class EstimatorModule(torch.nn.Module):
    """The product estimator as a `torch.nn.Module`, for dropping into the REFERENCE's own `ConditionalCFM`: `forward_estimator`
    (flow/flow_matching.py:126-153) dispatches on `isinstance(self.estimator, torch.nn.Module)`; anything else is taken for a TensorRT
    wrapper (`acquire_estimator()` ...).  `flow.decoder.estimator = EstimatorModule(cosyvoice_amd_flow)` is the whole integration
    (INTEGRATION.md section 2).  Inputs keep the reference's layouts ([2,80,T] etc.) and may be reused by the caller across Euler steps
    (solve_euler :103-108 writes into the same buffers): nothing is retained past the call.  The result comes back in x's dtype on x's
    device.  There are no parameters: the weights live in the library handle owned by the wrapped flow object."""

    def __init__(self, flow):
        super().__init__()
        self._flow = [flow]                    # kept out of nn.Module's attribute registration

    def forward(self, x, mask, mu, t, spks, cond, streaming=False):
        out = self._flow[0].decoder.estimator(x, mask, mu, t, spks, cond, streaming=streaming)
        return out.to(device=x.device, dtype=x.dtype)


class EstimatorEngine:
    """Boundary B3, second form: the object the reference's `forward_estimator` takes when `flow.decoder.estimator` is NOT an nn.Module
    (flow/flow_matching.py:129-153 - the shape of its TensorRT path: `TrtContextWrapper`, utils/common.py:199-214, around an engine and its execution contexts).
    `forward_estimator` then does

        [ctx, stream], engine = estimator.acquire_estimator()
        with stream: ctx.set_input_shape(name, shape) x 6; ctx.set_tensor_address(engine.get_tensor_name(i), ptr) for x, mask, mu, t, spks, cond and the OUTPUT
                     aliased on x; ctx.execute_async_v3(current stream handle)
        estimator.release_estimator(ctx, stream)

    i.e. raw device pointers and a stream handle - exactly the arguments of cv_flow_estimator, which this object calls (the output may alias x: x is read by the
    first launch only, the result written by the last).  Swap (cli/model.py:load_trt does the same two statements around its engine):

        del model.flow.decoder.estimator
        model.flow.decoder.estimator = EstimatorEngine(cosyvoice_amd_flow)

    Like a TensorRT engine - and unlike EstimatorModule - it sees no dtypes and no `streaming` flag: buffers must be fp32 (the reference's fp16=False), the mask all
    ones (flow.inference is batch 1, flow/flow.py:270), and the attention mode is fixed at construction (`streaming=False`: full attention, what the reference's
    exported engine computes, bin/export_onnx.py:71-87).  One context (trt_concurrent = 1): concurrent callers queue on acquire_estimator, as they do on the
    reference's pool."""

    TENSOR_NAMES = ("x", "mask", "mu", "t", "spks", "cond", "estimator_out")      # bin/export_onnx.py:71-87

    class _Context:
        def __init__(self, engine):
            self.engine, self.shapes, self.addrs = engine, {}, {}

        def set_input_shape(self, name, shape):
            self.shapes[name] = tuple(int(d) for d in shape)
            return True

        def set_tensor_address(self, name, ptr):
            self.addrs[name] = int(ptr)
            return True

        def execute_async_v3(self, stream_handle):
            f = self.engine.flow
            mel = f.cfg.mel
            T = (self.shapes.get("x") or (0, 0, 0))[2]
            want = {"x": (2, mel, T), "mask": (2, 1, T), "mu": (2, mel, T), "t": (2,), "spks": (2, mel), "cond": (2, mel, T)}
            if any(self.shapes.get(k) != v for k, v in want.items()) or any(not self.addrs.get(k) for k in EstimatorEngine.TENSOR_NAMES):
                raise ValueError("EstimatorEngine: expected the estimator's six inputs as %s and seven tensor addresses, got shapes %s, addresses for %s"
                                 % (want, self.shapes, sorted(self.addrs)))
            a = self.addrs
            f.lib.cv_flow_estimator(f._h, *[C.c_void_p(a[k]) for k in ("x", "mask", "mu", "t", "spks", "cond")], C.c_int32(T), C.c_int32(int(self.engine.streaming)),
                                    C.c_void_p(a["estimator_out"]), C.c_void_p(int(stream_handle)) if stream_handle else None)
            return True

    def __init__(self, flow, streaming=False):
        import contextlib
        import queue
        self.flow, self.streaming = flow, bool(streaming)
        dev = flow.device
        stream = torch.cuda.stream(torch.cuda.Stream(dev)) if dev.type == "cuda" else contextlib.nullcontext()
        self._pool = queue.Queue(maxsize=1)
        self._pool.put([EstimatorEngine._Context(self), stream])

    def acquire_estimator(self):
        return self._pool.get(), self

    def release_estimator(self, context, stream):
        self._pool.put([context, stream])

    def get_tensor_name(self, i):
        return EstimatorEngine.TENSOR_NAMES[i]


class _Encoder(torch.nn.Module):
    """flow.encoder: (token_emb[1,n,dim], token_len, context=[1,3,dim] | empty, streaming) -> (h[1,2n,dim], mask[1,1,2n]).

    An `nn.Module` (without parameters: the weights live in the flow's library handle) because the reference's flow IS one: `model.flow.encoder = x`
    on a `torch.nn.Module` refuses anything that is not a Module for a registered child name (boundary B4, the reference's own TorchScript encoder
    swap cli/model.py:277-279 assigns a ScriptModule); tests/test_dropin_reference.py performs the swap inside the real class."""

    def __init__(self, flow):
        super().__init__()
        self._flow = [flow]                    # kept out of nn.Module's attribute registration (and no reference cycle through _modules)

    @property
    def flow(self):
        return self._flow[0]

    def output_size(self):
        return self.flow.cfg.dim

    def forward(self, xs, xs_lens, context=None, streaming=False):
        f = self.flow
        assert xs.shape[0] == 1
        n = xs.shape[1]
        xs = f.lib.hook(xs.to(f.device, torch.float32).contiguous())
        ctx = None
        if context is not None and context.numel() != 0:
            assert context.shape[1] == f.cfg.pre_lookahead
            ctx = f.lib.hook(context.to(f.device, torch.float32).contiguous())
        h = f.lib.hook(torch.empty(1, 2 * n, f.cfg.dim, dtype=torch.float32, device=f.device))
        f.lib.cv_flow_encoder(f._h, C.c_void_p(xs.data_ptr()), C.c_int32(n), C.c_void_p(ctx.data_ptr()) if ctx is not None else None,
                              C.c_int32(int(streaming)), C.c_void_p(h.data_ptr()), stream_ptr(f.lib))
        return h, torch.ones(1, 1, 2 * n, dtype=torch.bool, device=f.device)


class _CFM:
    def __init__(self, flow):
        self.estimator = _Estimator(flow)
        self.rand_noise = cfm_rand_noise()
        self.inference_cfg_rate = flow.cfg.cfg_rate
        self.t_scheduler = "cosine"


class CausalMaskedDiffWithXvec:
    def __init__(self, state_dict, cfg, lib=None, weight_dtype=torch.bfloat16, n_timesteps=None, precision="fp32", _tensors=None):
        """precision: "fp32" = W16A32 at fp32 accuracy (the exact three-term bf16 split of the activations on the bf16 matrix pipe, gemm_conv.h AX3; attention on the fp32 MFMA); "bf16" = the Linear / Conv1d operands are rounded to
        bf16 when staged into LDS and multiplied on the bf16 MFMA with fp32 accumulation (the reference's fp16 / TensorRT flow,
        cli/model.py:83-92, is the analogous mode; BASELINE.json configs[1] is quoted in bf16).  Attention, norms, the Euler update
        and every tensor in HBM stay fp32 in both modes."""
        assert precision in ("fp32", "bf16")
        self.precision = precision
        self.lib = lib or get_lib()
        self.cfg = cfg
        self.device = torch.device(self.lib.device)
        self.input_frame_rate = 25
        self.token_mel_ratio = 2
        self.pre_lookahead_len = cfg.pre_lookahead
        self.output_size = cfg.mel
        self.vocab_size = cfg.vocab
        self.n_timesteps = n_timesteps or cfg.n_timesteps
        pack = Wt.pack_flow_dit if cfg.estimator == "dit" else (lambda *a: Wt.pack_flow(*a, experiments=getattr(self.lib, "experiments", False)))
        # `_tensors`: the packed device weights of another instance (clone()): a new library handle = own workspaces / graphs, same weights
        self._tensors = _tensors if _tensors is not None else {k: self.lib.hook(v) for k, v in pack(state_dict, cfg, self.device, weight_dtype).items()}
        c = FlowConfigC(cfg.vocab, cfg.dim, cfg.enc_heads, cfg.ffn, cfg.enc_blocks, cfg.up_blocks, cfg.spk_dim, cfg.mel, cfg.est_ch,
                        cfg.est_heads, cfg.est_blocks, cfg.est_mid, cfg.pre_lookahead, cfg.chunk, cfg.cfg_rate, 1 if cfg.estimator == "dit" else 0)
        self._h = C.c_void_p()
        self.lib.cv_flow_create(C.byref(self._h), C.byref(c))
        register_tensors(self.lib, "cv_flow_set_tensor", self._h, self._tensors)
        self.lib.cv_flow_finalize(self._h)
        self.lib.cv_flow_set_option(self._h, b"bf16_mfma", C.c_int32(int(precision == "bf16")))
        self.decoder = _CFM(self)
        self.encoder = _Encoder(self)
        # channel-last copy of the fixed CFM noise, made once (not on the hot path)
        self._noise_cl = self.lib.hook(self.decoder.rand_noise[0].t().contiguous().to(self.device))

    def clone(self):
        """A second instance over the SAME device weights with its own library handle (workspaces, Euler graphs): one per token2wav lane
        of CosyVoice2Model, so that flow inferences of different requests run concurrently on different HIP streams."""
        c = type(self)(None, self.cfg, lib=self.lib, n_timesteps=self.n_timesteps, precision=self.precision, _tensors=self._tensors)
        if getattr(self, "_graph_rows", None) is not None:
            c.set_graph_rows(self._graph_rows)
        return c

    def set_graph_rows(self, n):
        """Passes of fewer than n estimator rows (2 x utterances x frames) replay a captured hipGraph of their Euler solve, larger ones are issued launch by launch
        (cv_flow option "graph_max_rows"; 1 = never capture, 0 = always).  The library's own default is 1000.  A single request is fastest without graphs (the host
        stays ahead of the launches: first chunk 56.5 -> 54.6 ms, U10 177.7 -> 175.5 ms); eight streaming clients are served best WITH them (their host threads are
        busy: p50 101.5 vs 104.9 ms) - so CosyVoice2Model asks for 1 and StreamScheduler for 3000 while it runs (profiles/r6_flow_graph_threshold.txt)."""
        self._graph_rows = int(n)
        self.lib.cv_flow_set_option(self._h, b"graph_max_rows", C.c_int32(int(n)))

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self.lib.raw("cv_flow_destroy", None)(self._h)
                self._h = None
        except Exception:
            pass

    @torch.inference_mode()
    def inference(self, token, token_len, prompt_token, prompt_token_len, prompt_feat, prompt_feat_len, embedding, streaming, finalize):
        assert token.shape[0] == 1
        ids = self.lib.hook(torch.cat([prompt_token.reshape(-1).to(self.device, torch.int32), token.reshape(-1).to(self.device, torch.int32)]).clamp(min=0).contiguous())
        n_tok = ids.numel()
        pf = self.lib.hook(prompt_feat.to(self.device, torch.float32).contiguous())
        mel_len1 = prompt_feat.shape[1]
        emb = self.lib.hook(embedding.to(self.device, torch.float32).reshape(-1).contiguous())
        n_enc = n_tok if finalize else n_tok - self.pre_lookahead_len
        mel_len2 = 2 * n_enc - mel_len1
        if mel_len2 <= 0:
            raise ValueError("no new frames to generate")
        if 2 * n_enc > self._noise_cl.shape[0]:
            # the reference slices a fixed [1, 80, 50 * 300] noise buffer (flow_matching.py:199-200, :215) and fails with a shape error beyond it
            raise ValueError("%d mel frames exceed the fixed CFM noise buffer (%d frames = %d s)" % (2 * n_enc, self._noise_cl.shape[0], self._noise_cl.shape[0] // 50))
        out = self.lib.hook(torch.empty(1, self.cfg.mel, mel_len2, dtype=torch.float32, device=self.device))
        got = C.c_int32(0)
        self.lib.cv_flow_inference(self._h, C.c_void_p(ids.data_ptr()), C.c_int32(n_tok), C.c_void_p(pf.data_ptr()) if mel_len1 else C.c_void_p(out.data_ptr()),
                                   C.c_int32(mel_len1), C.c_void_p(emb.data_ptr()), C.c_void_p(self._noise_cl.data_ptr()), C.c_int32(int(streaming)),
                                   C.c_int32(int(finalize)), C.c_int32(self.n_timesteps), C.c_void_p(out.data_ptr()), C.byref(got), stream_ptr(self.lib))
        assert got.value == mel_len2
        return out, None


    @torch.inference_mode()
    def inference_batch(self, items, streaming=False, finalize=True):
        """`items`: up to 8 dicts(token [1, n], prompt_token [1, p], prompt_feat [1, 2p', 80], embedding [1, spk]) -> list of mel
        [1, 80, 2 * n_new] tensors, each equal to what `inference()` returns for that item alone (flow/flow.py:246: "identical to running each
        utterance alone" is the reference's own contract for its batched flow).  One pass: the CFM Euler solve runs once over all items
        (estimator batch rows = 2 x items), so every GEMM of a step covers all of them."""
        nu = len(items)
        assert 1 <= nu <= 8
        n_tok = int(items[0]["prompt_token"].shape[1] + items[0]["token"].shape[1])
        mel_len1 = int(items[0]["prompt_feat"].shape[1])
        if not all(int(it["prompt_token"].shape[1] + it["token"].shape[1]) == n_tok and int(it["prompt_feat"].shape[1]) == mel_len1 for it in items):
            return self._inference_ragged(items, streaming, finalize)
        dev = self.device
        ids = self.lib.hook(torch.stack([torch.cat([it["prompt_token"].reshape(-1).to(dev, torch.int32), it["token"].reshape(-1).to(dev, torch.int32)]).clamp(min=0)
                                         for it in items]).contiguous())
        pf = self.lib.hook(torch.stack([it["prompt_feat"].to(dev, torch.float32).reshape(mel_len1, -1) for it in items]).contiguous())
        emb = self.lib.hook(torch.stack([it["embedding"].to(dev, torch.float32).reshape(-1) for it in items]).contiguous())
        n_enc = n_tok if finalize else n_tok - self.pre_lookahead_len
        mel_len2 = 2 * n_enc - mel_len1
        if mel_len2 <= 0:
            raise ValueError("no new frames to generate")
        if 2 * n_enc > self._noise_cl.shape[0]:
            raise ValueError("%d mel frames exceed the fixed CFM noise buffer (%d frames)" % (2 * n_enc, self._noise_cl.shape[0]))
        out = self.lib.hook(torch.empty(nu, self.cfg.mel, mel_len2, dtype=torch.float32, device=dev))
        got = C.c_int32(0)
        self.lib.cv_flow_inference_batch(self._h, C.c_int32(nu), C.c_void_p(ids.data_ptr()), C.c_int32(n_tok), C.c_void_p(pf.data_ptr()) if mel_len1 else C.c_void_p(out.data_ptr()),
                                         C.c_int32(mel_len1), C.c_void_p(emb.data_ptr()), C.c_void_p(self._noise_cl.data_ptr()), C.c_int32(int(streaming)),
                                         C.c_int32(int(finalize)), C.c_int32(self.n_timesteps), C.c_void_p(out.data_ptr()), C.byref(got), stream_ptr(self.lib))
        assert got.value == mel_len2
        return [out[i:i + 1] for i in range(nu)]


    def _inference_ragged(self, items, streaming, finalize):
        """Items of DIFFERENT lengths in one pass (cv_flow_inference_ragged): padded to the longest, attention limited to every item's own frames;
        each result is still bit-identical to `inference()` of the item alone."""
        dev, nu = self.device, len(items)
        toks = [torch.cat([it["prompt_token"].reshape(-1).to(dev, torch.int32), it["token"].reshape(-1).to(dev, torch.int32)]).clamp(min=0) for it in items]
        n_tok = [int(t.numel()) for t in toks]
        len1 = [int(it["prompt_feat"].shape[1]) for it in items]
        n_enc = [n if finalize else n - self.pre_lookahead_len for n in n_tok]
        len2 = [2 * e - l for e, l in zip(n_enc, len1)]
        if min(len2) <= 0:
            raise ValueError("no new frames to generate")
        if 2 * max(n_enc) > self._noise_cl.shape[0]:
            raise ValueError("%d mel frames exceed the fixed CFM noise buffer (%d frames)" % (2 * max(n_enc), self._noise_cl.shape[0]))
        ids = self.lib.hook(torch.cat(toks).contiguous())
        pf = self.lib.hook(torch.cat([it["prompt_feat"].to(dev, torch.float32).reshape(-1, self.cfg.mel) for it in items] + [torch.zeros(1, self.cfg.mel, device=dev)]).contiguous())
        emb = self.lib.hook(torch.stack([it["embedding"].to(dev, torch.float32).reshape(-1) for it in items]).contiguous())
        out = self.lib.hook(torch.empty(self.cfg.mel * sum(len2), dtype=torch.float32, device=dev))
        got = (C.c_int32 * nu)()
        self.lib.cv_flow_inference_ragged(self._h, C.c_int32(nu), C.c_void_p(ids.data_ptr()), (C.c_int32 * nu)(*n_tok), C.c_void_p(pf.data_ptr()), (C.c_int32 * nu)(*len1),
                                          C.c_void_p(emb.data_ptr()), C.c_void_p(self._noise_cl.data_ptr()), C.c_int32(int(streaming)), C.c_int32(int(finalize)),
                                          C.c_int32(self.n_timesteps), C.c_void_p(out.data_ptr()), got, stream_ptr(self.lib))
        assert list(got) == len2
        res, off = [], 0
        for l in len2:
            res.append(out[off:off + self.cfg.mel * l].view(1, self.cfg.mel, l))
            off += self.cfg.mel * l
        return res


class CausalMaskedDiffWithDiT(CausalMaskedDiffWithXvec):
    """cosyvoice.flow.flow.CausalMaskedDiffWithDiT for inference (flow/flow.py:284-414; Fun-CosyVoice3): the same
    `inference(...)` surface and the same CFM solver (CausalConditionalCFM.solve_euler with classifier-free guidance), with PreLookaheadLayer +
    repeat_interleave(2) in front instead of the conformer encoder and the DiT (22 blocks, adaLN-zero, head-0 rotary, causal grouped-conv position
    embedding) as `decoder.estimator`.  `cfg` = configs.cv3_flow()."""

    def __init__(self, state_dict, cfg, **kw):
        assert cfg.estimator == "dit", "CausalMaskedDiffWithDiT needs a DiT FlowConfig (configs.cv3_flow())"
        super().__init__(state_dict, cfg, **kw)
        self.encoder = None                      # no conformer encoder on this model (flow/flow.py:309-313)
