"""Host mirror of cosyvoice.llm.llm.Qwen2LM for the inference path (boundary B2).

Same call surface as the reference (`inference(text, text_len, prompt_text, prompt_text_len, prompt_speech_token,
prompt_speech_token_len, embedding, sampling=25, max_token_text_ratio=20, min_token_text_ratio=2, uuid='')` yielding
Python ints), but the whole autoregressive loop — backbone, speech-token head, repetition-aware sampling, stop test —
runs on the MI355X behind cv_llm_* (cosyvoice_amd/csrc/llm.hip); this class only builds `lm_input` with gather kernels
and drains tokens in chunks.
"""
import ctypes as C
import os
import threading

import torch

from . import weights as Wt
from ._lib import CV_BF16, get_lib, stream_ptr


class LLMConfigC(C.Structure):
    _fields_ = [("hidden", C.c_int32), ("layers", C.c_int32), ("heads", C.c_int32), ("kv_heads", C.c_int32), ("inter", C.c_int32),
                ("speech_vocab", C.c_int32), ("max_len", C.c_int32), ("rms_eps", C.c_float), ("rope_theta", C.c_float)]


class SamplingC(C.Structure):
    _fields_ = [("mode", C.c_int32), ("eos", C.c_int32), ("n_stop", C.c_int32), ("min_len", C.c_int32), ("max_len", C.c_int32),
                ("top_p", C.c_float), ("top_k", C.c_int32), ("win_size", C.c_int32), ("tau_r", C.c_float), ("seed", C.c_uint64),
                ("use_uniforms", C.c_int32)]


def register_tensors(lib, set_fn, handle, tensors):
    for name, t in tensors.items():
        dt = {torch.float32: 0, torch.bfloat16: 1, torch.int32: 2, torch.uint8: 3}[t.dtype]
        getattr(lib, set_fn)(handle, name.encode(), C.c_void_p(t.data_ptr()), C.c_int32(dt), C.c_int64(t.numel()))


class _Cursor:
    """Admission cursor shared by the decode groups of one inference_queue call: hands out request indices in order, once.  `cancel()` (the consumer abandoned the
    generator, or one chain failed) stops further admissions: the chains finish what they have in flight and end.
    `first` (round 6): chain k's FIRST first[k] admissions come from a block of its own - [0, first[0]) for chain 0, the next first[1] indices for chain 1, ... - and
    only then from the shared remainder.  With the requests sorted by length bound (tts_queue's default order) every chain starts on one length bucket: its
    sequences finish in the same decode chunks and share flow passes, instead of each chain holding a mix of all lengths."""

    def __init__(self, n, first=None):
        self.n, self.lock, self.cancelled = n, threading.Lock(), False
        self.blocks = []
        at = 0
        for f in (first or []):
            self.blocks.append([at, min(n, at + f)])
            at = min(n, at + f)
        self.k = at                                             # the shared remainder starts behind the blocks

    def take(self, chain=None):
        with self.lock:
            if self.cancelled:
                return None
            if chain is not None and chain < len(self.blocks) and self.blocks[chain][0] < self.blocks[chain][1]:
                self.blocks[chain][0] += 1
                return self.blocks[chain][0] - 1
            if self.k < self.n:
                self.k += 1
                return self.k - 1
            for b in self.blocks:                               # the remainder is empty: help out with what another chain has not started yet
                if b[0] < b[1]:
                    b[1] -= 1
                    return b[1]
            return None

    def view(self, chain):
        cur = self

        class _View:
            def take(self):
                return cur.take(chain)
        return _View()

    def cancel(self):
        with self.lock:
            self.cancelled = True


class Qwen2Encoder:
    """Boundary B2, the finer hook: `Qwen2LM.llm.forward_one_step(xs, masks, cache)` of the reference (cosyvoice/llm/llm.py:242-254) over the handle's device KV
    cache, so that the reference's OWN decode loop (llm.py:535-549: forward_one_step -> llm_decoder -> sampling_ids -> speech_embedding) can run unchanged on top of
    the kernels - with `Qwen2LM.llm_decoder` and `Qwen2LM.speech_embedding` below.  The device loop of `Qwen2LM.inference` is the fast path; this one costs a
    host round trip per token, exactly like the reference.

    xs: [1, n, hidden] fp32 (n >= 1: a prompt, an appended text / speech mix, or one token embedding).  cache: None starts a new sequence; otherwise the object the
    previous call returned (the KV cache lives in the handle: one live sequence per Qwen2LM object, a stale cache object is refused).  masks: accepted and
    ignored - the reference always passes the lower-triangular mask of everything forwarded so far (llm.py:540), which is what the kernels compute.
    Returns (y [1, 1, hidden], cache): the final-norm hidden state of the LAST position - the only row the reference's callers read (`y_pred[:, -1]`)."""

    def __init__(self, lm):
        self.lm, self._gen, self._pos = lm, 0, 0

    @torch.inference_mode()
    def forward_one_step(self, xs, masks=None, cache=None):
        from .ops import norm_rows
        lm = self.lm
        rows = lm.lib.hook(xs.reshape(-1, lm.cfg.hidden).to(lm.device, torch.float32).contiguous())
        n = rows.shape[0]
        with lm.lock:
            if cache is None:
                lm._kv_gen += 1
                self._gen, self._pos = lm._kv_gen, 0
            elif cache != ("cv_llm_kv", id(self), self._gen) or self._gen != lm._kv_gen:      # replaced by another forward_one_step sequence, or by inference() / inference_batch() / prefill() on the same handle
                raise ValueError("forward_one_step: this cache belongs to a sequence that was replaced (one live KV cache per Qwen2LM handle)")
            if self._pos + n + 2 >= lm.max_len:
                raise ValueError("forward_one_step: KV capacity %d exhausted" % lm.max_len)
            fn = lm.lib.cv_llm_prefill if cache is None else lm.lib.cv_llm_prefill_append
            fn(lm._h, C.c_void_p(rows.data_ptr()), C.c_int32(n), stream_ptr(lm.lib))
            self._pos += n
            h = lm.lib.hook(lm.last_hidden().to(lm.device).reshape(1, -1))             # last row of the residual stream (the final norm is fused into the head GEMV of the device loop)
            y = norm_rows(lm.lib, h, gamma=lm._tensors["norm"], eps=lm.cfg.rms_eps, rms=True)
        return y.reshape(1, 1, -1), ("cv_llm_kv", id(self), self._gen)


class Qwen2LM:
    """cosyvoice/llm/llm.py:257-549 (inference side).  `sampling` is 'ras' (the yaml default, cosyvoice2.yaml:32-36) or
    'greedy' (the sampler north-star parity is defined on)."""

    def __init__(self, state_dict, cfg, lib=None, max_len=2048, sampling="ras", top_p=0.8, top_k=25, win_size=10, tau_r=0.1,
                 seed=1986, decode_chunk=16, use_graph=True, attn_splits=8, batch_fp8=False, decode_groups=2, queue_groups=0, group_min_slots=24):
        """batch_fp8 (opt-in, BASELINE.json configs[4] "fp8 MFMA LLM path"): the BATCHED decode (inference_batch / _queue / serve_stream) runs on OCP
        e4m3 copies of the weight matrices (one fp32 scale per row) with the activations quantised per sequence in the kernel and
        v_mfma_f32_16x16x32_fp8_fp8 products; prefill and the single-sequence path keep the bf16 weights.  Token ids are then no longer the fp32
        oracle's - the mode is held to an oracle that mirrors the quantisation (tests/test_llm_fp8.py)."""
        self.lib = lib or get_lib()
        self.cfg = cfg
        self.device = torch.device(self.lib.device)
        self.speech_token_size = cfg.speech_token_size
        self.llm_input_size = self.llm_output_size = cfg.hidden
        self.sos, self.task_id = 0, 1
        self.eos_token = cfg.speech_token_size
        self.fill_token = cfg.speech_token_size + 2
        self.stop_token_ids = [cfg.speech_token_size + i for i in range(cfg.n_special)]
        self.sampling, self.top_p, self.top_k, self.win_size, self.tau_r, self.seed = sampling, top_p, top_k, win_size, tau_r, seed
        self.decode_chunk = decode_chunk
        self.max_len = max_len
        self.lock = threading.Lock()             # one KV cache per handle: requests on one object are serialised
        self._tensors, self._host = Wt.pack_llm(state_dict, cfg, self.device)
        if batch_fp8:
            self._tensors.update(Wt.quantize_llm_fp8(self._tensors, cfg))
        self._tensors = {k: self.lib.hook(v) for k, v in self._tensors.items()}
        self._host = {k: self.lib.hook(v) for k, v in self._host.items()}
        c = LLMConfigC(cfg.hidden, cfg.layers, cfg.heads, cfg.kv_heads, cfg.inter, cfg.speech_token_size + cfg.n_special, max_len, cfg.rms_eps, cfg.rope_theta)
        self._h = C.c_void_p()
        self.lib.cv_llm_create(C.byref(self._h), C.byref(c))
        register_tensors(self.lib, "cv_llm_set_tensor", self._h, self._tensors)
        self.lib.cv_llm_finalize(self._h)
        if os.environ.get("CV_LLM_GRAPH") == "0":              # A/B knob: the decode step launched kernel by kernel
            use_graph = False
        self.lib.cv_llm_set_option(self._h, b"use_graph", C.c_int32(int(use_graph)))
        self.lib.cv_llm_set_option(self._h, b"attn_splits", C.c_int32(int(attn_splits)))   # key-range slices per head in decode attention
        self.batch_fp8 = bool(batch_fp8)
        if batch_fp8:
            self.lib.cv_llm_set_option(self._h, b"batch_fp8", C.c_int32(1))
        # Decode groups (round 5): a lock-step batch of >= `group_min_slots` sequences is cut into `decode_groups` independent chains of equal size, each on a handle
        # of its own (a SIBLING: the same weight tensors, its own KV cache / workspaces / graphs), its own stream and host thread.  Every launch of the batched step is
        # latency-bound and fills the chip only partly, so two chains side by side cost far less than their sum (profiles/r5_batch_decode_ab.txt section 4: 2 x 16 slots
        # decode 1.17 x the tokens per second of 1 x 32, 2 x 12 1.25 x those of 1 x 24; 2 x 8 is SLOWER than 1 x 16: hence group_min_slots).  Slots are independent, so a
        # request's tokens do not depend on the cut - up to the fp32 summation order of the decode attention, whose kernel form follows the slot count (inference_batch's
        # docstring; option "batch_attn" pins it).  `decode_groups` (default 2, env CV_LLM_GROUPS) applies to inference_batch, where the LM has the chip to itself;
        # `queue_groups` (default 1, env CV_LLM_QUEUE_GROUPS) to inference_queue: next to a running vocoder the idle CUs two chains would fill are already taken, and the
        # chains' sequences finish apart, so fewer of them share a flow pass (measured: mixed64 509 -> 415 audio-s/s with 2, profiles/r5_decode_groups.txt).
        self._opts = dict(use_graph=int(use_graph), attn_splits=int(attn_splits), batch_fp8=int(bool(batch_fp8)))
        self._cfg_c = c
        self.decode_groups = int(os.environ.get("CV_LLM_GROUPS", decode_groups if decode_groups is not None else 1))
        # Round 6: `queue_groups` = 0 (the default) lets the queue choose - chains of SIXTEEN slots (one MFMA column tile of the batched GEMMs: 17 .. 32 slots run the
        # two-tile kernels at a higher cost per token), up to three of them: 32 slots -> 2 x 16, 48 and more -> 3 x 16 (mixed64 on the MI355X with the round-6 attention:
        # 1 x 32 534, 2 x 16 539, 2 x 24 558, 2 x 32 572, 3 x 16 586, 3 x 20 540, 4 x 16 489 audio-s/s - profiles/r6_queue_groups.txt), each chain starting on one
        # length bucket of the sorted request list (_Cursor).  A number pins the chain count (1 = one chain, the round-5 behaviour).
        self.queue_groups = int(os.environ.get("CV_LLM_QUEUE_GROUPS", queue_groups if queue_groups is not None else 0))
        # The queue path is the one whose results are compared ACROSS slot counts and rank counts (bench.py mixed64: one hash over the utterance hashes, "the same at any
        # rank count"): it pins the decode attention to the form with ONE key partition per sequence (option batch_attn = 2), so that a request's logits are the same
        # bits in a chain of 8 and of 16 slots, cut or uncut - 1 - 4 % of the step (batch 8 / 16 / 32: 254 -> 244, 424 -> 415, 537 -> 527 audio-s/s) for a determinism
        # contract that holds by construction rather than by the margins of the day.  inference_batch / serve_stream keep the per-call rule.
        self.queue_invariant = os.environ.get("CV_LLM_QUEUE_INVARIANT", "1") != "0"
        self.batch_attn = {"0": 0, "1": 1, "2": 2}.get(os.environ.get("CV_ATTN_BATCH", ""), -1)       # what the handle's option "batch_attn" is outside a queue (-1: the per-call rule)
        self.queue_buckets = os.environ.get("CV_LLM_QUEUE_BUCKETS", "1") != "0"       # decode groups of a queue start on one length bucket each (_Cursor `first`; A/B knob)
        self.group_min_slots = int(os.environ.get("CV_LLM_GROUP_MIN", group_min_slots))
        self._siblings, self._group_streams, self._sib_lock = [], [], threading.Lock()
        self.group_streams = None                            # streams for the second .. last chain (CosyVoice2Model.set_lanes hands over its lane streams)
        self._uniforms = None
        self._request = 0
        self._kv_gen = 0                                     # bumped by everything that resets the handle's KV cache (a stale forward_one_step cache must be refused whoever replaced it)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self.lib.raw("cv_llm_destroy", None)(self._h)
                self._h = None
        except Exception:
            pass

    def sibling(self):
        """A second handle over the SAME device weight tensors (nothing is copied): its own KV cache, workspaces, graphs, lock and request counter."""
        import copy
        sib = copy.copy(self)
        for k in [k for k in sib.__dict__ if callable(getattr(type(sib), k, None))]:
            del sib.__dict__[k]                                 # an instance attribute that shadows a method (a caller's wrapper around THIS handle's method) stays with this handle
        sib.lock = threading.Lock()
        sib._siblings, sib._group_streams, sib.group_streams, sib.decode_groups, sib.queue_groups = [], [], None, 1, 1
        sib._h = C.c_void_p()
        self.lib.cv_llm_create(C.byref(sib._h), C.byref(self._cfg_c))
        register_tensors(self.lib, "cv_llm_set_tensor", sib._h, self._tensors)
        self.lib.cv_llm_finalize(sib._h)
        for k, v in self._opts.items():
            if k != "batch_fp8" or v:
                self.lib.cv_llm_set_option(sib._h, k.encode(), C.c_int32(v))
        sib._uniforms, sib._kv_gen = None, 0
        return sib

    def _groups(self, n_slots, groups):
        """The handles a lock-step batch of `n_slots` sequences is cut over ([self] = no cut) and the stream each runs on (None: the caller's)."""
        g = groups if n_slots >= self.group_min_slots and not self.batch_fp8 and self._uniforms is None else 1
        g = max(1, min(g, n_slots))
        if g == 1:
            return [self], [None]
        with self._sib_lock:                                    # (two server threads may ask for the first cut at the same time)
            while len(self._siblings) < g - 1:
                self._siblings.append(self.sibling())
        if self.device.type == "cuda":
            # the first chain stays on the caller's stream; the others run on `group_streams` (set by the model: its token2wav lane streams, idle while tts_batch
            # decodes) or on streams of their own.  A process has 4 hardware queues and a busy stream beyond them shares one with another (set_lanes): two extra
            # streams for the chains slowed the pipelined runs that FOLLOWED a grouped batch in the same process (batch 32: 508 -> 441 audio-s/s, gpurun_out/r5w).
            extra = list(self.group_streams or [])[: g - 1]
            own = 0
            while len(extra) < g - 1:                           # (streams of its own are made once and kept)
                with self._sib_lock:
                    if len(self._group_streams) <= own:
                        self._group_streams.append(torch.cuda.Stream(self.device, priority=-1))
                extra.append(self._group_streams[own])
                own += 1
            return [self] + self._siblings[: g - 1], [torch.cuda.current_stream(self.device)] + extra
        return [self] + self._siblings[: g - 1], [None] * g

    def _run_groups(self, works):
        """works: [(handle, stream, callable)] - each callable on a thread of its own under its stream; the first exception is re-raised."""
        errs = []
        ev = None
        if self.device.type == "cuda":
            ev = torch.cuda.Event(); ev.record()                # what the caller's stream has queued (uploaded request tensors) precedes every chain

        def run(stream, fn):
            try:
                if stream is not None:
                    with torch.cuda.stream(stream):
                        stream.wait_event(ev)
                        fn()
                        stream.synchronize()
                else:
                    fn()
            except BaseException as e:                          # noqa: BLE001 - surfaces in the caller
                errs.append(e)

        ths = [threading.Thread(target=run, args=(st, fn), daemon=True) for _, st, fn in works]
        [t.start() for t in ths]
        [t.join() for t in ths]
        if errs:
            raise errs[0]

    # ---- lm_input = [sos | embed_tokens(prompt_text ++ text) | task_id | speech_embedding(prompt)]  (llm.py:472-494)
    def build_lm_input(self, text, prompt_text, prompt_speech_token, stream=None):
        ids_text = self.lib.hook(torch.cat([prompt_text.reshape(-1).to(self.device, torch.int32), text.reshape(-1).to(self.device, torch.int32)]).contiguous())
        ids_sp = self.lib.hook(prompt_speech_token.reshape(-1).to(self.device, torch.int32))
        n_t, n_s, H = ids_text.numel(), ids_sp.numel(), self.cfg.hidden
        L0 = 1 + n_t + 1 + n_s
        out = self.lib.hook(torch.empty(L0, H, dtype=torch.float32, device=self.device))
        st = stream if stream is not None else stream_ptr(self.lib)
        fixed = self.lib.hook(torch.tensor([self.sos, self.task_id], dtype=torch.int32, device=self.device))

        def gather(table, ids, row0):
            if ids.numel() == 0:
                return
            self.lib.cv_gather_rows(C.c_void_p(table.data_ptr()), C.c_int32(CV_BF16), C.c_int64(table.shape[0]), C.c_int32(H),
                                    C.c_void_p(ids.data_ptr()), C.c_int32(ids.numel()), C.c_void_p(out[row0:].data_ptr()), C.c_float(1.0), st)
        special = self._special_table()                  # llm_embedding (Qwen2LM) or speech_embedding (CosyVoice3LM)
        gather(special, fixed[0:1], 0)
        gather(self._host["embed.text"], ids_text, 1)
        gather(special, fixed[1:2], 1 + n_t)
        gather(self._tensors["embed.speech"], ids_sp, 2 + n_t)
        self._keep = (ids_text, ids_sp, fixed)
        return out

    def _special_table(self):
        return self._host["embed.llm"]

    def warmup(self):
        """Run one prefill + one decode step on the CURRENT stream so that the decode hipGraph for this stream is captured now
        (single-threaded) rather than inside the first request, when CosyVoice2Model runs the LLM on its own thread next to
        token2wav.  The graph does not depend on the request (sampling parameters live in device memory)."""
        with self.lock:
            x = self.lib.hook(torch.zeros(4, self.cfg.hidden, dtype=torch.float32, device=self.device))
            self.prefill(x)
            self.decode(1, SamplingC(0, self.cfg.speech_token_size, self.cfg.n_special, 1, 2, self.top_p, self.top_k, self.win_size, self.tau_r, 0, 0))

    def set_uniforms(self, u):
        """Parity hook: explicit uniform variates (2 per step) instead of the on-device counter RNG."""
        self._uniforms = None if u is None else torch.as_tensor(u, dtype=torch.float32).contiguous()

    def prefill(self, lm_input, stream=None):
        st = stream if stream is not None else stream_ptr(self.lib)
        self._kv_gen += 1                                # a new sequence takes the handle's KV cache: caches handed out by forward_one_step go stale
        self.lib.cv_llm_prefill(self._h, C.c_void_p(lm_input.data_ptr()), C.c_int32(lm_input.shape[0]), st)

    def last_logits(self):
        out = torch.empty(self.cfg.speech_token_size + self.cfg.n_special, dtype=torch.float32)
        self.lib.cv_llm_last_logits(self._h, C.c_void_p(out.data_ptr()), stream_ptr(self.lib))
        return out

    def last_hidden(self):
        out = torch.empty(self.cfg.hidden, dtype=torch.float32)
        self.lib.cv_llm_last_hidden(self._h, C.c_void_p(out.data_ptr()), stream_ptr(self.lib))
        return out

    # ---- the reference's module attributes used by its own decode loop (llm/llm.py:535-549); see Qwen2Encoder
    @property
    def llm(self):
        if getattr(self, "_encoder", None) is None:
            self._encoder = Qwen2Encoder(self)
        return self._encoder

    @torch.inference_mode()
    def llm_decoder(self, y):
        """nn.Linear(hidden, speech_token_size + specials) (llm.py:283): y [..., hidden] -> logits [..., V] on the device (bf16 weights, fp32-exact product)."""
        from .ops import gemm_conv
        w, H = self._tensors["head.w"], self.cfg.hidden
        rows = self.lib.hook(y.reshape(-1, H).to(self.device, torch.float32).contiguous())
        out = gemm_conv(self.lib, rows, w, w.shape[1], M=rows.shape[0], N=w.shape[0], K=H, bias=self._tensors["head.b"])
        return out.reshape(*y.shape[:-1], w.shape[0])

    @torch.inference_mode()
    def speech_embedding(self, ids):
        """nn.Embedding(speech_token_size + specials, hidden) (llm.py:287): int ids [...] -> [..., hidden] fp32 on the device."""
        ids = torch.as_tensor(ids)
        return self._rows(self._tensors["embed.speech"], ids.reshape(-1)).reshape(*ids.shape, self.cfg.hidden)

    def decode(self, n_steps, sp, stream=None):
        buf = (C.c_int32 * max(n_steps, 1))()
        n_out, fin = C.c_int32(0), C.c_int32(0)
        st = stream if stream is not None else stream_ptr(self.lib)
        self.lib.cv_llm_decode(self._h, C.c_int32(n_steps), C.byref(sp), buf, C.byref(n_out), C.byref(fin), st)
        return list(buf[: n_out.value]), bool(fin.value)

    def clamp_max_len(self, L0, min_len, max_len, what="request"):
        """The reference's `max_len` is only a loop bound that is almost never reached (llm/llm.py:499-500, 538); the KV cache here is a
        fixed-capacity buffer, so the bound is clamped to what still fits instead of refusing the request.  Raises only when not even
        `min_len` tokens fit (eos is suppressed below min_len, so such a request could not terminate inside the cache)."""
        room = self.max_len - L0 - 2
        if room < max(min_len, 1):
            raise ValueError("%s: prompt (%d) + min_len (%d) exceeds the KV capacity %d" % (what, L0, min_len, self.max_len))
        return min(max_len, room)

    def reserve_keys(self, n):
        """n consecutive draw-stream keys of this handle (base + 1 .. base + n) for the requests of one batch / queue; returns base.  Under a lock: overlapping callers
        (two server threads, two queues) never see the same keys."""
        with self._sib_lock:
            base = self._request
            self._request = base + int(n)
        return base

    def make_sampling(self, min_len, max_len, seed_key=None):
        """Sampling parameters of one request.  The draw stream of the device sampler is keyed by `seed + k`: k = the handle's running request count, or - `seed_key`,
        a request's `seed_key` entry in inference_batch / inference_queue - a number the CALLER ties to the request, so that under 'ras' sampling a request's tokens do
        not depend on the order in which a queue admitted it (tts_queue passes the request's index in its list)."""
        if seed_key is None:
            with self._sib_lock:
                self._request += 1
        # `eos` of the C sampler = first special id = the index sampling_ids masks while ignore_eos (llm/llm.py:150-160: literally
        # `speech_token_size`, which is eos for Qwen2LM and - a quirk kept as is - sos for CosyVoice3LM); n_stop ids from there stop decoding
        sp = SamplingC(1 if self.sampling == "ras" else 0, self.cfg.speech_token_size, self.cfg.n_special, min_len, max_len, self.top_p, self.top_k, self.win_size,
                       self.tau_r, self.seed + (self._request if seed_key is None else int(seed_key)), 1 if self._uniforms is not None else 0)
        if self._uniforms is not None:
            self.lib.cv_llm_set_uniforms(self._h, C.c_void_p(self._uniforms.data_ptr()), C.c_int32(min(self._uniforms.numel(), 2 * self.max_len)), stream_ptr(self.lib))
        return sp

    @torch.inference_mode()
    def inference(self, text, text_len, prompt_text, prompt_text_len, prompt_speech_token, prompt_speech_token_len, embedding=None,
                  sampling=25, max_token_text_ratio=20, min_token_text_ratio=2, uuid="", first_chunk=None):
        """Generator of Python ints, one per speech token (llm/llm.py:458-502, 535-549).  `first_chunk` (not in the reference): how many
        tokens the caller needs before it can do anything (tts(stream=True): hop + prompt pad + look-ahead) - the device loop hands tokens
        back in chunks of `decode_chunk` steps, and the first hand-back is cut to exactly that count instead of making the first audio
        chunk wait for a full decode chunk.
        The handle lock is held for the whole generation, yields included: a handle owns ONE KV cache, so a second request on this object cannot start before this one
        ends (the reference's model object has the same property through its single `cache`).  Abandon a request by closing the generator (`gen.close()` / leaving the
        `for`), which releases the lock; concurrent requests go through `inference_queue` / `serve_stream` (one cache per slot) or through separate handles."""
        n_text = int(text.shape[1])
        min_len = int(n_text * min_token_text_ratio)
        max_len = int(n_text * max_token_text_ratio)
        with self.lock:
            lm_input = self.build_lm_input(text, prompt_text, prompt_speech_token)
            max_len = self.clamp_max_len(lm_input.shape[0], min_len, max_len)
            self.prefill(lm_input)
            sp = self.make_sampling(min_len, max_len)
            emitted = 0
            while emitted < max_len:
                chunk = self.decode_chunk if (first_chunk is None or emitted > 0) else max(1, min(int(first_chunk), self.decode_chunk))
                toks, fin = self.decode(min(chunk, max_len - emitted + 1), sp)
                for t in toks:
                    yield int(t)
                emitted += len(toks)
                if fin:
                    break


    # ------------------------------------------------------------------------------------------------ lock-step batched decode
    @torch.inference_mode()
    def inference_batch(self, requests, max_token_text_ratio=20, min_token_text_ratio=2, _cut=True):
        """Up to 8 requests decoded in lock step on this handle (BASELINE.json configs[2]/[3]; the reference batches through vLLM,
        cli/model.py:281-290): every weight matrix is streamed once per step for all sequences (llm_batch_kernels.h).  `requests` is a
        list of dicts with `text`, `prompt_text`, `prompt_speech_token` ([1, n] id tensors) and, optionally, per-request `min_token_text_ratio` /
        `max_token_text_ratio`.  Returns one token list per request - the tokens `inference()` yields for that request alone: per sequence the same products are summed,
        and every GEMM sums them in the same order whatever the batch; the decode ATTENTION has two kernel forms with different fp32 summation orders, chosen per decode
        call from the slot count and the longest context (csrc/llm.hip batch_decode), so a request's logits can differ in their last bits with the batch composition
        and with the cut into decode groups.  Every test and bench workload (all 64 mixed64 requests, 32 slots, the groups) yields the oracle's ids either way; a caller
        that needs independence down to near-tie decisions on real weights pins one form: `lib.cv_llm_set_option(h, b"batch_attn", 0 | 1)`."""
        nb = len(requests)
        assert 1 <= nb <= (16 if self.batch_fp8 else 32), "1..32 requests per batch (16 on the fp8 path)"
        # (_cut=False: this call IS one chain of a cut batch.  An argument, not handle state: concurrent callers of one handle must not see each other's cut)
        handles, streams = self._groups(nb, self.decode_groups) if _cut else ([self], [None])
        if len(handles) > 1:
            # decode groups: contiguous runs of the requests ordered by their length bound (a chain runs as long as its longest member: similar lengths together), every
            # request with the sampler key it would have had on this handle alone
            bound = lambda i: int(requests[i]["text"].shape[1]) * float(requests[i].get("max_token_text_ratio", max_token_text_ratio))
            order = sorted(range(nb), key=lambda i: -bound(i))
            base = self.reserve_keys(nb)
            reqs = [dict(requests[i], seed_key=requests[i].get("seed_key", base + 1 + i)) for i in range(nb)]
            g = len(handles)
            parts = [order[k * nb // g:(k + 1) * nb // g] for k in range(g)]
            outs = [None] * nb

            def work(h, part):
                def fn():
                    for i, toks in zip(part, type(h).inference_batch(h, [reqs[i] for i in part], max_token_text_ratio, min_token_text_ratio, _cut=False)):
                        outs[i] = toks
                return fn
            self._run_groups([(h, st_, work(h, part)) for h, st_, part in zip(handles, streams, parts)])
            return outs
        assert 1 <= nb <= (16 if self.batch_fp8 else 32), "1..32 requests per batch (16 on the fp8 path)"
        with self.lock:
            st = stream_ptr(self.lib)
            self._kv_gen += 1
            self.lib.cv_llm_batch_begin(self._h, C.c_int32(nb), st)
            max_lens, inputs, sps = [], [], []
            for i, r in enumerate(requests):
                lm_input = self.build_lm_input(r["text"], r["prompt_text"], r["prompt_speech_token"])
                n_text = int(r["text"].shape[1])
                min_len = int(n_text * r.get("min_token_text_ratio", min_token_text_ratio))
                max_len = int(n_text * r.get("max_token_text_ratio", max_token_text_ratio))
                if max_len > 0:
                    max_len = self.clamp_max_len(lm_input.shape[0], min_len, max_len, "request %d" % i)
                inputs.append(lm_input); sps.append(self.make_sampling(min_len, max(max_len, 1), r.get("seed_key")))
                max_lens.append(max_len)
            self._prefill_slots(list(range(nb)), inputs, sps, st)
            outs, fin = [[] for _ in range(nb)], [m == 0 for m in max_lens]
            chunk = self.decode_chunk
            while not all(fin):
                buf = (C.c_int32 * (nb * chunk))()
                n_out, f = (C.c_int32 * nb)(), (C.c_int32 * nb)()
                self.lib.cv_llm_batch_decode(self._h, C.c_int32(chunk), buf, n_out, f, st)
                for i in range(nb):
                    outs[i].extend(int(buf[i * chunk + k]) for k in range(n_out[i]))
                    fin[i] = fin[i] or bool(f[i]) or len(outs[i]) >= max_lens[i]
            return [o[:m] for o, m in zip(outs, max_lens)]


    def _prefill_slots(self, slots, inputs, sps, st):
        """One prefill pass for several slots (cv_llm_batch_prefill_many): the prompts are stacked row-wise, every GEMM of the prefill sees
        M = sum of the prompt lengths and the weights are read once."""
        if not slots:
            return
        rows = self.lib.hook(torch.cat(inputs, 0).contiguous()) if len(inputs) > 1 else inputs[0]
        n = len(slots)
        self.lib.cv_llm_batch_prefill_many(self._h, C.c_int32(n), (C.c_int32 * n)(*slots), C.c_void_p(rows.data_ptr()),
                                           (C.c_int32 * n)(*[int(x.shape[0]) for x in inputs]), (SamplingC * n)(*sps), st)

    @torch.inference_mode()
    def inference_queue(self, requests, slots=8, max_token_text_ratio=20, min_token_text_ratio=2, _cursor=None):
        """Continuous batching over the lock-step decoder: any number of requests, at most `slots` (<= 8) in flight; when a
        sequence finishes its slot is re-filled from the queue (a normal prefill parked into the free slot) while the other sequences keep
        decoding.  Yields (request_index, token_list) in completion order; every token list equals `inference()` of that request alone."""
        n = len(requests)
        if n == 0:
            return
        if _cursor is None:
            want = min(slots, n)
            g = self.queue_groups if self.queue_groups > 0 else max(1, min(3, want // 16))
            if self.queue_groups <= 0 and g > 1:
                slots = 16 * g                                  # chains of exactly sixteen (see __init__)
            handles, streams = self._groups(min(slots, n), g)
            if self.queue_invariant and not self.batch_fp8:    # one key partition per sequence whatever the slot count (see __init__); every form keeps its own captured step
                for h in handles:
                    self.lib.cv_llm_set_option(h._h, b"batch_attn", C.c_int32(2))
                try:
                    yield from self._inference_queue(requests, slots, max_token_text_ratio, min_token_text_ratio, None, handles, streams)
                finally:
                    for h in handles:
                        self.lib.cv_llm_set_option(h._h, b"batch_attn", C.c_int32(self.batch_attn))
                return
        else:
            handles, streams = [self], [None]
        yield from self._inference_queue(requests, slots, max_token_text_ratio, min_token_text_ratio, _cursor, handles, streams)

    @torch.inference_mode()
    def _inference_queue(self, requests, slots, max_token_text_ratio, min_token_text_ratio, _cursor, handles, streams):
        n = len(requests)
        if len(handles) > 1:
            # decode groups: every chain runs this generator over ONE shared admission cursor (the requests keep their order of admission) with slots / groups slots
            import queue as _q
            g = len(handles)
            cur, res, END = _Cursor(n, first=[(slots + g - 1 - k) // g for k in range(g)] if self.queue_buckets else None), _q.Queue(), object()
            base = self.reserve_keys(n)                         # overlapping queues on one handle must not be handed the same draw-stream keys
            reqs = [dict(r, seed_key=r.get("seed_key", base + 1 + i)) for i, r in enumerate(requests)]

            def work(h, k):
                def fn():
                    try:
                        for item in type(h).inference_queue(h, reqs, slots=(slots + g - 1 - k) // g, max_token_text_ratio=max_token_text_ratio, min_token_text_ratio=min_token_text_ratio, _cursor=cur.view(k)):
                            res.put(item)
                    except BaseException:                       # noqa: BLE001 - a failed chain stops the admissions of the others: the error surfaces as soon as they drain
                        cur.cancel()
                        raise
                    finally:
                        res.put(END)
                return fn
            errs = []

            def runner():
                try:
                    self._run_groups([(h, st_, work(h, k)) for k, (h, st_) in enumerate(zip(handles, streams))])        # (_cursor marks each inner call as one chain)
                except BaseException as e:                      # noqa: BLE001
                    errs.append(e)
                    for _ in range(g):
                        res.put(END)
            th = threading.Thread(target=runner, daemon=True)
            th.start()
            ended = 0
            try:
                while ended < g:
                    item = res.get()
                    if item is END:
                        ended += 1
                    else:
                        yield item
            finally:                                            # GeneratorExit included: an abandoned queue admits nothing more, its chains end after their chunk in flight
                cur.cancel()
            th.join()
            if errs:
                raise errs[0]
            return
        assert 1 <= slots <= (16 if self.batch_fp8 else 32)
        with self.lock:
            st = stream_ptr(self.lib)
            nb = min(slots, n)
            self._kv_gen += 1
            self.lib.cv_llm_batch_begin(self._h, C.c_int32(nb), st)
            owner, outs, limit = [None] * nb, {}, {}
            nxt = 0

            pending = []                                      # (slot, lm_input, sampling) admitted but not yet prefilled

            def fill(slot):
                nonlocal nxt
                while True:
                    if _cursor is not None:
                        i = _cursor.take()
                        if i is None:
                            break
                    elif nxt < n:
                        i = nxt
                        nxt += 1
                    else:
                        break
                    r = requests[i]
                    lm_input = self.build_lm_input(r["text"], r["prompt_text"], r["prompt_speech_token"])
                    n_text = int(r["text"].shape[1])
                    min_len = int(n_text * r.get("min_token_text_ratio", min_token_text_ratio))
                    max_len = int(n_text * r.get("max_token_text_ratio", max_token_text_ratio))
                    if max_len == 0:
                        done.append((i, []))
                        continue
                    max_len = self.clamp_max_len(lm_input.shape[0], min_len, max_len, "request %d" % i)
                    pending.append((slot, lm_input, self.make_sampling(min_len, max_len, r.get("seed_key"))))
                    owner[slot], outs[i], limit[i] = i, [], max_len
                    return
                owner[slot] = None

            def flush():                                      # every slot freed in the same decode chunk is re-filled by one prefill pass
                if pending:
                    self._prefill_slots([p[0] for p in pending], [p[1] for p in pending], [p[2] for p in pending], st)
                    pending.clear()

            done = []
            for s_ in range(nb):
                fill(s_)
            flush()
            chunk = self.decode_chunk
            while True:
                for item in done:
                    yield item
                done = []
                if all(o is None for o in owner):
                    return
                buf = (C.c_int32 * (nb * chunk))()
                n_out, f = (C.c_int32 * nb)(), (C.c_int32 * nb)()
                self.lib.cv_llm_batch_decode(self._h, C.c_int32(chunk), buf, n_out, f, st)
                for s_ in range(nb):
                    i = owner[s_]
                    if i is None:
                        continue
                    outs[i].extend(int(buf[s_ * chunk + k]) for k in range(n_out[s_]))
                    if bool(f[s_]) or len(outs[i]) >= limit[i]:
                        done.append((i, outs.pop(i)[: limit[i]]))
                        fill(s_)
                flush()


    @torch.inference_mode()
    def serve_stream(self, source, on_tokens, slots=8, max_token_text_ratio=20, min_token_text_ratio=2, step_chunk=None):
        """Continuous batching with STREAMED tokens, the LLM half of a Triton-free serving scheduler (the reference gets this from vLLM /
        TensorRT-LLM: cli/model.py:281-290, runtime/triton_trtllm/model_repo/cosyvoice2/1/model.py:307-313).  `source` is a queue.Queue of
        (key, request) items - requests as for inference_batch - closed by a None item; `on_tokens(key, new_tokens, finished, error)` is called
        from this thread after every decode chunk of `step_chunk` lock-step steps (shorter when a request's optional "first_chunk" token count falls
        inside the chunk) for every sequence that produced tokens or finished; a callback that returns True cancels the sequence (its slot is
        re-filled like a finished one's).  Free
        slots are re-filled as soon as a sequence ends (a normal prefill parked into the slot); with nothing in flight the call blocks on the
        queue.  Every sequence yields exactly the tokens `inference()` yields for its request alone."""
        import queue as _q
        assert 1 <= slots <= (16 if self.batch_fp8 else 32)
        chunk = step_chunk or min(self.decode_chunk, 8)
        with self.lock:
            st = stream_ptr(self.lib)
            self._kv_gen += 1
            self.lib.cv_llm_batch_begin(self._h, C.c_int32(slots), st)
            owner, emitted, limit, first = [None] * slots, {}, {}, {}
            closed = False

            def admit(slot, item):
                key, r = item
                try:
                    lm_input = self.build_lm_input(r["text"], r["prompt_text"], r["prompt_speech_token"])
                    n_text = int(r["text"].shape[1])
                    min_len = int(n_text * r.get("min_token_text_ratio", min_token_text_ratio))
                    max_len = int(n_text * r.get("max_token_text_ratio", max_token_text_ratio))
                    if max_len == 0:
                        on_tokens(key, [], True, None)
                        return False
                    max_len = self.clamp_max_len(lm_input.shape[0], min_len, max_len, "request %r" % (key,))
                    sp = self.make_sampling(min_len, max_len)
                    self.lib.cv_llm_batch_prefill(self._h, C.c_int32(slot), C.c_void_p(lm_input.data_ptr()), C.c_int32(lm_input.shape[0]), C.byref(sp), st)
                except Exception as e:                      # a bad request must not take the server down: report it on its own channel
                    on_tokens(key, [], True, e)
                    return False
                owner[slot], emitted[key], limit[key], first[key] = key, 0, max_len, int(r.get("first_chunk", 0))
                return True

            while True:
                # fill free slots; block only when nothing is decoding
                for s_ in range(slots):
                    while owner[s_] is None and not closed:
                        idle = all(o is None for o in owner)
                        try:
                            item = source.get(block=idle)
                        except _q.Empty:
                            break
                        if item is None:
                            closed = True
                            break
                        admit(s_, item)
                if all(o is None for o in owner):
                    if closed:
                        return
                    continue
                # a request that announced the token count of its FIRST audio chunk gets its tokens handed over the moment that count is
                # reached, not at the end of a full decode chunk (up to chunk - 1 steps later)
                n = chunk
                for key in owner:
                    if key is not None and 0 < first[key] - emitted[key] < n:
                        n = first[key] - emitted[key]
                buf = (C.c_int32 * (slots * n))()
                n_out, f = (C.c_int32 * slots)(), (C.c_int32 * slots)()
                self.lib.cv_llm_batch_decode(self._h, C.c_int32(n), buf, n_out, f, st)
                for s_ in range(slots):
                    key = owner[s_]
                    if key is None:
                        continue
                    room = limit[key] - emitted[key]
                    toks = [int(buf[s_ * n + k]) for k in range(min(n_out[s_], room))]
                    emitted[key] += len(toks)
                    fin = bool(f[s_]) or emitted[key] >= limit[key]
                    cancel = False
                    if toks or fin:
                        cancel = on_tokens(key, toks, fin, None) is True     # the client went away: the slot is handed to the next request
                    if fin or cancel:
                        owner[s_] = None
                        emitted.pop(key), limit.pop(key), first.pop(key)

    # ------------------------------------------------------------------------------------------------ bi-directional streaming
    def _rows(self, table, ids):
        """Embedding rows [n, hidden] fp32 on the device (cv_gather_rows), n may be 0."""
        ids = self.lib.hook(torch.as_tensor(ids, dtype=torch.int32).reshape(-1).to(self.device).contiguous())
        out = self.lib.hook(torch.empty(ids.numel(), self.cfg.hidden, dtype=torch.float32, device=self.device))
        if ids.numel():
            self.lib.cv_gather_rows(C.c_void_p(table.data_ptr()), C.c_int32(CV_BF16), C.c_int64(table.shape[0]), C.c_int32(self.cfg.hidden),
                                    C.c_void_p(ids.data_ptr()), C.c_int32(ids.numel()), C.c_void_p(out.data_ptr()), C.c_float(1.0), stream_ptr(self.lib))
        return out

    def _forward_rows(self, rows, first):
        """forward_one_step(lm_input, cache) of the reference for a multi-row lm_input: a fresh prefill or an append."""
        rows = self.lib.hook(rows.contiguous())
        if self._bistream_pos + rows.shape[0] + 2 >= self.max_len:
            raise ValueError("inference_bistream: KV capacity %d exhausted" % self.max_len)
        if first:
            self._kv_gen += 1
            self.lib.cv_llm_prefill(self._h, C.c_void_p(rows.data_ptr()), C.c_int32(rows.shape[0]), stream_ptr(self.lib))
        else:
            self.lib.cv_llm_prefill_append(self._h, C.c_void_p(rows.data_ptr()), C.c_int32(rows.shape[0]), stream_ptr(self.lib))
        self._bistream_pos += rows.shape[0]

    @torch.inference_mode()
    def inference_bistream(self, text, prompt_text, prompt_text_len, prompt_speech_token, prompt_speech_token_len, embedding=None,
                           sampling=25, max_token_text_ratio=20, min_token_text_ratio=2, mix_ratio=(5, 15)):
        """cosyvoice/llm/llm.py:551-661: `text` is a generator of [1, n] id tensors arriving while speech is already being produced.
        Text and speech interleave 5 : 15 (mix_ratio); a fill token (forced every 15 speech tokens, or sampled) means "feed the next 5
        text ids"; when the text ends, the rest of the text + task_id is appended and decoding runs to eos.

        Mapping onto the device loop: the reference iteration is [forward(lm_input) -> head -> sample]; a device step is
        [head -> sample -> forward(embedding of the sample)], i.e. the same pipeline shifted by half an iteration - a sampled
        special id never runs the backbone (`done`), exactly as the reference never forwards it.  Multi-row inputs (text / prompt
        mixes) go through cv_llm_prefill_append.  `lm_input` of the reference stays visible after it was forwarded and is
        forwarded AGAIN in front of the final text (:642) - that is reproduced through `stale`."""
        sts, H = self.cfg.speech_token_size, self.cfg.hidden
        fill, eos = self.fill_token, self.eos_token
        BIG = 1 << 30
        with self.lock:
            special = self._special_table()
            sos_row, task_row = self._rows(special, [self.sos]), self._rows(special, [self.task_id])
            prompt_sp = self._rows(self._tensors["embed.speech"], prompt_speech_token.reshape(-1))
            prompt_ids = prompt_text.reshape(-1)
            lm_input = sos_row                                  # the reference's `lm_input`
            forwarded = False                                   # ... and whether it has been forwarded already (then it is "stale")
            if self.cfg.cv3:                                    # :583-588
                pl = prompt_ids.tolist()
                assert self.cfg.endofprompt_id in pl, "<|endofprompt|> not detected in CosyVoice3 prompt_text, check your input!"
                k = pl.index(self.cfg.endofprompt_id)
                lm_input = torch.cat([lm_input, self._rows(self._host["embed.text"], prompt_ids[: k + 1])], 0)
                prompt_ids = prompt_ids[k + 1:]
            text_cache = self._rows(self._host["embed.text"], prompt_ids)
            n_prompt = int(prompt_speech_token.shape[1])
            next_fill_index = (int(n_prompt / mix_ratio[1]) + 1) * mix_ratio[1] - n_prompt
            out_tokens = []
            self._bistream_pos, started = 0, False

            def forward_pending():
                nonlocal forwarded, started
                if not forwarded:
                    self._forward_rows(lm_input, first=not started)
                    started, forwarded = True, True

            def run_steps(n, ignore_eos):
                """up to n device steps; returns (real tokens, stop id or None)"""
                sp = SamplingC(1 if self.sampling == "ras" else 0, sts, self.cfg.n_special, BIG if ignore_eos else 0, BIG, self.top_p, self.top_k,
                               self.win_size, self.tau_r, self.seed + self._request, 0)
                toks, fin = self.decode(n, sp)
                self._bistream_pos += len(toks)
                stop = self.lib.raw("cv_llm_last_stop_token", C.c_int)(self._h) if fin else None
                return toks, (stop if fin and stop is not None and stop >= 0 else None)

            self._request += 1
            for this_text in text:
                text_cache = torch.cat([text_cache, self._rows(self._host["embed.text"], this_text.reshape(-1))], 0)
                while prompt_sp.shape[0] != 0:                                                  # :591-600
                    if text_cache.shape[0] >= mix_ratio[0]:
                        lm_input = torch.cat([lm_input, text_cache[: mix_ratio[0]], prompt_sp[: mix_ratio[1]]], 0)
                        assert not forwarded
                        text_cache, prompt_sp = text_cache[mix_ratio[0]:], prompt_sp[mix_ratio[1]:]
                    else:
                        break
                if prompt_sp.shape[0] != 0:
                    continue
                if (out_tokens and out_tokens[-1] == fill) or (not out_tokens and lm_input.shape[0] == 1):   # :602-614
                    if text_cache.shape[0] >= mix_ratio[0]:
                        chunk = text_cache[: mix_ratio[0]]
                        if out_tokens and out_tokens[-1] == fill:
                            lm_input, forwarded = chunk, False
                        else:
                            lm_input = torch.cat([lm_input, chunk], 0)
                            assert not forwarded
                        text_cache = text_cache[mix_ratio[0]:]
                    else:
                        continue
                forward_pending()
                while True:                                                                      # :615-636
                    if next_fill_index != -1 and len(out_tokens) == next_fill_index:
                        top = fill
                        next_fill_index += mix_ratio[1] + 1
                    else:
                        budget = (next_fill_index - len(out_tokens)) if next_fill_index > len(out_tokens) else self.decode_chunk
                        toks, stop = run_steps(max(1, min(budget, self.decode_chunk)), ignore_eos=True)
                        for t in toks:
                            out_tokens.append(int(t))
                            yield int(t)
                        if toks:                                 # the device forwarded the last real token: that is the reference's lm_input now
                            lm_input, forwarded = self._rows(self._tensors["embed.speech"], [toks[-1]]), True
                        if stop is None:
                            continue
                        top = stop
                    if top == fill:
                        next_fill_index = len(out_tokens) + mix_ratio[1] + 1
                    out_tokens.append(top)
                    self.lib.cv_llm_push_token(self._h, C.c_int32(top), stream_ptr(self.lib))   # out_tokens feeds the sampler's window
                    if top == fill:
                        break
                    raise ValueError("should not get token {}".format(top))
            # 3. final decode (:638-661): the (possibly already forwarded) lm_input is fed in front of the remaining text + task_id
            lm_input, forwarded = torch.cat([lm_input, text_cache, task_row], 0), False
            forward_pending()
            while True:
                toks, stop = run_steps(self.decode_chunk, ignore_eos=False)
                for t in toks:
                    out_tokens.append(int(t))
                    yield int(t)
                if stop is None:
                    continue
                if stop == eos:
                    break
                raise ValueError("should not get token {}".format(stop))


class CosyVoice3LM(Qwen2LM):
    """cosyvoice/llm/llm.py:664-706 (inference side; SURVEY.md §8 row a17, LM part).  Same backbone and the same device kernels as
    Qwen2LM; what differs is bookkeeping: sos / eos / task_id / fill are ids speech_token_size + {0,1,2,3} whose embeddings are rows
    of `speech_embedding` (there is no llm_embedding), the head is bias-free over speech_token_size + 200 ids, every id >=
    speech_token_size stops decoding, and requests must carry <|endofprompt|> (cfg.endofprompt_id) in prompt_text ++ text."""

    def __init__(self, state_dict, cfg, **kw):
        assert cfg.cv3 and cfg.n_special == 200, "CosyVoice3LM needs a cv3 LLMConfig (configs.cv3_llm())"
        super().__init__(state_dict, cfg, **kw)
        sts = cfg.speech_token_size
        self.sos, self.eos_token, self.task_id, self.fill_token = sts + 0, sts + 1, sts + 2, sts + 3

    def _special_table(self):
        return self._tensors["embed.speech"]

    @torch.inference_mode()
    def inference(self, text, text_len, prompt_text, prompt_text_len, prompt_speech_token, prompt_speech_token_len, embedding=None, **kw):
        both = torch.cat([prompt_text.reshape(-1), text.reshape(-1)])
        assert bool((both == self.cfg.endofprompt_id).any()), "<|endofprompt|> not detected in CosyVoice3 text or prompt_text, check your input!"
        yield from super().inference(text, text_len, prompt_text, prompt_text_len, prompt_speech_token, prompt_speech_token_len, embedding, **kw)
