"""Host mirror of cosyvoice.cli.model.CosyVoice2Model (boundaries B1 / B7): same `load`, `tts`,
`token2wav`, `llm_job` call surface, same per-uuid state dicts and chunking rules, over the MI355X stages.

Differences that are deliberate:
  * C.1  `token_hop_len` is per request (the reference mutates the attribute, so a streamed request changes the next one);
  * C.2  the streaming loop waits on a condition variable signalled by the LLM thread instead of `time.sleep(0.1)`;
  * C.12 `fade_in_out` stays on the device (cv_fade_in_out) and there is no per-request `empty_cache()`.
"""
import ctypes as C
import queue
import os
import threading
import zlib
from types import GeneratorType
import uuid as uuid_mod
from contextlib import contextmanager, nullcontext

import numpy as np
import torch

from ._lib import get_lib, stream_ptr
from .flow import CausalMaskedDiffWithDiT, CausalMaskedDiffWithXvec
from .hift import CausalHiFTGenerator, HiFTGenerator
from .llm import CosyVoice3LM, Qwen2LM


class _Lane:
    """One token2wav lane: a flow + HiFT instance (own library handles = own workspaces and graphs, weights shared with lane 0) and the HIP
    stream its kernels go to (None = the caller's current stream, the single-lane default)."""
    __slots__ = ("flow", "hift", "stream")

    def __init__(self, flow, hift, stream):
        # all lanes at normal priority: putting FIRST chunks on a high-priority stream (the LM's level) starves the LM decode chain - at 8
        # streaming clients the time until a request's first 41 tokens exist went 94 -> 243 ms (profiles/r2_first_chunk_priority_ab.txt)
        self.flow, self.hift, self.stream = flow, hift, stream


class SilentTokenFilter:
    """The silent / breath-token rule of llm_job (cli/model.py:122-128) as a reusable, stateful filter: a token of `silent_tokens` is dropped once
    more than `max_run` of them came in a row; any other token resets the run.  One instance per request (the run count carries over the
    chunks a request's tokens arrive in), shared by llm_job, tts_batch, tts_queue and the serving scheduler so that every path feeds the
    vocoder exactly what tts() would."""
    __slots__ = ("silent", "max_run", "run")

    def __init__(self, silent_tokens, max_run=5):
        self.silent, self.max_run, self.run = frozenset(int(t) for t in silent_tokens), max_run, 0

    def keep(self, tok):
        if tok in self.silent:
            self.run += 1
            return self.run <= self.max_run
        self.run = 0
        return True

    def __call__(self, tokens):
        if not self.silent:
            return list(tokens)
        return [t for t in tokens if self.keep(t)]


class CosyVoice2Model:
    def __init__(self, llm, flow, hift, fp16=False, lib=None):
        """llm / flow / hift: cosyvoice_amd.{llm.Qwen2LM, flow.CausalMaskedDiffWithXvec, hift.HiFTGenerator} (or None before load())."""
        self.llm, self.flow, self.hift = llm, flow, hift
        self.lib = lib or ((llm or flow or hift).lib if (llm or flow or hift) is not None else get_lib())
        self.device = torch.device(self.lib.device)
        # reference: fp16=True halves llm + flow (cli/model.py:50-52).  Here: the flow's Linear/Conv1d products run on the bf16 MFMA
        # (precision="bf16", fp32 accumulate and fp32 tensors in HBM); the LLM is W16A32 either way (token ids stay bit-exact).
        self.fp16 = fp16
        self.token_hop_len = 25                # must match the training static_chunk_size (cli/model.py:257-259)
        self.token_max_hop_len = 4 * self.token_hop_len
        self.stream_scale_factor = 2
        self.mel_cache_len = 8
        self.source_cache_len = int(self.mel_cache_len * 480)
        self.speech_window = np.hamming(2 * self.source_cache_len)
        self._window_dev = self.lib.hook(torch.from_numpy(self.speech_window.astype(np.float32)).to(self.device))
        use_cuda = self.device.type == "cuda"
        # high priority: with several token2wav lanes the LM decode chain (one short kernel after another) must not queue behind the
        # flow / vocoder kernels of other requests - it sets every request's token rate and first-chunk latency
        self.llm_stream = torch.cuda.Stream(self.device, priority=-1) if use_cuda else None
        self.llm_context = torch.cuda.stream(self.llm_stream) if use_cuda else nullcontext()
        self.lock = threading.Lock()
        # token2wav lanes: flow / hift handles own their workspaces, so a lane serves one token2wav at a time; `set_lanes(n)` adds lanes
        # (cloned handles over the same weights, one HIP stream each) and concurrent token2wav calls then overlap on the GPU - the flow and
        # the vocoder are chains of small latency-bound kernels that leave most of the 256 CUs idle
        self.n_lanes, self._lane_q = 0, queue.Queue()
        self.flow_batch = 8                    # offline batch paths (tts_batch / tts_queue): up to this many sequences share one flow pass (4 until round 4: from 3 per pass the large-M kernels serve it, 13 vs 17 ms per utterance at 8) ...
        self.flow_pad = 1.25                   # ... when the longest of them has at most this many times the frames of the shortest (padded pass)
        # ... and, opt-in until measured on the MI355X (bench.py --hift-batch / CV_HIFT_BATCH=1), the equal-length members of such a group share ONE HiFT launch
        # sequence as well (HiFTGenerator.inference_batch; bit-identical per utterance)
        self.hift_batch = os.environ.get("CV_HIFT_BATCH", "0") == "1"
        self.set_lanes(1)
        self.tts_speech_token_dict, self.llm_end_dict, self.hift_cache_dict, self._cond = {}, {}, {}, {}
        self._llm_error = {}                   # uuid -> exception raised on the LLM thread, re-raised by tts() on the caller's thread
        self.first_chunk_exclusive = True      # tts(stream=True): the LM pauses after the first chunk's tokens until that chunk has been vocoded (llm_job)
        self._first_gate = {}                  # uuid -> threading.Event
        self.silent_tokens = []
        self._warmup()

    def set_lanes(self, n):
        """n >= 1 token2wav lanes.  Call while no request is in flight.  Keep lanes + 2 (the LM stream, the default stream) within the runtime's hardware queues
        (ROCm: 4 per process): beyond that, which streams share a queue depends on the process's history and two busy lanes may serialise
        (profiles/r3_stream_after_batch.txt).  With shared flow passes (flow_batch = 8) two lanes serve eight streaming clients.  (Lane streams restricted to a subset of the CUs with
        hipExtStreamCreateWithCUMask, to keep CUs free for the LM chain, were measured: every mask - even 224 of 256 CUs - more than doubled
        both the LM's and the vocoder's latency at 8 streaming clients, profiles/r2_lane_cu_mask_ab.txt.  Plain streams.)"""
        assert n >= 1
        while not self._lane_q.empty():
            self._lane_q.get_nowait()
        use_cuda = self.device.type == "cuda"
        for i in range(n):
            own = i == 0 or self.flow is None
            flow = self.flow if own else self.flow.clone()
            hift = self.hift if own else self.hift.clone()
            self._lane_q.put(_Lane(flow, hift, torch.cuda.Stream(self.device) if (use_cuda and n > 1) else None))
            if self.flow is None:
                break
        self.n_lanes = n
        self.set_flow_graph_rows(getattr(self, "flow_graph_rows", int(os.environ.get("CV_MODEL_GRAPH_ROWS", 1))))      # (the variable: A/B knob)
        if getattr(self, "llm", None) is not None and hasattr(self.llm, "group_streams"):
            # the LM's decode groups (Qwen2LM._groups) borrow the lane streams for their second chain: lanes are idle while tts_batch decodes, and the process stays
            # within its four hardware queues (LM stream, default stream, two lanes)
            lanes = list(self._lane_q.queue)
            self.llm.group_streams = [ln.stream for ln in reversed(lanes) if ln.stream is not None] or None

    def set_flow_graph_rows(self, n):
        """Which flow passes of this model's lanes replay a captured hipGraph (flow.set_graph_rows): 1 = none, the model's default - a request served alone is
        fastest when its solve is issued launch by launch; a scheduler that serves many concurrent streams sets 3000 (serving.StreamScheduler).  Call while no
        request is in flight (like set_lanes)."""
        self.flow_graph_rows = int(n)
        for ln in list(self._lane_q.queue):
            if ln.flow is not None and hasattr(ln.flow, "set_graph_rows"):
                ln.flow.set_graph_rows(self.flow_graph_rows)

    @contextmanager
    def _lane(self):
        """Check a lane out for one token2wav (blocks while all are busy); with several lanes the work runs on the lane's stream and is
        complete when the context exits (per-uuid cache tensors are then safe to read from any other lane)."""
        lane = self._lane_q.get()
        try:
            st = lane.stream
            if st is None:
                yield lane
            else:
                st.wait_stream(torch.cuda.current_stream(self.device))   # inputs produced on the caller's stream (front-end mel, ...) are complete first
                try:
                    with torch.cuda.stream(st):
                        yield lane
                finally:
                    st.synchronize()                                     # also when the body raised: the lane goes back idle
        finally:
            self._lane_q.put(lane)

    @staticmethod
    def _noise_key(token, token_offset):
        """Key of the vocoder's counter RNG for one token2wav call, derived from the request's own tokens: the audio of a request does not
        depend on the lane, rank or order it is served in (SURVEY.md section 8e determinism requirement)."""
        return (zlib.crc32(token.to(torch.int32).cpu().contiguous().numpy().tobytes()) << 8) + int(token_offset)

    def _warmup(self):
        if self.llm is not None:
            with self.llm_context:
                self.llm.warmup()
            if self.device.type == "cuda":
                torch.cuda.synchronize()

    # ------------------------------------------------------------------------------------------------ B7
    @classmethod
    def from_state_dicts(cls, llm_sd, flow_sd, hift_sd, cfgs, lib=None, fp16=False, **llm_kw):
        lc, fc, hc = cfgs
        lib = lib or get_lib()
        flow = CausalMaskedDiffWithXvec(flow_sd, fc, lib=lib, precision="bf16" if fp16 else "fp32")
        return cls(Qwen2LM(llm_sd, lc, lib=lib, **llm_kw), flow, HiFTGenerator(hift_sd, hc, lib=lib), fp16=fp16)

    def load(self, llm_model, flow_model, hift_model, cfgs=None, **llm_kw):
        """cli/model.py:65-73: state-dict files (llm.pt, flow.pt, hift.pt; `generator.` prefix stripped from hift keys)."""
        from .configs import cv2
        lc, fc, hc = cfgs or cv2()
        llm_sd = torch.load(llm_model, map_location="cpu", weights_only=True)
        flow_sd = torch.load(flow_model, map_location="cpu", weights_only=True)
        hift_sd = {k.replace("generator.", ""): v for k, v in torch.load(hift_model, map_location="cpu", weights_only=True).items()}
        self.llm = Qwen2LM(llm_sd, lc, lib=self.lib, **llm_kw)
        self.flow = CausalMaskedDiffWithXvec(flow_sd, fc, lib=self.lib, precision="bf16" if self.fp16 else "fp32")
        self.hift = HiFTGenerator(hift_sd, hc, lib=self.lib)
        self.set_lanes(max(1, self.n_lanes))
        self._warmup()

    # the reference's accelerator hooks are meaningless here: the MI355X kernels ARE the accelerated path
    def load_jit(self, *a, **k):
        raise NotImplementedError("TorchScript export is replaced by the native flow encoder (cv_flow_encoder)")

    def load_trt(self, *a, **k):
        raise NotImplementedError("TensorRT is replaced by the native flow estimator (cv_flow_estimator)")

    def load_vllm(self, *a, **k):
        raise NotImplementedError("vLLM is replaced by the native LLM decode loop (cv_llm_decode)")

    # ------------------------------------------------------------------------------------------------ llm_job / vc_job
    def llm_job(self, text, prompt_text, llm_prompt_speech_token, llm_embedding, uuid, first_chunk=None):
        """cli/model.py:101-129: `text` is a tensor (offline text) or a generator of [1, n] id tensors (streaming text ->
        Qwen2LM.inference_bistream)."""
        keep = SilentTokenFilter(self.silent_tokens).keep
        cond = self._cond[uuid]
        try:
            with self.llm_context:
                t = lambda n: torch.tensor([n], dtype=torch.int32)
                if isinstance(text, GeneratorType):
                    gen = self.llm.inference_bistream(text=text, prompt_text=prompt_text, prompt_text_len=t(prompt_text.shape[1]),
                                                      prompt_speech_token=llm_prompt_speech_token,
                                                      prompt_speech_token_len=t(llm_prompt_speech_token.shape[1]), embedding=llm_embedding)
                else:
                    gen = self.llm.inference(text=text, text_len=t(text.shape[1]), prompt_text=prompt_text, prompt_text_len=t(prompt_text.shape[1]),
                                             prompt_speech_token=llm_prompt_speech_token, prompt_speech_token_len=t(llm_prompt_speech_token.shape[1]),
                                             embedding=llm_embedding, uuid=uuid, **({} if first_chunk is None else {"first_chunk": first_chunk}))
                gate = self._first_gate.get(uuid)
                for i in gen:
                    if not keep(i):
                        continue
                    with cond:
                        self.tts_speech_token_dict[uuid].append(i)
                        cond.notify_all()
                        n = len(self.tts_speech_token_dict[uuid])
                    if gate is not None and first_chunk is not None and n >= first_chunk:
                        # first_chunk_exclusive: the tokens of the first audio chunk are out; the decode chain steps aside until that chunk's flow + HiFT
                        # have run (tts() sets the gate), so the listener's first audio is not stretched by the LM racing ahead on the same GPU.  The LM
                        # runs ~5x faster than playback: the tokens of the second chunk are still there long before they are needed.
                        gate.wait(timeout=30.0)          # (always set by tts(): after the first chunk, or in its finally)
                        gate = None
        except BaseException as e:             # a thread's exception would only reach threading.excepthook: hand it to tts()
            self._llm_error[uuid] = e
        finally:
            with cond:
                self.llm_end_dict[uuid] = True
                cond.notify_all()

    def vc_job(self, source_speech_token, uuid):
        with self._cond[uuid]:
            self.tts_speech_token_dict[uuid] = source_speech_token.flatten().tolist()
            self.llm_end_dict[uuid] = True
            self._cond[uuid].notify_all()

    # ------------------------------------------------------------------------------------------------ B1
    def _fade(self, speech, cached_speech):
        n = self.source_cache_len
        tail = cached_speech[:, -n:].contiguous()
        self.lib.cv_fade_in_out(C.c_void_p(speech.data_ptr()), C.c_void_p(tail.data_ptr()), C.c_void_p(self._window_dev.data_ptr()), C.c_int32(n), stream_ptr(self.lib))
        return speech

    @torch.inference_mode()
    def token2wav(self, token, prompt_token, prompt_feat, embedding, token_offset, uuid, stream=False, finalize=False, speed=1.0):
        """cli/model.py:292-326."""
        with self._lane() as lane:
            t = lambda n: torch.tensor([n], dtype=torch.int32)
            tts_mel, _ = lane.flow.inference(token=token.to(torch.int32), token_len=t(token.shape[1]), prompt_token=prompt_token, prompt_token_len=t(prompt_token.shape[1]),
                                             prompt_feat=prompt_feat, prompt_feat_len=t(prompt_feat.shape[1]), embedding=embedding, streaming=stream, finalize=finalize)
            return self._t2w_tail(lane, tts_mel, token, token_offset, uuid, finalize, speed)

    @torch.inference_mode()
    def token2wav_batch(self, jobs, stream=False, finalize=False, on_ready=None):
        """token2wav for several requests at once (round 3; the serving scheduler's chunk batches): `jobs` = dicts(token, prompt_token, prompt_feat,
        embedding, token_offset, uuid [, speed]) that share `stream` / `finalize`.  ONE flow pass over all of them (inference_batch: equal shapes
        unpadded, different lengths as the padded pass of cv_flow_inference_ragged - every mel bit-identical to the request alone), then each
        request's own HiFT call with its own cache, exactly as token2wav does.  Returns the list of tts_speech tensors, in job order;
        `on_ready(i, wav)` (optional) is called the moment job i's audio has been enqueued - a serving scheduler hands it to its listener while the
        other members of the batch are still in the vocoder (a `.cpu()` inside the callback waits on the lane's stream only)."""
        if len(jobs) == 1 or not hasattr(self.flow, "inference_batch"):
            outs = []
            for i, j in enumerate(jobs):
                outs.append(self.token2wav(stream=stream, finalize=finalize, **j))
                if on_ready is not None:
                    on_ready(i, outs[-1])
            return outs
        with self._lane() as lane:
            mels = lane.flow.inference_batch([dict(token=j["token"].to(torch.int32), prompt_token=j["prompt_token"], prompt_feat=j["prompt_feat"], embedding=j["embedding"])
                                              for j in jobs], streaming=stream, finalize=finalize)
            outs = []
            for i, (j, mel) in enumerate(zip(jobs, mels)):
                outs.append(self._t2w_tail(lane, mel, j["token"], j["token_offset"], j["uuid"], finalize, j.get("speed", 1.0)))
                if on_ready is not None:
                    on_ready(i, outs[-1])
            return outs

    def _t2w_tail(self, lane, tts_mel, token, token_offset, uuid, finalize, speed):
        """Everything of token2wav after the flow (cli/model.py:301-326): new frames -> mel / source caches -> HiFT -> fade."""
        flow, hift = lane.flow, lane.hift
        tts_mel = tts_mel[:, :, token_offset * flow.token_mel_ratio:]
        cache = self.hift_cache_dict.get(uuid)
        if cache is not None:
            tts_mel = torch.concat([cache["mel"], tts_mel], dim=2)
            hift_cache_source = cache["source"]
        else:
            hift_cache_source = torch.zeros(1, 1, 0)
        hift._next_seed = self._noise_key(token, token_offset)
        if finalize is False:
            tts_speech, tts_source = hift.inference(speech_feat=tts_mel, cache_source=hift_cache_source)
            if cache is not None:
                tts_speech = self._fade(tts_speech, cache["speech"])
            self.hift_cache_dict[uuid] = {"mel": tts_mel[:, :, -self.mel_cache_len:].clone(), "source": tts_source[:, :, -self.source_cache_len:].clone(),
                                          "speech": tts_speech[:, -self.source_cache_len:].clone()}
            tts_speech = tts_speech[:, :-self.source_cache_len]
        else:
            if speed != 1.0:
                assert cache is None, "speed change only support non-stream inference mode"
                tn = int(tts_mel.shape[2] / speed)
                src = tts_mel.contiguous()
                dst = torch.empty(1, src.shape[1], tn, dtype=torch.float32, device=self.device)
                self.lib.cv_interp_linear(C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), C.c_int32(src.shape[1]), C.c_int32(src.shape[2]), C.c_int32(tn), stream_ptr(self.lib))
                tts_mel = dst
            tts_speech, tts_source = hift.inference(speech_feat=tts_mel, cache_source=hift_cache_source)
            if cache is not None:
                tts_speech = self._fade(tts_speech, cache["speech"])
        return tts_speech

    def _vocode_group(self, group, speed):
        """Offline vocoding of one group of finished sequences [(index, request, tokens)] on ONE lane.  Groups of equal shape (token count, prompt
        tokens, prompt frames) go through the flow in one pass (CausalMaskedDiffWithXvec / WithDiT.inference_batch: the Euler solve covers all of
        them, each result identical to the utterance alone), then through HiFT one by one."""
        if len(group) == 1 or not hasattr(self.flow, "inference_batch") or type(self).token2wav not in (CosyVoice2Model.token2wav, CosyVoice3Model.token2wav):
            outs = []
            for i, r, toks in group:
                uid = str(uuid_mod.uuid1())
                self.hift_cache_dict[uid] = None
                try:
                    wav = self.token2wav(token=torch.tensor(toks).unsqueeze(0), prompt_token=r["flow_prompt_speech_token"], prompt_feat=r["prompt_speech_feat"],
                                         embedding=r["flow_embedding"], token_offset=0, uuid=uid, finalize=True, speed=speed)
                    outs.append((i, {"tts_speech": wav.cpu()}))
                finally:
                    self.hift_cache_dict.pop(uid, None)
            return outs
        with self._lane() as lane, torch.inference_mode():
            toks_t = [torch.tensor(toks, dtype=torch.int32).unsqueeze(0) for _, _, toks in group]
            mels = lane.flow.inference_batch([dict(token=t, prompt_token=r["flow_prompt_speech_token"], prompt_feat=r["prompt_speech_feat"], embedding=r["flow_embedding"])
                                              for (_, r, _), t in zip(group, toks_t)], streaming=False, finalize=True)
            outs = []
            if (self.hift_batch and speed == 1.0 and len(mels) > 1 and hasattr(lane.hift, "inference_batch") and not lane.hift.cfg.causal
                    and not getattr(lane.hift, "f0_float64", False) and len({m_.shape[2] for m_ in mels}) == 1):
                # batched vocoding (cv_hift_inference_batch): utterances of equal length share ONE HiFT launch sequence; each waveform is bit-identical to
                # _vocode_mel of it alone (its own RNG key)
                speech, _ = lane.hift.inference_batch(torch.cat(list(mels), 0), [self._noise_key(t, 0) for t in toks_t])
                wav = speech.cpu()
                return [(i, {"tts_speech": wav[k:k + 1]}) for k, (i, _, _) in enumerate(group)]
            for (i, r, _), t, mel in zip(group, toks_t, mels):
                if speed != 1.0:
                    tn = int(mel.shape[2] / speed)
                    src = mel.contiguous()
                    dst = torch.empty(1, src.shape[1], tn, dtype=torch.float32, device=self.device)
                    self.lib.cv_interp_linear(C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), C.c_int32(src.shape[1]), C.c_int32(src.shape[2]), C.c_int32(tn), stream_ptr(self.lib))
                    mel = dst
                outs.append((i, {"tts_speech": self._vocode_mel(lane, mel, t).cpu()}))
            return outs

    def _vocode_mel(self, lane, mel, token):
        """HiFT half of a one-shot token2wav on `lane` (cli/model.py:319-325 with an empty cache)."""
        lane.hift._next_seed = self._noise_key(token, 0)
        return lane.hift.inference(speech_feat=mel, cache_source=torch.zeros(1, 1, 0))[0]

    def _flow_groups(self, jobs):
        """Cut a list of finished sequences [(index, request, tokens)] into the groups that share a flow pass: longest first; a group takes the
        next sequences while their frame count stays within 1 / flow_pad of the group's longest and it has fewer than flow_batch members.  The
        padded pass (cv_flow_inference_ragged) computes every member at the longest length, so at most flow_pad - 1 of a member's work is
        padding; equal shapes need none."""
        n_tok = lambda j: len(j[2]) + int(j[1]["flow_prompt_speech_token"].shape[1])
        order = sorted(jobs, key=lambda j: (-n_tok(j), j[0]))
        k, cap = 0, max(1, self.flow_batch)
        while k < len(order):
            e = k + 1
            while e < len(order) and e - k < cap and n_tok(order[e]) * self.flow_pad >= n_tok(order[k]):
                e += 1
            yield order[k:e]
            k = e

    def _vocode_all(self, job_lists, speed):
        """job_lists: iterable of lists of (index, request, tokens) - each list holds the sequences that became available together; yields
        (index, {'tts_speech'}) as they complete.  A list is cut into groups of up to `flow_batch` sequences of similar length (equal shapes share an
        unpadded pass); the groups run on the token2wav lanes, one worker thread per lane."""
        groups = self._flow_groups
        if self.n_lanes == 1:
            for jobs in job_lists:
                for grp in groups(jobs):
                    yield from self._vocode_group(grp, speed)
            return
        from concurrent.futures import ThreadPoolExecutor
        # `job_lists` may block (tts_queue: token sequences arrive from the LM thread), so it is drained by a feeder thread and the results come
        # back through one queue: finished audio is yielded the moment its lane is done, not when the LM next hands something over.
        res = queue.Queue()
        END = object()

        def run(grp):
            try:
                res.put(self._vocode_group(grp, speed))
            except BaseException as e:
                res.put(e)

        def feed(ex):
            n = 0
            try:
                for jobs in job_lists:
                    for grp in groups(jobs):
                        ex.submit(run, grp)
                        n += 1
                res.put((END, n))
            except BaseException as e:
                res.put(e)

        with ThreadPoolExecutor(max_workers=self.n_lanes) as ex:
            feeder = threading.Thread(target=feed, args=(ex,), daemon=True)
            feeder.start()
            submitted, got = None, 0
            try:
                while submitted is None or got < submitted:
                    item = res.get()
                    if isinstance(item, BaseException):
                        raise item
                    if isinstance(item, tuple) and len(item) == 2 and item[0] is END:
                        submitted = item[1]
                        continue
                    got += 1
                    yield from item
            finally:
                feeder.join()

    def tts_batch(self, requests, speed=1.0):
        """Offline synthesis of up to 16 requests (dicts with the keyword arguments of tts()): the speech-token LM runs lock-step
        batched (Qwen2LM.inference_batch: weights streamed once per step for all requests), flow + HiFT then run per utterance, `n_lanes`
        utterances at a time (set_lanes).
        Returns one {'tts_speech': [1, S]} per request, equal to what tts(**request, stream=False) yields for it."""
        reqs = [dict(text=r["text"], prompt_text=r["prompt_text"], prompt_speech_token=r["llm_prompt_speech_token"]) for r in requests]
        with self.llm_context:
            tokens = self.llm.inference_batch(reqs)
        if self.device.type == "cuda":
            self.llm_stream.synchronize()
        outs = [None] * len(requests)
        tokens = [SilentTokenFilter(self.silent_tokens)(toks) for toks in tokens]          # what llm_job would have handed to token2wav
        for i, o in self._vocode_all([[(i, r, toks) for i, (r, toks) in enumerate(zip(requests, tokens))]], speed):
            outs[i] = o
        return outs

    def tts_queue(self, requests, slots=8, speed=1.0, order="longest_first"):
        """Throughput pipeline for many offline requests (BASELINE.json configs[2]/[3]): a producer thread runs the LM with continuous
        batching (Qwen2LM.inference_queue, <= `slots` sequences in flight, on the LLM stream) while the calling thread turns every finished
        token sequence into audio (flow + HiFT on the caller's stream) - LM decode of the next sequences overlaps the vocoding of the
        finished ones.  Yields (request_index, {'tts_speech': [1, S]}) in completion order; each waveform equals tts(**request).
        order: "longest_first" (default, round 4) admits the requests by decreasing length bound (text ids x max_token_text_ratio - what the reference's own
        stopping rule makes the expected length proportional to, llm.py:484-485): the slots do not idle behind one long request admitted last, and requests of
        similar length finish - and share flow passes - together; "fifo" admits them as listed.  The audio of a request does not depend on the order: greedy decoding
        has no random stream, and under 'ras' sampling the device sampler's stream of a request is keyed by the request's INDEX in `requests` (`seed_key`,
        Qwen2LM.make_sampling), not by when the queue admitted it.  (tts(**request) alone draws from the handle's running request count instead: equal audio between
        the two entry points is a property of greedy decoding only.)"""
        import queue
        q = queue.Queue()
        assert order in ("longest_first", "fifo"), order
        bound = lambda r: int(r["text"].shape[1]) * float(r.get("max_token_text_ratio", 20))
        perm = sorted(range(len(requests)), key=lambda i: -bound(requests[i])) if order == "longest_first" else list(range(len(requests)))       # (stable: ties keep their order)
        base = self.llm.reserve_keys(len(requests)) if hasattr(self.llm, "reserve_keys") else getattr(self.llm, "_request", 0)
        lm_reqs = [dict(text=requests[i]["text"], prompt_text=requests[i]["prompt_text"], prompt_speech_token=requests[i]["llm_prompt_speech_token"], seed_key=base + 1 + i,
                        **{k: requests[i][k] for k in ("min_token_text_ratio", "max_token_text_ratio") if k in requests[i]}) for i in perm]

        def produce():
            try:
                with self.llm_context:
                    for item in self.llm.inference_queue(lm_reqs, slots=slots):
                        q.put(item)
                q.put(None)
            except BaseException as e:          # surfaces in the consumer
                q.put(e)

        th = threading.Thread(target=produce, daemon=True)
        th.start()

        def finished():                                        # lists of the sequences that finished together (the same decode chunk)
            closed = False
            while not closed:
                items = [q.get()]
                while True:
                    try:
                        items.append(q.get_nowait())
                    except queue.Empty:
                        break
                jobs = []
                for item in items:
                    if item is None:
                        closed = True
                    elif isinstance(item, BaseException):
                        raise item
                    else:
                        jobs.append((perm[item[0]], requests[perm[item[0]]], SilentTokenFilter(self.silent_tokens)(item[1])))
                if jobs:
                    yield jobs

        try:
            yield from self._vocode_all(finished(), speed)
        finally:
            th.join()

    def tts(self, text=torch.zeros(1, 0, dtype=torch.int32), flow_embedding=torch.zeros(0, 192), llm_embedding=torch.zeros(0, 192),
            prompt_text=torch.zeros(1, 0, dtype=torch.int32), llm_prompt_speech_token=torch.zeros(1, 0, dtype=torch.int32),
            flow_prompt_speech_token=torch.zeros(1, 0, dtype=torch.int32), prompt_speech_feat=torch.zeros(1, 0, 80),
            source_speech_token=torch.zeros(1, 0, dtype=torch.int32), stream=False, speed=1.0, **kwargs):
        """cli/model.py:328-394: generator of {'tts_speech': [1, S] fp32 cpu}."""
        this_uuid = str(uuid_mod.uuid1())
        with self.lock:
            self.tts_speech_token_dict[this_uuid], self.llm_end_dict[this_uuid] = [], False
            self.hift_cache_dict[this_uuid] = None
            self._cond[this_uuid] = threading.Condition()
        cond = self._cond[this_uuid]
        if source_speech_token.shape[1] == 0:
            first_need = None
            if stream is True and self.flow is not None:       # tokens the first audio chunk waits for (cli/model.py:345-349)
                hop0 = self.token_hop_len
                first_need = int(np.ceil(flow_prompt_speech_token.shape[1] / hop0) * hop0 - flow_prompt_speech_token.shape[1]) + hop0 + self.flow.pre_lookahead_len
                if self.first_chunk_exclusive and not isinstance(text, GeneratorType):
                    self._first_gate[this_uuid] = threading.Event()
            p = threading.Thread(target=self.llm_job, args=(text, prompt_text, llm_prompt_speech_token, llm_embedding, this_uuid, first_need))
        else:
            p = threading.Thread(target=self.vc_job, args=(source_speech_token, this_uuid))
        p.start()

        def check_llm():
            err = self._llm_error.pop(this_uuid, None)
            if err is not None:
                raise err

        try:
            if stream is True:
                token_offset = 0
                token_hop_len = self.token_hop_len
                la = self.flow.pre_lookahead_len
                prompt_token_pad = int(np.ceil(flow_prompt_speech_token.shape[1] / token_hop_len) * token_hop_len - flow_prompt_speech_token.shape[1])
                while True:
                    this_token_hop_len = token_hop_len + prompt_token_pad if token_offset == 0 else token_hop_len
                    with cond:
                        cond.wait_for(lambda: len(self.tts_speech_token_dict[this_uuid]) - token_offset >= this_token_hop_len + la or self.llm_end_dict[this_uuid])
                        n_have = len(self.tts_speech_token_dict[this_uuid])
                        ended = self.llm_end_dict[this_uuid]
                        toks = list(self.tts_speech_token_dict[this_uuid][: token_offset + this_token_hop_len + la])
                    if n_have - token_offset >= this_token_hop_len + la:
                        this_tts_speech = self.token2wav(token=torch.tensor(toks).unsqueeze(dim=0), prompt_token=flow_prompt_speech_token, prompt_feat=prompt_speech_feat,
                                                         embedding=flow_embedding, token_offset=token_offset, uuid=this_uuid, stream=stream, finalize=False)
                        token_offset += this_token_hop_len
                        token_hop_len = min(self.token_max_hop_len, token_hop_len * self.stream_scale_factor)
                        out = {"tts_speech": this_tts_speech.cpu()}
                        gate = self._first_gate.get(this_uuid)
                        if gate is not None:
                            gate.set()                         # first chunk is on the host: the LM carries on
                        yield out
                    elif ended:
                        break
                p.join()
                check_llm()
                this_tts_speech_token = torch.tensor(self.tts_speech_token_dict[this_uuid]).unsqueeze(dim=0)
                this_tts_speech = self.token2wav(token=this_tts_speech_token, prompt_token=flow_prompt_speech_token, prompt_feat=prompt_speech_feat,
                                                 embedding=flow_embedding, token_offset=token_offset, uuid=this_uuid, finalize=True)
                yield {"tts_speech": this_tts_speech.cpu()}
            else:
                p.join()
                check_llm()
                this_tts_speech_token = torch.tensor(self.tts_speech_token_dict[this_uuid]).unsqueeze(dim=0)
                this_tts_speech = self.token2wav(token=this_tts_speech_token, prompt_token=flow_prompt_speech_token, prompt_feat=prompt_speech_feat,
                                                 embedding=flow_embedding, token_offset=0, uuid=this_uuid, finalize=True, speed=speed)
                yield {"tts_speech": this_tts_speech.cpu()}
        finally:
            gate = self._first_gate.pop(this_uuid, None)
            if gate is not None:
                gate.set()
            p.join()
            with self.lock:
                self.tts_speech_token_dict.pop(this_uuid, None)
                self.llm_end_dict.pop(this_uuid, None)
                self.hift_cache_dict.pop(this_uuid, None)
                self._cond.pop(this_uuid, None)
                self._llm_error.pop(this_uuid, None)


class CosyVoice3Model(CosyVoice2Model):
    """Host mirror of cosyvoice.cli.model.CosyVoice3Model (cli/model.py:397-450; Fun-CosyVoice3): the tts() /
    llm_job / streaming loop of CosyVoice2Model over CosyVoice3LM + CausalMaskedDiffWithDiT + CausalHiFTGenerator, with the silent / breath
    token filter switched on (:423) and its own token2wav: the mel of a request accumulates in `hift_cache_dict[uuid]['mel']`, every call
    vocodes the whole mel so far with the causal generator (finalize flag = look-ahead handling) and returns the samples beyond
    `speech_offset` - no overlap cross-fade, no source cache.

    fp16=True: the reference runs flow AND vocoder of this model under `torch.cuda.amp.autocast(self.fp16)` (cli/model.py:426-447).  Here the flow's products go to
    the bf16 MFMA with fp32 accumulation (precision="bf16", as for CosyVoice2) and the vocoder's decoder convolutions keep 16 significand bits per factor with fp32
    accumulation (HiFT option "terms" = 3: more bits than autocast's fp16 operands, measured 96 dB against the fp32-exact class at full size, profiles/r6_hift.txt);
    the f0 predictor stays in double as in the reference (autocast leaves float64 operands alone, generator.py:716-717)."""

    def __init__(self, llm, flow, hift, fp16=False, lib=None):
        super().__init__(llm, flow, hift, fp16=fp16, lib=lib)
        self.silent_tokens = [1, 2, 28, 29, 55, 248, 494, 2241, 2242, 2322, 2323]

    @classmethod
    def from_state_dicts(cls, llm_sd, flow_sd, hift_sd, cfgs, lib=None, fp16=False, f0_float64=True, **llm_kw):
        """f0_float64: the vocoder's f0 predictor in double, the reference's mode (hifigan/generator.py:716-717) - the default since round 4; False = fp32 sums."""
        lc, fc, hc = cfgs
        lib = lib or get_lib()
        flow = CausalMaskedDiffWithDiT(flow_sd, fc, lib=lib, precision="bf16" if fp16 else "fp32")
        return cls(CosyVoice3LM(llm_sd, lc, lib=lib, **llm_kw), flow, CausalHiFTGenerator(hift_sd, hc, lib=lib, f0_float64=f0_float64, terms=3 if fp16 else 6), fp16=fp16)

    def load(self, llm_model, flow_model, hift_model, cfgs=None, f0_float64=True, **llm_kw):
        from .configs import cv3_flow, cv3_hift, cv3_llm
        lc, fc, hc = cfgs or (cv3_llm(), cv3_flow(), cv3_hift())
        llm_sd = torch.load(llm_model, map_location="cpu", weights_only=True)
        flow_sd = torch.load(flow_model, map_location="cpu", weights_only=True)
        hift_sd = {k.replace("generator.", ""): v for k, v in torch.load(hift_model, map_location="cpu", weights_only=True).items()}
        self.llm = CosyVoice3LM(llm_sd, lc, lib=self.lib, **llm_kw)
        self.flow = CausalMaskedDiffWithDiT(flow_sd, fc, lib=self.lib, precision="bf16" if self.fp16 else "fp32")
        self.hift = CausalHiFTGenerator(hift_sd, hc, lib=self.lib, f0_float64=f0_float64, terms=3 if self.fp16 else 6)
        self.set_lanes(max(1, self.n_lanes))
        self._warmup()

    @torch.inference_mode()
    def _vocode_mel(self, lane, mel, token):
        """cli/model.py:446-449 for a one-shot request (speech_offset 0)."""
        return lane.hift.inference(speech_feat=mel, finalize=True)[0]

    def token2wav(self, token, prompt_token, prompt_feat, embedding, token_offset, uuid, stream=False, finalize=False, speed=1.0):
        """cli/model.py:425-450."""
        with self._lane() as lane:
            t = lambda n: torch.tensor([n], dtype=torch.int32)
            tts_mel, _ = lane.flow.inference(token=token.to(torch.int32), token_len=t(token.shape[1]), prompt_token=prompt_token, prompt_token_len=t(prompt_token.shape[1]),
                                             prompt_feat=prompt_feat, prompt_feat_len=t(prompt_feat.shape[1]), embedding=embedding, streaming=stream, finalize=finalize)
            return self._t2w_tail(lane, tts_mel, token, token_offset, uuid, finalize, speed)

    @torch.inference_mode()
    def _t2w_tail(self, lane, tts_mel, token, token_offset, uuid, finalize, speed):
        """cli/model.py:432-450: the request's whole mel accumulates in the cache, the causal HiFT re-runs over it and the new samples are cut out."""
        flow, hift = lane.flow, lane.hift
        tts_mel = tts_mel[:, :, token_offset * flow.token_mel_ratio:]
        cache = self.hift_cache_dict.get(uuid)
        if cache is not None:
            tts_mel = torch.concat([cache["mel"], tts_mel], dim=2)
            cache["mel"] = tts_mel
        else:
            cache = self.hift_cache_dict[uuid] = {"mel": tts_mel, "speech_offset": 0}
        if speed != 1.0:
            assert token_offset == 0 and finalize is True, "speed change only support non-stream inference mode"
            tn = int(tts_mel.shape[2] / speed)
            src = tts_mel.contiguous()
            dst = torch.empty(1, src.shape[1], tn, dtype=torch.float32, device=self.device)
            self.lib.cv_interp_linear(C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), C.c_int32(src.shape[1]), C.c_int32(src.shape[2]), C.c_int32(tn), stream_ptr(self.lib))
            tts_mel = dst
        tts_speech, _ = hift.inference(speech_feat=tts_mel, finalize=finalize)
        tts_speech = tts_speech[:, cache["speech_offset"]:]
        cache["speech_offset"] += tts_speech.shape[1]
        return tts_speech
