"""Serving layer: a Triton-free batched streaming scheduler and the HTTP adapter of the reference runtime.

* `StreamScheduler` - what `runtime/triton_trtllm/model_repo/cosyvoice2/1/model.py:315-454` does with Triton + TensorRT-LLM: many concurrent
  streaming requests on ONE model object.  The speech-token LM runs continuous batching with streamed tokens on its own thread / HIP stream
  (`Qwen2LM.serve_stream`: up to 8 sequences per lock-step decode step, every weight matrix streamed once per step for all of them); a
  vocoder thread turns token prefixes into audio chunks (`CosyVoice2Model.token2wav`, per-uuid caches) under the reference's chunk rules:
  first chunk after `token_hop_len + prompt pad + pre_lookahead` tokens, then `exponential` (hop doubles up to `token_max_hop_len`,
  cli/model.py:345-360 - the default) or `time_based` (model.py:410-426) hop growth.  Every request's audio equals `tts(stream=True)` of
  that request alone under `exponential` (chunk boundaries depend only on token counts).
* `create_app(engine)` - the FastAPI surface of `runtime/python/fastapi/server.py:46-86` (same routes, int16 PCM `StreamingResponse`)
  over any engine exposing the `CosyVoice2.inference_*` generators; `create_grpc_server(engine)` - the `CosyVoice.Inference` service of
  `runtime/python/grpc/server.py:34-77`, wire-compatible with the reference's cosyvoice.proto.  Text normalisation / tokenisation and the ONNX extractors (speech
  tokenizer, CAM++) are the reference front end's job (`cosyvoice/cli/frontend.py`, out of scope here - SURVEY.md section 8f item 2): `Engine`
  takes such a front end as an object and only replaces its mel extractor and the model underneath.
"""
import collections
import os
import queue
import threading
import time
import uuid as uuid_mod

import numpy as np
import torch


class _Request:
    __slots__ = ("key", "req", "tokens", "llm_done", "error", "out", "token_offset", "hop", "chunk_index", "t_submit", "t_first", "stream", "pad", "closed", "busy", "t_ready", "t_pick",
                 "filt")

    def __init__(self, key, req, stream, hop, pad, silent_tokens=()):
        from .model import SilentTokenFilter
        self.key, self.req, self.stream = key, req, stream
        self.filt = SilentTokenFilter(silent_tokens)     # llm_job's silent / breath-token rule, run count carried over the decode chunks
        self.tokens, self.llm_done, self.error = [], False, None
        self.out = queue.Queue()
        self.token_offset, self.hop, self.chunk_index, self.pad = 0, hop, 0, pad
        self.t_submit, self.t_first, self.closed, self.busy = time.perf_counter(), None, False, False
        self.t_ready = self.t_pick = None


class StreamScheduler:
    def __init__(self, model, slots=8, strategy="exponential", step_chunk=8, chunk_batch=None):
        assert strategy in ("exponential", "time_based")
        self.model, self.slots, self.strategy, self.step_chunk = model, slots, strategy, step_chunk
        # up to this many ready requests share one flow pass (_mates); None = the model's offline flow_batch when it can batch, 1 = one by one (round 2)
        if chunk_batch is None and os.environ.get("CV_CHUNK_BATCH"):                       # dev knob for A/B runs
            chunk_batch = int(os.environ["CV_CHUNK_BATCH"])
        self.chunk_batch = (getattr(model, "flow_batch", 1) if hasattr(model, "token2wav_batch") else 1) if chunk_batch is None else chunk_batch
        self.chunk_batch = max(1, min(8, int(self.chunk_batch)))       # a flow pass takes at most 8 utterances (cv_flow_inference_batch / _ragged)
        self.batched_passes = self.batched_jobs = 0     # passes that carried more than one request, and the requests in them
        # many concurrent streams: their chunk passes replay captured graphs (a request served alone is faster without them: model.set_flow_graph_rows)
        self._graph_rows_before = getattr(model, "flow_graph_rows", None)
        if hasattr(model, "set_flow_graph_rows"):
            model.set_flow_graph_rows(int(os.environ.get("CV_SERVE_GRAPH_ROWS", 3000)))      # (the variable: A/B knob)
        self._src = queue.Queue()
        self._cv = threading.Condition()
        self._reqs = {}
        self._stop = False
        self._dead = None                               # exception that ended the LM thread, if any (submit() re-raises it)
        self.token_log = None                            # optional dict key -> every token the LM delivered for the request (bench.py's self-check sets it)
        # (LM ms until the first chunk's tokens exist, ms waiting for a vocoder lane, ms of the first token2wav) of the last requests
        self.first_chunk_stats = collections.deque(maxlen=4096)
        self._llm_thread = threading.Thread(target=self._llm_loop, daemon=True)
        # one vocoder thread per token2wav lane of the model (CosyVoice2Model.set_lanes): chunks of DIFFERENT requests are vocoded concurrently
        # on different HIP streams, the chunks of one request stay sequential (`busy`).  With >= 3 lanes the first thread takes FIRST chunks only:
        # they are short (T = 2 * (prompt + 38 tokens)) and are what a listener waits for, so they never queue behind a 40 ms late-chunk flow.
        n_voc = max(1, getattr(model, "n_lanes", 1))
        self._voc_threads = [threading.Thread(target=self._vocoder_loop, args=(n_voc >= 3 and i == 0,), daemon=True) for i in range(n_voc)]
        self._llm_thread.start()
        for t in self._voc_threads:
            t.start()

    # ---- client side ---------------------------------------------------------------------------------------------------------
    def submit(self, stream=True, **req):
        """req: the keyword tensors of CosyVoice2Model.tts (text, prompt_text, llm_prompt_speech_token, flow_prompt_speech_token,
        prompt_speech_feat, llm_embedding, flow_embedding [, min/max_token_text_ratio]).  Returns a generator of {'tts_speech': [1, S] cpu}."""
        m = self.model
        key = str(uuid_mod.uuid1())
        n_prompt = int(req["flow_prompt_speech_token"].shape[1])
        pad = int(np.ceil(n_prompt / m.token_hop_len) * m.token_hop_len - n_prompt)
        r = _Request(key, req, stream, m.token_hop_len, pad, getattr(m, "silent_tokens", ()))
        with self._cv:
            if self._stop:
                raise RuntimeError("scheduler is shut down")
            if self._dead is not None:                    # the LM thread is gone: refuse instead of queueing a request nobody will serve
                raise RuntimeError("scheduler: the LM thread died: %r" % (self._dead,)) from self._dead
            self._reqs[key] = r
        with m.lock:
            m.hift_cache_dict[key] = None
        lm_req = dict(text=req["text"], prompt_text=req["prompt_text"], prompt_speech_token=req["llm_prompt_speech_token"],
                      **{k: req[k] for k in ("min_token_text_ratio", "max_token_text_ratio") if k in req})
        if stream:
            lm_req["first_chunk"] = m.token_hop_len + pad + m.flow.pre_lookahead_len
        self._src.put((key, lm_req))
        return self._drain(r)

    def _drain(self, r):
        try:
            while True:
                item = r.out.get()
                if item is None:
                    return
                if isinstance(item, BaseException):
                    raise item
                yield item
        finally:
            with self._cv:
                r.closed = True                           # a client that stopped listening: _ready() turns the request into a "cancel"
                self._cv.notify_all()

    def first_chunk_latency(self, key_request):
        return None if key_request.t_first is None else key_request.t_first - key_request.t_submit

    def shutdown(self):
        with self._cv:
            self._stop = True
            self._cv.notify_all()
        self._src.put(None)
        self._llm_thread.join()
        for t in self._voc_threads:
            t.join()
        if self._graph_rows_before is not None and hasattr(self.model, "set_flow_graph_rows"):
            self.model.set_flow_graph_rows(self._graph_rows_before)        # the model goes back to its own policy (requests served one at a time)
            self._graph_rows_before = None

    # ---- LLM thread: continuous batching, tokens streamed per decode chunk ------------------------------------------------------------
    def _on_tokens(self, key, toks, finished, error):
        with self._cv:
            if self.token_log is not None:
                self.token_log.setdefault(key, []).extend(int(t) for t in toks)
            r = self._reqs.get(key)
            if r is None or r.closed:
                return True                              # unknown / abandoned request: serve_stream frees its slot
            r.tokens.extend(r.filt(toks))
            if error is not None:
                r.error = error
            if finished:
                r.llm_done = True
            if r.t_ready is None and r.token_offset == 0 and self._ready(r) is not None:
                r.t_ready = time.perf_counter()
            self._cv.notify_all()

    def _llm_loop(self):
        m = self.model
        try:
            with m.llm_context:
                m.llm.serve_stream(self._src, self._on_tokens, slots=self.slots, step_chunk=self.step_chunk)
        except BaseException as e:                   # the LM thread died: fail every open request instead of hanging its client
            with self._cv:
                self._dead = e
                for r in self._reqs.values():
                    r.error, r.llm_done = r.error or e, True
                self._cv.notify_all()

    # ---- vocoder thread: chunk rules of cli/model.py:341-371 per request, first-come first-served over ready requests -----------------
    def _ready(self, r):
        la = self.model.flow.pre_lookahead_len
        if r.closed:
            return "cancel"                               # the client closed its generator: drop the request, free its lane and caches
        if r.error is not None:
            return "error"
        hop = r.hop + r.pad if r.token_offset == 0 else r.hop
        if r.stream and len(r.tokens) - r.token_offset >= hop + la:
            return "chunk"
        if r.llm_done:
            return "final"
        return None

    def _job_tokens(self, r, what):
        """Token count (prompt included) of the flow pass the ready work of `r` needs - what decides which requests may share a padded pass."""
        la = self.model.flow.pre_lookahead_len
        n_prompt = int(r.req["flow_prompt_speech_token"].shape[1])
        if what == "chunk":
            return n_prompt + r.token_offset + (r.hop + r.pad if r.token_offset == 0 else r.hop) + la
        return n_prompt + len(r.tokens)

    def _mates(self, pick, first_only):
        """Called under the lock with `pick` = (request, 'chunk' | 'final', tokens) already chosen: other ready, idle requests whose work is of the
        SAME kind and whose flow pass is of similar length (the model's offline grouping rule: within 1 / flow_pad of the longest, at most
        chunk_batch members) - they go through the flow together (CosyVoice2Model.token2wav_batch: one Euler solve over all of them, every mel
        bit-identical to the request alone), so a loaded GPU runs one pass of 2-4 x the rows instead of 2-4 passes side by side."""
        r0, what, _ = pick
        if what == "final" and not r0.stream and r0.req.get("speed", 1.0) != 1.0:
            return []
        n0, pad = self._job_tokens(r0, what), getattr(self.model, "flow_pad", 1.25)
        out = []
        for r in self._reqs.values():
            if len(out) + 1 >= self.chunk_batch:
                break
            if r is r0 or r.busy or (first_only and r.chunk_index > 0) or self._ready(r) != what:
                continue
            if what == "final" and not r.stream and r.req.get("speed", 1.0) != 1.0:
                continue
            n = self._job_tokens(r, what)
            if min(n, n0) * pad < max(n, n0):
                continue
            r.busy = True
            if r.t_pick is None:
                r.t_pick = time.perf_counter()
            out.append((r, what, list(r.tokens)))
        return out

    def _vocode_together(self, picks):
        """The 'chunk' / 'final' branch of _vocoder_loop for several requests in one token2wav_batch call; per-request bookkeeping as there."""
        m = self.model
        la = m.flow.pre_lookahead_len
        what = picks[0][1]
        jobs = []
        for r, _, toks in picks:
            rq = r.req
            if what == "chunk":
                hop = r.hop + r.pad if r.token_offset == 0 else r.hop
                toks = toks[:r.token_offset + hop + la]
            jobs.append(dict(token=torch.tensor(toks).unsqueeze(0), prompt_token=rq["flow_prompt_speech_token"], prompt_feat=rq["prompt_speech_feat"],
                             embedding=rq["flow_embedding"], token_offset=r.token_offset, uuid=r.key))
        delivered = set()

        def deliver(i, wav):                     # called inside token2wav_batch right after request i's HiFT: its listener does not wait for the others
            r = picks[i][0]
            out = {"tts_speech": wav.cpu()}
            if what == "chunk":
                hop = r.hop + r.pad if r.token_offset == 0 else r.hop
                r.token_offset += hop
                r.chunk_index += 1
                self._next_hop(r)
            if r.t_first is None:
                r.t_first = time.perf_counter()
                if what == "chunk" and r.t_ready is not None:
                    self.first_chunk_stats.append(((r.t_ready - r.t_submit) * 1e3, (r.t_pick - r.t_ready) * 1e3, (r.t_first - r.t_pick) * 1e3))
            r.out.put(out)
            if what == "final":
                r.out.put(None)
            delivered.add(i)
        failed = {}
        # A member's vocoder caches (mel / source / speech tail) advance inside token2wav_batch BEFORE its listener runs; if the pass - or deliver() itself - fails after
        # that, a solo retry must start from the caches the chunk started from, or it would vocode the chunk a second time against already-advanced state.
        # SHALLOW COPIES, not references: CosyVoice3Model._t2w_tail updates its cache dict in place (cache['mel'] = concat(...), cache['speech_offset'] += ...), so a kept
        # reference would show the advanced state and the restore below would be a no-op; the tensors themselves are never written in place.
        snap = lambda c: dict(c) if isinstance(c, dict) else c
        with m.lock:
            before = {i: (r.key in m.hift_cache_dict, snap(m.hift_cache_dict.get(r.key))) for i, (r, _, _) in enumerate(picks)}
        try:
            m.token2wav_batch(jobs, stream=(what == "chunk"), finalize=(what == "final"), on_ready=deliver)
        except Exception:
            # The shared pass (or one member's vocoder call) failed.  A request must not pay for a neighbour it happened to share a pass with:
            # members that already got this chunk carry on untouched; the others are vocoded again ONE BY ONE from the cache state the chunk started from, and
            # only those that fail alone end with their error.
            for i, (r, _, _) in enumerate(picks):
                if i in delivered:
                    continue
                try:
                    with m.lock:
                        if before[i][0] and r.key in m.hift_cache_dict:     # (a request popped in the meantime - ended, cancelled - is not given its key back)
                            m.hift_cache_dict[r.key] = snap(before[i][1])
                    deliver(i, m.token2wav(stream=(what == "chunk"), finalize=(what == "final"), **jobs[i]))
                except Exception as e:                      # handed to the request's listener
                    failed[i] = e
                    r.out.put(e)
        self.batched_passes += 1
        self.batched_jobs += len(picks)
        gone = [r for i, (r, _, _) in enumerate(picks) if what == "final" or i in failed]
        with self._cv:
            for r, _, _ in picks:
                r.busy = False
            for r in gone:
                self._reqs.pop(r.key, None)
            self._cv.notify_all()
        if gone:
            with m.lock:
                for r in gone:
                    m.hift_cache_dict.pop(r.key, None)

    def _next_hop(self, r):
        """Hop growth after a chunk (cli/model.py:360 `exponential`; triton model.py:410-426 `time_based`)."""
        m = self.model
        if self.strategy == "exponential":
            r.hop = min(m.token_max_hop_len, r.hop * m.stream_scale_factor)
        else:                                   # time_based: grow the hop while synthesis runs ahead of playback
            cost, dur = time.perf_counter() - r.t_submit, r.token_offset / 25.0
            mult = (dur - cost) / max(cost / r.chunk_index, 1e-6)
            pend = len(r.tokens) - r.token_offset
            base = m.token_hop_len
            r.hop = max(base, (pend // base + 1) * base if mult > 4 else (pend // base) * base if mult > 2 else base)

    def _vocoder_loop(self, first_only=False):
        m = self.model
        la = m.flow.pre_lookahead_len
        while True:
            with self._cv:
                while True:
                    # Ready work, lowest chunk index first (ties: submission order): a request's FIRST chunk is what its listener is waiting
                    # for (first-chunk latency), while its later chunks only have to arrive before the audio already delivered runs out.
                    pick, best = None, None
                    for r in self._reqs.values():
                        what = None if (r.busy or (first_only and r.chunk_index > 0)) else self._ready(r)
                        if what is not None and (best is None or r.chunk_index < best):
                            pick, best = (r, what, list(r.tokens)), r.chunk_index
                            if best == 0:
                                break
                    if pick is not None or (self._stop and not self._reqs):
                        break
                    self._cv.wait(timeout=0.5)
                if pick is None:
                    return
                pick[0].busy = True
                if pick[0].t_pick is None:
                    pick[0].t_pick = time.perf_counter()
                mates = self._mates(pick, first_only) if self.chunk_batch > 1 and pick[1] in ("chunk", "final") else []
            if mates:
                self._vocode_together([pick] + mates)
                continue
            r, what, toks = pick
            rq = r.req
            finished = True
            try:
                if what == "cancel":
                    pass
                elif what == "error":
                    raise r.error
                elif what == "chunk":
                    hop = r.hop + r.pad if r.token_offset == 0 else r.hop
                    n = r.token_offset + hop + la
                    wav = m.token2wav(token=torch.tensor(toks[:n]).unsqueeze(0), prompt_token=rq["flow_prompt_speech_token"], prompt_feat=rq["prompt_speech_feat"],
                                      embedding=rq["flow_embedding"], token_offset=r.token_offset, uuid=r.key, stream=True, finalize=False)
                    r.token_offset += hop
                    r.chunk_index += 1
                    self._next_hop(r)
                    out = {"tts_speech": wav.cpu()}
                    if r.t_first is None:
                        r.t_first = time.perf_counter()
                        if r.t_ready is not None:
                            self.first_chunk_stats.append(((r.t_ready - r.t_submit) * 1e3, (r.t_pick - r.t_ready) * 1e3, (r.t_first - r.t_pick) * 1e3))
                    r.out.put(out)
                    finished = False
                else:
                    wav = m.token2wav(token=torch.tensor(toks).unsqueeze(0), prompt_token=rq["flow_prompt_speech_token"], prompt_feat=rq["prompt_speech_feat"],
                                      embedding=rq["flow_embedding"], token_offset=r.token_offset, uuid=r.key, finalize=True,
                                      speed=1.0 if r.stream else rq.get("speed", 1.0))
                    out = {"tts_speech": wav.cpu()}
                    if r.t_first is None:
                        r.t_first = time.perf_counter()
                    r.out.put(out)
                    r.out.put(None)
            except BaseException as e:
                r.out.put(e)
            with self._cv:
                r.busy = False
                if finished:                                # done or failed: drop the per-request state (cli/model.py:388-391)
                    self._reqs.pop(r.key, None)
                self._cv.notify_all()
            if finished:
                with m.lock:
                    m.hift_cache_dict.pop(r.key, None)


# ---------------------------------------------------------------------------------------------------------------------------------
# HTTP adapter (runtime/python/fastapi/server.py)
# ---------------------------------------------------------------------------------------------------------------------------------
def pcm16_stream(model_output):
    """server.py:38-41: every yielded chunk as little-endian int16 PCM bytes."""
    for i in model_output:
        yield (i["tts_speech"].numpy() * (2 ** 15)).astype(np.int16).tobytes()


class Engine:
    """`CosyVoice2`-shaped facade (cosyvoice/cli/cosyvoice.py:166-226) over a StreamScheduler: `frontend` is a CosyVoiceFrontEnd-like object
    whose `frontend_zero_shot / frontend_cross_lingual / frontend_instruct2 / frontend_sft` return the model-input dicts of the reference
    (cli/frontend.py:157-222) and whose `text_normalize(text, split=True)` splits text into segments."""

    def __init__(self, scheduler, frontend, sample_rate=24000):
        self.scheduler, self.frontend, self.sample_rate = scheduler, frontend, sample_rate

    def _run(self, segments, make_input, stream, speed):
        for seg in segments:
            mi = make_input(seg)
            req = {k: mi[k] for k in ("text", "prompt_text", "llm_prompt_speech_token", "flow_prompt_speech_token", "prompt_speech_feat", "llm_embedding", "flow_embedding") if k in mi}
            req.setdefault("prompt_text", torch.zeros(1, 0, dtype=torch.int32))
            req.setdefault("llm_prompt_speech_token", torch.zeros(1, 0, dtype=torch.int32))
            req.setdefault("flow_prompt_speech_token", torch.zeros(1, 0, dtype=torch.int32))
            req.setdefault("prompt_speech_feat", torch.zeros(1, 0, 80))
            yield from self.scheduler.submit(stream=stream, speed=speed, **req)

    def inference_sft(self, tts_text, spk_id, stream=False, speed=1.0):
        yield from self._run(self.frontend.text_normalize(tts_text, split=True), lambda s: self.frontend.frontend_sft(s, spk_id), stream, speed)

    def inference_zero_shot(self, tts_text, prompt_text, prompt_wav, zero_shot_spk_id="", stream=False, speed=1.0):
        yield from self._run(self.frontend.text_normalize(tts_text, split=True),
                             lambda s: self.frontend.frontend_zero_shot(s, prompt_text, prompt_wav, self.sample_rate, zero_shot_spk_id), stream, speed)

    def inference_cross_lingual(self, tts_text, prompt_wav, zero_shot_spk_id="", stream=False, speed=1.0):
        yield from self._run(self.frontend.text_normalize(tts_text, split=True),
                             lambda s: self.frontend.frontend_cross_lingual(s, prompt_wav, self.sample_rate, zero_shot_spk_id), stream, speed)

    def inference_instruct2(self, tts_text, instruct_text, prompt_wav, zero_shot_spk_id="", stream=False, speed=1.0):
        yield from self._run(self.frontend.text_normalize(tts_text, split=True),
                             lambda s: self.frontend.frontend_instruct2(s, instruct_text, prompt_wav, self.sample_rate, zero_shot_spk_id), stream, speed)


def create_app(engine, load_wav=None):
    """FastAPI app with the routes of runtime/python/fastapi/server.py:46-86.  `load_wav(file, sr)` decodes an uploaded prompt (the reference
    uses cosyvoice.utils.file_utils.load_wav -> torchaudio, not available offline); the default reads 16-bit PCM WAV with the stdlib."""
    from fastapi import FastAPI, File, Form, UploadFile
    from fastapi.middleware.cors import CORSMiddleware
    from fastapi.responses import StreamingResponse

    def default_load_wav(f, sr):
        import wave
        with wave.open(f, "rb") as w:
            assert w.getsampwidth() == 2, "16-bit PCM WAV expected"
            data = np.frombuffer(w.readframes(w.getnframes()), dtype=np.int16).astype(np.float32) / 32768.0
            if w.getnchannels() > 1:
                data = data.reshape(-1, w.getnchannels()).mean(1)
            assert w.getframerate() == sr, "prompt must be sampled at %d Hz (resampling is the front end's job)" % sr
        return torch.from_numpy(data).unsqueeze(0)

    load = load_wav or default_load_wav
    app = FastAPI()
    app.add_middleware(CORSMiddleware, allow_origins=["*"], allow_credentials=True, allow_methods=["*"], allow_headers=["*"])
    try:
        import multipart  # noqa: F401  (python-multipart: FastAPI needs it for Form / File parameters)
        have_multipart = True
    except ImportError:
        have_multipart = False

    if have_multipart:                                          # the reference's exact signatures (multipart forms)
        @app.get("/inference_sft")
        @app.post("/inference_sft")
        async def inference_sft(tts_text: str = Form(), spk_id: str = Form()):
            return StreamingResponse(pcm16_stream(engine.inference_sft(tts_text, spk_id)))

        @app.get("/inference_zero_shot")
        @app.post("/inference_zero_shot")
        async def inference_zero_shot(tts_text: str = Form(), prompt_text: str = Form(), prompt_wav: UploadFile = File()):
            return StreamingResponse(pcm16_stream(engine.inference_zero_shot(tts_text, prompt_text, load(prompt_wav.file, 16000))))

        @app.get("/inference_cross_lingual")
        @app.post("/inference_cross_lingual")
        async def inference_cross_lingual(tts_text: str = Form(), prompt_wav: UploadFile = File()):
            return StreamingResponse(pcm16_stream(engine.inference_cross_lingual(tts_text, load(prompt_wav.file, 16000))))

        @app.get("/inference_instruct2")
        @app.post("/inference_instruct2")
        async def inference_instruct2(tts_text: str = Form(), instruct_text: str = Form(), prompt_wav: UploadFile = File()):
            return StreamingResponse(pcm16_stream(engine.inference_instruct2(tts_text, instruct_text, load(prompt_wav.file, 16000))))
        return app

    # python-multipart is not installed: same routes, text fields as query parameters, the prompt WAV as the raw request body
    import io
    from fastapi import Request

    @app.get("/inference_sft")
    @app.post("/inference_sft")
    async def inference_sft_q(tts_text: str, spk_id: str):
        return StreamingResponse(pcm16_stream(engine.inference_sft(tts_text, spk_id)))

    @app.post("/inference_zero_shot")
    async def inference_zero_shot_q(request: Request, tts_text: str, prompt_text: str):
        wav = load(io.BytesIO(await request.body()), 16000)
        return StreamingResponse(pcm16_stream(engine.inference_zero_shot(tts_text, prompt_text, wav)))

    @app.post("/inference_cross_lingual")
    async def inference_cross_lingual_q(request: Request, tts_text: str):
        wav = load(io.BytesIO(await request.body()), 16000)
        return StreamingResponse(pcm16_stream(engine.inference_cross_lingual(tts_text, wav)))

    @app.post("/inference_instruct2")
    async def inference_instruct2_q(request: Request, tts_text: str, instruct_text: str):
        wav = load(io.BytesIO(await request.body()), 16000)
        return StreamingResponse(pcm16_stream(engine.inference_instruct2(tts_text, instruct_text, wav)))

    return app


# ---------------------------------------------------------------------------------------------------------------------------------
# gRPC adapter (runtime/python/grpc/server.py + cosyvoice.proto)
# ---------------------------------------------------------------------------------------------------------------------------------
_GRPC_MESSAGES = None


def grpc_messages():
    """(Request, Response) message classes of runtime/python/grpc/cosyvoice.proto (package `cosyvoice`: Request = oneof {sft_request = 1,
    zero_shot_request = 2, cross_lingual_request = 3, instruct_request = 4}, Response{bytes tts_audio = 1}), built from a descriptor at import
    time - the image has grpcio and protobuf but no protoc plugin.  Same field names and numbers, so the reference's client.py talks to it."""
    global _GRPC_MESSAGES
    if _GRPC_MESSAGES is not None:
        return _GRPC_MESSAGES
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    f = descriptor_pb2.FileDescriptorProto(name="cosyvoice_amd/cosyvoice.proto", package="cosyvoice", syntax="proto3")
    STR, BYTES, MSG = (descriptor_pb2.FieldDescriptorProto.TYPE_STRING, descriptor_pb2.FieldDescriptorProto.TYPE_BYTES,
                       descriptor_pb2.FieldDescriptorProto.TYPE_MESSAGE)

    def message(name, fields):
        m = f.message_type.add(name=name)
        for i, (fname, ftype) in enumerate(fields, 1):
            m.field.add(name=fname, number=i, type=ftype, label=descriptor_pb2.FieldDescriptorProto.LABEL_OPTIONAL)
        return m
    message("sftRequest", [("spk_id", STR), ("tts_text", STR)])
    message("zeroshotRequest", [("tts_text", STR), ("prompt_text", STR), ("prompt_audio", BYTES)])
    message("crosslingualRequest", [("tts_text", STR), ("prompt_audio", BYTES)])
    message("instructRequest", [("tts_text", STR), ("spk_id", STR), ("instruct_text", STR)])
    message("Response", [("tts_audio", BYTES)])
    req = f.message_type.add(name="Request")
    req.oneof_decl.add(name="RequestPayload")
    for i, (fname, tname) in enumerate([("sft_request", "sftRequest"), ("zero_shot_request", "zeroshotRequest"),
                                        ("cross_lingual_request", "crosslingualRequest"), ("instruct_request", "instructRequest")], 1):
        req.field.add(name=fname, number=i, type=MSG, type_name=".cosyvoice." + tname, label=descriptor_pb2.FieldDescriptorProto.LABEL_OPTIONAL, oneof_index=0)
    pool = descriptor_pool.DescriptorPool()
    pool.Add(f)
    get = lambda n: message_factory.GetMessageClass(pool.FindMessageTypeByName("cosyvoice." + n))
    _GRPC_MESSAGES = (get("Request"), get("Response"))
    return _GRPC_MESSAGES


GRPC_METHOD = "/cosyvoice.CosyVoice/Inference"


def _pcm16_prompt(raw):
    """server.py:47-48: the prompt travels as raw little-endian int16 samples at 16 kHz."""
    return torch.from_numpy(np.array(np.frombuffer(raw, dtype=np.int16))).unsqueeze(dim=0).float() / (2 ** 15)


def create_grpc_server(engine, port=50000, max_conc=4, host="0.0.0.0"):
    """The `CosyVoice.Inference` service of runtime/python/grpc/server.py:34-77 (one request message, a stream of int16 PCM responses) over any
    engine exposing the `inference_*` generators.  Returns (grpc.Server - not started -, bound port)."""
    from concurrent import futures
    import grpc
    Request, Response = grpc_messages()

    def inference(request, context):
        kind = request.WhichOneof("RequestPayload")
        if kind == "sft_request":
            out = engine.inference_sft(request.sft_request.tts_text, request.sft_request.spk_id)
        elif kind == "zero_shot_request":
            r = request.zero_shot_request
            out = engine.inference_zero_shot(r.tts_text, r.prompt_text, _pcm16_prompt(r.prompt_audio))
        elif kind == "cross_lingual_request":
            r = request.cross_lingual_request
            out = engine.inference_cross_lingual(r.tts_text, _pcm16_prompt(r.prompt_audio))
        elif kind == "instruct_request" and hasattr(engine, "inference_instruct"):
            r = request.instruct_request
            out = engine.inference_instruct(r.tts_text, r.spk_id, r.instruct_text)
        else:       # CosyVoice2 / 3 engines have inference_instruct2 (needs a prompt wav, which this message does not carry): say so instead of guessing
            context.abort(grpc.StatusCode.UNIMPLEMENTED, "this engine has no inference_instruct (instructRequest carries no prompt audio)" if kind else "empty request")
            return
        for chunk in pcm16_stream(out):
            yield Response(tts_audio=chunk)
    handler = grpc.method_handlers_generic_handler("cosyvoice.CosyVoice", {"Inference": grpc.unary_stream_rpc_method_handler(
        inference, request_deserializer=Request.FromString, response_serializer=Response.SerializeToString)})
    server = grpc.server(futures.ThreadPoolExecutor(max_workers=max_conc), maximum_concurrent_rpcs=max_conc)
    server.add_generic_rpc_handlers((handler,))
    bound = server.add_insecure_port("%s:%d" % (host, port))
    return server, bound
