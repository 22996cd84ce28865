"""the batched streaming scheduler (Triton-free equivalent of runtime/triton_trtllm/model_repo/cosyvoice2/1/model.py:
315-454) and the FastAPI adapter (runtime/python/fastapi/server.py:46-86)."""
import dataclasses
import io
import threading
import time
import wave

import numpy as np
import pytest
import torch

from cosyvoice_amd.model import CosyVoice2Model
from cosyvoice_amd.serving import Engine, StreamScheduler, create_app, pcm16_stream
from cosyvoice_amd import synthetic as W


@pytest.fixture(scope="module")
def setup():
    lc, fc, hc = W.tiny()
    fc = dataclasses.replace(fc, chunk=5, n_timesteps=2)
    return (lc, fc, hc), (W.make_llm(lc), W.make_flow(fc), W.make_hift(hc))


def _model(lib, cfgs, sds):
    m = CosyVoice2Model.from_state_dicts(*sds, cfgs, lib=lib, max_len=160, sampling="greedy", decode_chunk=8)
    m.token_hop_len, m.token_max_hop_len = 5, 20
    inf = m.hift.inference
    m.hift.inference = lambda speech_feat, cache_source=None: inf(speech_feat, cache_source, noise=torch.zeros(speech_feat.shape[2] * 480, 9))
    return m


def _req(cfgs, seed, n_text, n_prompt):
    u = W.synthetic_utterance(cfgs[0], cfgs[1], n_prompt_tok=n_prompt, n_prompt_text=3, n_text=n_text, seed=seed)
    return {k: u[k] for k in ("text", "prompt_text", "llm_prompt_speech_token", "flow_prompt_speech_token", "prompt_speech_feat", "llm_embedding", "flow_embedding")}


def test_scheduler_streams_equal_single_requests(lib, setup):
    """Concurrent streaming requests through ONE scheduler (LM continuous batching with streamed tokens + chunked token2wav) produce,
    chunk for chunk, what CosyVoice2Model.tts(stream=True) produces for each request alone; offline requests likewise."""
    if lib.emulated:
        pytest.skip("hardware only: ~6 min under the CPU emulator (the scheduling logic is covered by test_scheduler_logic_with_fake_model, "
                    "the kernels it drives by tests/test_zz_llm_batch.py and tests/test_model.py)")
    cfgs, sds = setup
    m = _model(lib, cfgs, sds)
    reqs = [_req(cfgs, 41, 2, 8), _req(cfgs, 42, 1, 6)]
    alone = [[o["tts_speech"] for o in m.tts(stream=True, **r)] for r in reqs]
    alone_off = [o["tts_speech"] for o in m.tts(stream=False, **reqs[0])]
    sch = StreamScheduler(m, slots=4, step_chunk=4)
    try:
        got, errs = [None] * 3, []

        def run(i, r, stream):
            try:
                got[i] = [o["tts_speech"] for o in sch.submit(stream=stream, **r)]
            except Exception as e:          # pragma: no cover
                errs.append(repr(e))
        th = [threading.Thread(target=run, args=(i, r, True)) for i, r in enumerate(reqs)] + [threading.Thread(target=run, args=(2, reqs[0], False))]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errs, errs
        for a, b in zip(got[:2], alone):
            assert len(a) == len(b)
            for x, y in zip(a, b):
                assert torch.equal(x, y)
        assert len(got[0]) >= 2                                                  # the first request really streamed in several chunks
        assert len(got[2]) == 1 and torch.equal(got[2][0], alone_off[0])
        assert not m.hift_cache_dict and not sch._reqs
        # a bad request fails on its own channel and the server keeps serving
        bad = dict(reqs[1]); bad["text"] = torch.zeros(1, 200, dtype=torch.int32)      # prompt + min_len cannot fit the KV capacity of 160
        with pytest.raises(ValueError):
            list(sch.submit(stream=True, **bad))
        again = [o["tts_speech"] for o in sch.submit(stream=False, **reqs[0])]
        assert len(again) == 1 and torch.equal(again[0], alone_off[0])
    finally:
        sch.shutdown()


class _StubFrontend:
    """Stands in for cosyvoice.cli.frontend.CosyVoiceFrontEnd (tokenizer + ONNX extractors, out of scope): maps text to ids by length."""

    def __init__(self, cfgs):
        self.cfgs = cfgs

    def text_normalize(self, text, split=True):
        return [t for t in text.split("|") if t]

    def frontend_zero_shot(self, tts_text, prompt_text, prompt_wav, resample_rate, zero_shot_spk_id):
        assert prompt_wav.dim() == 2 and resample_rate == 24000
        return _req(self.cfgs, 50 + len(tts_text), max(1, len(tts_text) % 3), 6)

    def frontend_cross_lingual(self, tts_text, prompt_wav, resample_rate, zero_shot_spk_id):
        r = self.frontend_zero_shot(tts_text, "", prompt_wav, resample_rate, zero_shot_spk_id)
        del r["prompt_text"], r["llm_prompt_speech_token"]
        return r


class _FakeModel:
    """Host-logic double of CosyVoice2Model for the scheduler: the LM emits scripted tokens in chunks, token2wav returns a waveform that
    encodes which tokens it was asked to vocode."""

    class _Flow:
        pre_lookahead_len = 3

    class _LLM:
        def __init__(self, scripts):
            self.scripts = scripts

        def serve_stream(self, source, on_tokens, slots=8, step_chunk=8, **kw):
            import queue as _q
            active, closed = {}, False
            while True:
                while not closed:
                    try:
                        item = source.get(block=not active, timeout=None if not active else 0)
                    except _q.Empty:
                        break
                    if item is None:
                        closed = True
                        break
                    key, r = item
                    n = int(r["text"].shape[1])
                    if n == 0:
                        on_tokens(key, [], True, ValueError("empty text"))
                    else:
                        active[key] = list(self.scripts[n])
                if not active:
                    if closed:
                        return
                    continue
                for key in list(active):
                    toks, active[key] = active[key][:step_chunk], active[key][step_chunk:]
                    on_tokens(key, toks, not active[key], None)
                    if not active[key]:
                        del active[key]

    def __init__(self, scripts):
        import contextlib
        self.llm, self.flow = self._LLM(scripts), self._Flow()
        self.llm_context = contextlib.nullcontext()
        self.lock = threading.Lock()
        self.hift_cache_dict = {}
        self.token_hop_len, self.token_max_hop_len, self.stream_scale_factor = 5, 20, 2
        self.calls = []

    def token2wav(self, token, prompt_token, prompt_feat, embedding, token_offset, uuid, stream=False, finalize=False, speed=1.0):
        self.calls.append((uuid, token.shape[1], token_offset, stream, finalize))
        n_new = token.shape[1] - token_offset - (0 if finalize else 3)
        return torch.full((1, n_new * 960), float(token_offset))


@pytest.mark.timeout(120)
def test_scheduler_logic_with_fake_model():
    """Chunk rules of cli/model.py:341-371 under the scheduler: first chunk after hop + prompt pad + lookahead tokens, hop doubling up to the
    maximum, final call with everything; offline requests vocode once; a failing request reports on its own channel; state is cleaned up."""
    scripts = {1: list(range(47)), 2: list(range(9))}
    fm = _FakeModel(scripts)
    sch = StreamScheduler(fm, slots=4, step_chunk=4)
    try:
        def req(n_text, n_prompt=8):
            return dict(text=torch.zeros(1, n_text, dtype=torch.int32), prompt_text=torch.zeros(1, 2, dtype=torch.int32),
                        llm_prompt_speech_token=torch.zeros(1, n_prompt, dtype=torch.int32), flow_prompt_speech_token=torch.zeros(1, n_prompt, dtype=torch.int32),
                        prompt_speech_feat=torch.zeros(1, 2 * n_prompt, 80), llm_embedding=torch.zeros(1, 192), flow_embedding=torch.zeros(1, 192))
        outs = [o["tts_speech"] for o in sch.submit(stream=True, **req(1))]
        # prompt 8 -> pad 2: chunks end at token 7, 17, 37 (hops 5+2, 10, 20), each needs 3 lookahead tokens; 47 tokens in total
        calls = [c[1:] for c in fm.calls]
        assert calls == [(10, 0, True, False), (20, 7, True, False), (40, 17, True, False), (47, 37, False, True)]
        assert [o.shape[1] // 960 for o in outs] == [7, 10, 20, 10]
        fm.calls.clear()
        outs = [o["tts_speech"] for o in sch.submit(stream=False, **req(2))]
        assert [c[1:] for c in fm.calls] == [(9, 0, False, True)] and len(outs) == 1
        with pytest.raises(ValueError, match="empty text"):
            list(sch.submit(stream=True, **req(0)))
        assert not sch._reqs and not fm.hift_cache_dict
        # two interleaved streams keep their own schedules
        fm.calls.clear()
        got = [None, None]
        th = [threading.Thread(target=lambda i=i: got.__setitem__(i, [o["tts_speech"].shape[1] // 960 for o in sch.submit(stream=True, **req(1, 8 if i == 0 else 5))])) for i in (0, 1)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert got[0] == [7, 10, 20, 10] and got[1] == [5, 10, 20, 12]
    finally:
        sch.shutdown()


@pytest.mark.timeout(120)
def test_scheduler_batches_ready_requests_into_one_flow_pass():
    """Round 3: ready work of the same kind and similar length goes through token2wav_batch together (StreamScheduler._mates): with ONE vocoder
    lane that takes 50 ms per call, the requests that become ready while it is busy share the next call; every request still gets exactly its
    own chunks (offsets chain, the final call sees everything), requests of very different length or kind are never put together, and
    chunk_batch = 1 restores the one-by-one behaviour."""
    import time as _t
    scripts = {1: list(range(47)), 2: list(range(9)), 3: list(range(47))}

    class Fm(_FakeModel):
        flow_batch, flow_pad = 4, 1.25

        def token2wav(self, **kw):
            _t.sleep(0.05)
            return super().token2wav(**kw)

        def token2wav_batch(self, jobs, stream=False, finalize=False, on_ready=None):
            _t.sleep(0.05)
            self.batches.append([(j["uuid"], j["token"].shape[1], j["token_offset"], stream, finalize) for j in jobs])
            n = [j["token"].shape[1] + j["prompt_token"].shape[1] for j in jobs]
            assert max(n) <= 1.25 * min(n), "requests of too different length in one pass"
            for i, j in enumerate(jobs):
                on_ready(i, _FakeModel.token2wav(self, stream=stream, finalize=finalize, **j))

    for chunk_batch in (None, 1):
        fm = Fm(scripts); fm.batches = []
        sch = StreamScheduler(fm, slots=8, step_chunk=4, chunk_batch=chunk_batch)
        try:
            got = [None] * 6
            kinds = [(1, True), (3, True), (1, True), (2, False), (3, True), (2, False)]        # (text length -> script, stream)
            th = [threading.Thread(target=lambda i=i, k=k: got.__setitem__(i, [o["tts_speech"].shape[1] // 960 for o in sch.submit(stream=k[1], **_fake_req(k[0]))]))
                  for i, k in enumerate(kinds)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            for g, k in zip(got, kinds):
                assert g == ([7, 10, 20, 10] if k[1] else [9]), (g, k)
            per = {}
            for c in fm.calls:
                per.setdefault(c[0], []).append(c[1:])
            for key, calls in per.items():                       # every request saw exactly its own schedule, whoever it shared a pass with
                assert calls in ([(10, 0, True, False), (20, 7, True, False), (40, 17, True, False), (47, 37, False, True)], [(9, 0, False, True)]), calls
            assert not sch._reqs and not fm.hift_cache_dict
            if chunk_batch == 1:
                assert not fm.batches and sch.batched_jobs == 0
            else:
                assert sch.batched_jobs >= 2 and all(len(b) >= 2 and len({(x[3], x[4]) for x in b}) == 1 for b in fm.batches)
                # a shared pass that fails costs nobody else their request: a member that already got its audio keeps it, the others are vocoded again
                # one by one, and only a request that fails ALONE ends with its error; the server goes on
                boom = RuntimeError("flow pass failed")

                def failing(jobs, stream=False, finalize=False, on_ready=None):
                    i = next(k for k, j in enumerate(jobs) if j["prompt_token"].shape[1] != 5)     # one healthy member gets its audio before the pass dies
                    on_ready(i, _FakeModel.token2wav(fm, stream=stream, finalize=finalize, **jobs[i]))
                    raise boom
                fm.token2wav_batch = failing
                single = Fm.token2wav

                def poisoned(self, **kw):                       # a malformed request: fails on its own as well (5 prompt tokens mark it)
                    if kw["prompt_token"].shape[1] == 5:
                        raise ValueError("bad prompt_feat")
                    return single(self, **kw)
                Fm.token2wav = poisoned
                errs, oks = [], []

                def one(n_prompt):
                    try:
                        oks.append([o["tts_speech"].shape[1] // 960 for o in sch.submit(stream=False, **_fake_req(2, n_prompt))])
                    except (RuntimeError, ValueError) as e:
                        errs.append(e)
                th = [threading.Thread(target=one, args=(n,)) for n in (8, 5, 8, 8)]
                for t in th:
                    t.start()
                for t in th:
                    t.join()
                Fm.token2wav = single
                assert len(errs) == 1 and isinstance(errs[0], ValueError) and oks == [[9], [9], [9]], (errs, oks)
                assert not sch._reqs and not fm.hift_cache_dict
                del fm.token2wav_batch                                # back to the class's method
                assert [o["tts_speech"].shape[1] // 960 for o in sch.submit(stream=False, **_fake_req(2))] == [9]
        finally:
            sch.shutdown()


def _fake_req(n_text, n_prompt=8):
    return dict(text=torch.zeros(1, n_text, dtype=torch.int32), prompt_text=torch.zeros(1, 2, dtype=torch.int32),
                llm_prompt_speech_token=torch.zeros(1, n_prompt, dtype=torch.int32), flow_prompt_speech_token=torch.zeros(1, n_prompt, dtype=torch.int32),
                prompt_speech_feat=torch.zeros(1, 2 * n_prompt, 80), llm_embedding=torch.zeros(1, 192), flow_embedding=torch.zeros(1, 192))


@pytest.mark.timeout(120)
def test_scheduler_time_based_strategy():
    """strategy="time_based" (runtime/triton_trtllm/model_repo/cosyvoice2/1/model.py:410-426): after every chunk the next hop is chosen from how far
    synthesis runs ahead of playback - the base hop when it does not (multiplier <= 2), all pending whole hops when it does.  Whatever hops
    are chosen, the chunks must tile the token sequence exactly once: offsets chain, every non-final call sees hop + lookahead tokens past its offset,
    the final call sees everything; every hop is a positive multiple of the base hop (the static chunk size the flow was trained with)."""
    import time as _time
    n_tok, n_calls = 60, {}
    for slow in (False, True):
        fm = _FakeModel({1: list(range(n_tok))})
        if slow:                                            # synthesis slower than playback: multiplier < 0 -> the hop stays at the base
            inner = fm.token2wav
            fm.token2wav = lambda **kw: (_time.sleep(0.3), inner(**kw))[1]      # 0.3 s per 5-token (0.2 s of audio) chunk
        sch = StreamScheduler(fm, slots=2, strategy="time_based", step_chunk=64)
        try:
            outs = [o["tts_speech"] for o in sch.submit(stream=True, **_fake_req(1))]
        finally:
            sch.shutdown()
        calls = [c[1:] for c in fm.calls]
        assert calls[-1][0] == n_tok and calls[-1][3] is True and all(c[3] is False and c[2] is True for c in calls[:-1])
        offs = [c[1] for c in calls]
        hops = [b - a for a, b in zip(offs[:-1], offs[1:])]
        assert offs[0] == 0 and hops[0] == 5 + 2                                     # first hop = base + prompt pad (prompt 8 -> pad 2)
        assert all(h > 0 and h % 5 == 0 for h in hops[1:]), hops
        assert all(c[0] == c[1] + h + 3 for c, h in zip(calls[:-1], hops)), (calls, hops)   # tokens seen = offset + hop + lookahead
        assert sum(o.shape[1] for o in outs) == n_tok * 960                          # every token vocoded exactly once
        n_calls[slow] = len(calls)
        if slow:
            assert all(h == 5 for h in hops[1:4]), hops                             # behind real time: the base hop
        else:
            assert all(h > 5 for h in hops[1:]), hops                               # far ahead of real time: every later hop takes all pending whole hops (+ 1)
        assert not sch._reqs and not fm.hift_cache_dict
    assert n_calls[False] < n_calls[True], n_calls                                  # running ahead means fewer, longer chunks


@pytest.mark.timeout(120)
def test_scheduler_silent_filter_abandoned_client_and_dead_lm():
    """(1) The silent / breath-token rule of llm_job (cli/model.py:122-128) applies to scheduler-served requests, with the run count carried
    across decode chunks.  (2) A client that stops listening frees its request state, its LM sequence is cancelled.  (3) When the LM thread
    dies, open requests fail and submit() refuses new ones instead of queueing them forever."""
    from cosyvoice_amd.model import SilentTokenFilter
    script = [7, 1, 1, 2, 1, 2, 1, 1, 2, 9, 1, 1, 1, 1, 1, 1, 1, 4]            # a run of 8 silent ids straddling the 4-step chunks, then a run of 7
    want = SilentTokenFilter([1, 2])(script)
    assert want == [7, 1, 1, 2, 1, 2, 9, 1, 1, 1, 1, 1, 4]
    fm = _FakeModel({1: script, 2: list(range(400))})
    fm.silent_tokens = [1, 2]
    seen = []
    inner = fm.token2wav
    fm.token2wav = lambda **kw: (seen.append(kw["token"].flatten().tolist()), inner(**kw))[1]
    sch = StreamScheduler(fm, slots=2, step_chunk=4)
    try:
        list(sch.submit(stream=False, **_fake_req(1)))
        assert seen == [want]
        # (2) abandon a long streaming request after its first chunk
        cancelled = []
        on = sch._on_tokens
        sch._on_tokens = lambda key, toks, fin, err: cancelled.append(on(key, toks, fin, err)) or cancelled[-1]
        sch.model.llm_on = None
        gen = sch.submit(stream=True, **_fake_req(2))
        next(gen)
        gen.close()
        for _ in range(200):
            if not sch._reqs and not fm.hift_cache_dict:
                break
            threading.Event().wait(0.02)
        assert not sch._reqs and not fm.hift_cache_dict
    finally:
        sch.shutdown()

    class _Boom(_FakeModel._LLM):
        def serve_stream(self, source, on_tokens, **kw):
            source.get()
            raise RuntimeError("LM thread exploded")
    fm = _FakeModel({1: [1, 2, 3]})
    fm.llm = _Boom({})
    sch = StreamScheduler(fm, slots=2)
    try:
        with pytest.raises(RuntimeError, match="exploded"):
            list(sch.submit(stream=True, **_fake_req(1)))
        sch._llm_thread.join(5)
        with pytest.raises(RuntimeError, match="LM thread died"):
            sch.submit(stream=True, **_fake_req(1))
    finally:
        sch._stop = True
        with sch._cv:
            sch._cv.notify_all()
        for t in sch._voc_threads:
            t.join(5)


def test_fastapi_adapter_streams_pcm16(lib, setup):
    from starlette.testclient import TestClient
    if lib.emulated:
        pytest.skip("hardware only: end to end through the real model (the HTTP adapter itself is covered by test_fastapi_adapter_with_stub_engine)")
    cfgs, sds = setup
    m = _model(lib, cfgs, sds)
    sch = StreamScheduler(m, slots=2, step_chunk=4)
    try:
        eng = Engine(sch, _StubFrontend(cfgs))
        client = TestClient(create_app(eng))
        buf = io.BytesIO()
        with wave.open(buf, "wb") as w:
            w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000)
            w.writeframes((np.sin(np.arange(8000) * 0.05) * 8000).astype(np.int16).tobytes())
        r = client.post("/inference_zero_shot", params={"tts_text": "ab|c", "prompt_text": "x"}, content=buf.getvalue())
        assert r.status_code == 200, r.text
        want = b"".join(pcm16_stream(eng.inference_zero_shot("ab|c", "x", torch.zeros(1, 8000))))
        assert len(r.content) == len(want) > 0 and len(r.content) % (2 * 480 * 2) == 0 and r.content == want
        r2 = client.post("/inference_cross_lingual", params={"tts_text": "a"}, content=buf.getvalue())
        assert r2.status_code == 200 and len(r2.content) > 0
    finally:
        sch.shutdown()


@pytest.mark.timeout(120)
def test_fastapi_adapter_with_stub_engine():
    """Routes and the int16 PCM streaming contract of runtime/python/fastapi/server.py:38-86 over a stub engine."""
    from starlette.testclient import TestClient

    class Stub:
        def inference_sft(self, tts_text, spk_id):
            yield {"tts_speech": torch.full((1, 4), 0.5)}

        def inference_zero_shot(self, tts_text, prompt_text, prompt_wav):
            assert prompt_wav.shape == (1, 8000) and abs(float(prompt_wav.abs().max()) - 8000 / 32768.0) < 1e-3
            for ch in tts_text:
                yield {"tts_speech": torch.full((1, 3), ord(ch) / 32768.0)}

        inference_cross_lingual = lambda self, tts_text, prompt_wav: iter([{"tts_speech": torch.zeros(1, 2)}])
        inference_instruct2 = lambda self, tts_text, instruct_text, prompt_wav: iter([{"tts_speech": torch.ones(1, 2) * 0.25}])

    client = TestClient(create_app(Stub()))
    buf = io.BytesIO()
    with wave.open(buf, "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000)
        w.writeframes((np.sign(np.sin(np.arange(8000) * 0.05)) * 8000).astype(np.int16).tobytes())
    r = client.post("/inference_sft", params={"tts_text": "x", "spk_id": "s"})
    assert r.status_code == 200 and np.frombuffer(r.content, dtype=np.int16).tolist() == [16384] * 4
    r = client.post("/inference_zero_shot", params={"tts_text": "ab", "prompt_text": "p"}, content=buf.getvalue())
    assert r.status_code == 200 and np.frombuffer(r.content, dtype=np.int16).tolist() == [97] * 3 + [98] * 3
    r = client.post("/inference_instruct2", params={"tts_text": "ab", "instruct_text": "i"}, content=buf.getvalue())
    assert r.status_code == 200 and np.frombuffer(r.content, dtype=np.int16).tolist() == [8192, 8192]


@pytest.mark.timeout(120)
def test_grpc_adapter_with_stub_engine():
    """runtime/python/grpc/server.py:34-77 + cosyvoice.proto over a stub engine: the four request kinds, the raw int16 prompt, the PCM16 response
    stream, and the wire format (field numbers of the reference's .proto, checked against hand-encoded bytes)."""
    import grpc
    from cosyvoice_amd.serving import GRPC_METHOD, create_grpc_server, grpc_messages
    Request, Response = grpc_messages()
    # wire compatibility: zero_shot_request is field 2 of Request; tts_text / prompt_text / prompt_audio are fields 1 / 2 / 3 of zeroshotRequest
    r = Request(); r.zero_shot_request.tts_text = "ab"; r.zero_shot_request.prompt_text = "p"; r.zero_shot_request.prompt_audio = b"\x01\x02"
    assert r.SerializeToString() == b"\x12\x0b" + b"\x0a\x02ab" + b"\x12\x01p" + b"\x1a\x02\x01\x02"
    assert Response(tts_audio=b"xy").SerializeToString() == b"\x0a\x02xy"
    seen = {}

    class Stub:
        def inference_sft(self, tts_text, spk_id):
            seen["sft"] = (tts_text, spk_id)
            yield {"tts_speech": torch.full((1, 4), 0.5)}

        def inference_zero_shot(self, tts_text, prompt_text, prompt_wav):
            seen["zs"] = (prompt_text, tuple(prompt_wav.shape), float(prompt_wav[0, 1]))
            for ch in tts_text:
                yield {"tts_speech": torch.full((1, 3), ord(ch) / 32768.0)}

        def inference_cross_lingual(self, tts_text, prompt_wav):
            yield {"tts_speech": torch.zeros(1, 2)}

    server, port = create_grpc_server(Stub(), port=0, max_conc=2, host="127.0.0.1")
    server.start()
    try:
        with grpc.insecure_channel("127.0.0.1:%d" % port) as ch:
            call = ch.unary_stream(GRPC_METHOD, request_serializer=Request.SerializeToString, response_deserializer=Response.FromString)
            q = Request(); q.sft_request.spk_id = "s"; q.sft_request.tts_text = "hello"
            out = [np.frombuffer(x.tts_audio, dtype=np.int16).tolist() for x in call(q)]
            assert out == [[16384] * 4] and seen["sft"] == ("hello", "s")
            q = Request(); q.zero_shot_request.tts_text = "ab"; q.zero_shot_request.prompt_text = "p"
            q.zero_shot_request.prompt_audio = np.array([0, 16384, -32768], dtype=np.int16).tobytes()
            out = [np.frombuffer(x.tts_audio, dtype=np.int16).tolist() for x in call(q)]
            assert out == [[97] * 3, [98] * 3] and seen["zs"] == ("p", (1, 3), 0.5)                     # one response per yielded chunk
            q = Request(); q.cross_lingual_request.tts_text = "c"; q.cross_lingual_request.prompt_audio = b"\x00\x00"
            assert [x.tts_audio for x in call(q)] == [b"\x00\x00\x00\x00"]
            q = Request(); q.instruct_request.tts_text = "t"; q.instruct_request.spk_id = "s"; q.instruct_request.instruct_text = "i"
            with pytest.raises(grpc.RpcError) as e:
                list(call(q))
            assert e.value.code() == grpc.StatusCode.UNIMPLEMENTED
    finally:
        server.stop(0)


@pytest.mark.timeout(120)
def test_solo_retry_starts_from_the_cache_the_chunk_started_from_even_when_the_model_updates_it_in_place():
    """CosyVoice3Model._t2w_tail updates a request's vocoder cache IN PLACE (cache['mel'] = concat(...), cache['speech_offset'] += ...).  The scheduler's
    snapshot for the solo retry after a failed shared pass must therefore be a copy: here the model keeps a per-request dict {'speech_offset': samples vocoded so far}
    that token2wav advances in place and checks against the chunk it is given; the shared pass advances EVERY member's cache and then dies before any member got
    its audio.  With a reference snapshot the retry would see the advanced offset (the model raises); with the copy every stream gets each chunk exactly once."""
    scripts = {1: list(range(47))}

    class Cv3Like(_FakeModel):
        flow_batch, flow_pad = 4, 1.25

        def token2wav(self, token, prompt_token, prompt_feat, embedding, token_offset, uuid, stream=False, finalize=False, speed=1.0):
            with self.lock:
                cache = self.hift_cache_dict.get(uuid)
                if cache is None:
                    cache = self.hift_cache_dict[uuid] = {"speech_offset": 0}
            if cache["speech_offset"] != token_offset * 960:
                raise AssertionError("chunk at token %d vocoded against a cache at sample %d" % (token_offset, cache["speech_offset"]))
            out = _FakeModel.token2wav(self, token, prompt_token, prompt_feat, embedding, token_offset, uuid, stream, finalize, speed)
            cache["speech_offset"] += out.shape[1]                  # in place, like CosyVoice3Model._t2w_tail
            return out

        def token2wav_batch(self, jobs, stream=False, finalize=False, on_ready=None):
            self.n_batches = getattr(self, "n_batches", 0) + 1
            outs = [self.token2wav(stream=stream, finalize=finalize, **j) for j in jobs]       # every member's cache advances ...
            if self.n_batches % 2 == 1:
                del self.calls[-len(jobs):]
                raise RuntimeError("the shared pass died before anyone was served")           # ... then every other pass dies
            for i, o in enumerate(outs):
                on_ready(i, o)

    shared = 0
    for attempt in range(8):                                        # whether three streams' chunks meet in ONE pass is a matter of thread timing (a loaded machine can serve them
        fm = Cv3Like(scripts)                                       # one by one): every attempt must be served correctly, and one of them must have gone through a shared pass
        sch = StreamScheduler(fm, slots=8, step_chunk=4)
        try:
            got = [None] * 3
            th = [threading.Thread(target=lambda i=i: got.__setitem__(i, [o["tts_speech"].shape[1] // 960 for o in sch.submit(stream=True, **_fake_req(1))])) for i in range(3)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            assert got == [[7, 10, 20, 10]] * 3, got
            deadline = time.time() + 5.0                            # (a worker hands the last chunk over BEFORE it drops the request's state: the client can be back first)
            while (sch._reqs or fm.hift_cache_dict) and time.time() < deadline:
                time.sleep(0.01)
            assert not sch._reqs and not fm.hift_cache_dict
            shared += getattr(fm, "n_batches", 0)
        finally:
            sch.shutdown()
        if shared:
            break
    assert shared >= 1
