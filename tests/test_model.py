"""Boundary B1: CosyVoice2Model.tts / token2wav end to end (LLM -> flow -> HiFT, offline and streaming) vs the oracle pipeline."""
import dataclasses

import numpy as np
import pytest
import torch

from cosyvoice_amd.model import CosyVoice2Model
from oracle import llm as OL
from oracle import model as OM
from cosyvoice_amd import synthetic as W


@pytest.fixture(scope="module")
def setup():
    lc, fc, hc = W.tiny()
    fc = dataclasses.replace(fc, chunk=5, n_timesteps=2)           # small streaming chunks so the emulator run stays short
    sds = (W.make_llm(lc), W.make_flow(fc), W.make_hift(hc))
    u = W.synthetic_utterance(lc, fc, n_prompt_tok=8, n_prompt_text=4, n_text=2, seed=21)
    return (lc, fc, hc), sds, u


def _build(lib, cfgs, sds):
    m = CosyVoice2Model.from_state_dicts(*sds, cfgs, lib=lib, max_len=160, sampling="greedy")
    m.token_hop_len, m.token_max_hop_len = 5, 20
    inf = m.hift.inference
    m.hift.inference = lambda speech_feat, cache_source=None: inf(speech_feat, cache_source, noise=torch.zeros(speech_feat.shape[2] * 480, 9))
    return m


@pytest.mark.parametrize("stream", [False, True])
def test_tts_matches_oracle(lib, setup, stream):
    cfgs, sds, u = setup
    lc, fc, hc = cfgs
    m = _build(lib, cfgs, sds)
    seen = []                                  # tokens the LM had delivered when each token2wav call returned
    t2w = m.token2wav
    m.token2wav = lambda **kw: (t2w(**kw), seen.append(len(m.tts_speech_token_dict[kw["uuid"]])))[0]
    # the ratio arguments are fixed inside llm_job (20 / 2), so the length is set through the text: 2 text tokens -> 4..40 tokens
    outs = [o["tts_speech"] for o in m.tts(text=u["text"], flow_embedding=u["flow_embedding"], llm_embedding=u["llm_embedding"], prompt_text=u["prompt_text"],
                                           llm_prompt_speech_token=u["llm_prompt_speech_token"], flow_prompt_speech_token=u["flow_prompt_speech_token"],
                                           prompt_speech_feat=u["prompt_speech_feat"], stream=stream)]
    tokens = OL.inference(sds[0], lc, u["text"], u["prompt_text"], u["llm_prompt_speech_token"])
    pipe = OM.Pipeline(sds, cfgs, token_hop_len=5, n_timesteps=2)
    want = pipe.tts(tokens, u, stream=stream)
    assert len(outs) == len(want) and (len(outs) > 2 if stream else len(outs) == 1)
    for a, b in zip(outs, want):
        assert a.shape == b.shape and a.device.type == "cpu"
        # waveform tolerance: the harmonic source amplifies fp32 round-off of f0 (see tests/test_oracle_golden.py), and exp() in
        # the iSTFT magnitude amplifies it again -> 5e-3 absolute on a signal clamped to +-0.99
        torch.testing.assert_close(a, b, rtol=0, atol=5e-3)
    assert sum(o.shape[1] for o in outs) == len(tokens) * 2 * 480
    assert not m.tts_speech_token_dict and not m.hift_cache_dict             # per-request state is cleaned up (cli/model.py:388-391)
    if stream:                                 # first_chunk_exclusive: the LM stood still while the first chunk was vocoded (hop 5 + pad 5 - 4 + look-ahead 3 tokens)
        first_need = 5 + (-u["flow_prompt_speech_token"].shape[1]) % 5 + m.flow.pre_lookahead_len
        assert seen[0] == first_need and seen[-1] == len(tokens) and not m._first_gate


def test_speed_and_vc(lib, setup):
    """speed != 1 (non-stream only, cli/model.py:320-322) and the voice-conversion path (source_speech_token -> vc_job)."""
    cfgs, sds, u = setup
    lc, fc, hc = cfgs
    m = _build(lib, cfgs, sds)
    src = torch.randint(0, fc.vocab, (1, 12), generator=torch.Generator().manual_seed(2), dtype=torch.int32)
    kw = dict(flow_embedding=u["flow_embedding"], flow_prompt_speech_token=u["flow_prompt_speech_token"], prompt_speech_feat=u["prompt_speech_feat"],
              source_speech_token=src)
    a = next(iter(m.tts(**kw)))["tts_speech"]
    b = next(iter(m.tts(speed=1.5, **kw)))["tts_speech"]
    assert a.shape[1] == 12 * 2 * 480 and b.shape[1] == int(24 / 1.5) * 480
    pipe = OM.Pipeline(sds, cfgs, token_hop_len=5, n_timesteps=2)
    torch.testing.assert_close(a, pipe.tts(src[0].tolist(), u, stream=False)[0], rtol=0, atol=5e-3)


def test_two_concurrent_streaming_requests(lib, setup):
    """tts() is called concurrently from server threads (runtime/python/grpc/server.py:69, SURVEY.md §8b threading): two streaming
    requests on ONE model object must each produce exactly what they produce alone (per-uuid state, serialised handles)."""
    import threading
    if lib.emulated:
        pytest.skip("hardware only: ~3 min under the CPU emulator (the LLM-thread + caller-thread pattern is covered by the streaming test)")
    cfgs, sds, u = setup
    lc, fc, hc = cfgs
    m = _build(lib, cfgs, sds)
    u2 = W.synthetic_utterance(lc, fc, n_prompt_tok=6, n_prompt_text=3, n_text=2, seed=33)
    kw = lambda x: dict(text=x["text"], flow_embedding=x["flow_embedding"], llm_embedding=x["llm_embedding"], prompt_text=x["prompt_text"],
                        llm_prompt_speech_token=x["llm_prompt_speech_token"], flow_prompt_speech_token=x["flow_prompt_speech_token"],
                        prompt_speech_feat=x["prompt_speech_feat"], stream=True)
    alone = [torch.cat([o["tts_speech"] for o in m.tts(**kw(x))], dim=1) for x in (u, u2)]
    got, errs = [None, None], []

    def run(i, x):
        try:
            got[i] = torch.cat([o["tts_speech"] for o in m.tts(**kw(x))], dim=1)
        except Exception as e:              # pragma: no cover
            errs.append(repr(e))

    th = [threading.Thread(target=run, args=(i, x)) for i, x in enumerate((u, u2))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    for a, b in zip(got, alone):
        assert torch.equal(a, b)
    assert not m.tts_speech_token_dict and not m.hift_cache_dict


def test_llm_job_silent_token_filter(lib, setup):
    """cli/model.py:101-129: tokens listed in `silent_tokens` (CosyVoice3: FSQ silence / breath ids, :423) are kept for the first
    5 consecutive occurrences and dropped beyond that; any other token resets the run.  Host logic, driven with a stub generator."""
    import threading
    cfgs, sds, u = setup
    m = _build(lib, cfgs, sds)
    m.silent_tokens = [1, 2, 28]
    seq = [5, 1, 2, 1, 2, 1, 2, 28, 1, 7, 1, 1, 1, 1, 1, 1, 1, 9]
    m.llm.inference = lambda **kw: iter(seq)
    uuid = "filter-test"
    m._cond[uuid] = threading.Condition()
    m.tts_speech_token_dict[uuid], m.llm_end_dict[uuid] = [], False
    m.llm_job(u["text"], u["prompt_text"], u["llm_prompt_speech_token"], u["llm_embedding"], uuid)
    want, run = [], 0
    for tok in seq:                                  # restatement of the reference loop
        if tok in m.silent_tokens:
            run += 1
            if run > 5:
                continue
        else:
            run = 0
        want.append(tok)
    assert m.tts_speech_token_dict[uuid] == want == [5, 1, 2, 1, 2, 1, 7, 1, 1, 1, 1, 1, 9]
    assert m.llm_end_dict[uuid] is True


def test_tts_with_streaming_text(lib, setup):
    """tts(text=<generator>) -> llm_job -> Qwen2LM.inference_bistream (cli/model.py:101-117): the audio has exactly the length the
    oracle's bistream token sequence implies."""
    cfgs, sds, u = setup
    lc, fc, hc = cfgs
    llm_sd = W.bistream_fixture(sds[0], lc, 8.0)              # strong eos bias: a short utterance keeps the emulator run short
    m = _build(lib, cfgs, (llm_sd, sds[1], sds[2]))
    g = torch.Generator().manual_seed(3)
    chunks = [torch.randint(0, lc.text_vocab, (1, n), generator=g, dtype=torch.int32) for n in (5,)]
    want = OL.inference_bistream(llm_sd, lc, chunks, u["prompt_text"], u["llm_prompt_speech_token"])
    out = [o["tts_speech"] for o in m.tts(text=(c for c in chunks), flow_embedding=u["flow_embedding"], llm_embedding=u["llm_embedding"],
                                          prompt_text=u["prompt_text"], llm_prompt_speech_token=u["llm_prompt_speech_token"],
                                          flow_prompt_speech_token=u["flow_prompt_speech_token"], prompt_speech_feat=u["prompt_speech_feat"], stream=False)]
    assert len(want) >= 5 and len(out) == 1 and out[0].shape[1] == len(want) * 2 * 480


def test_flow_graph_policy_of_the_model_and_the_scheduler(lib, setup):
    """Round 6: a request served alone is fastest when its flow solve is issued launch by launch, many concurrent streams when their chunk passes replay captured graphs
    (profiles/r6_flow_graph_threshold.txt): CosyVoice2Model asks every lane's flow handle for graph_max_rows = 1 (new lanes and clones included), StreamScheduler for
    3000 while it runs and gives the model's own policy back when it shuts down; the audio does not depend on it."""
    from cosyvoice_amd.serving import StreamScheduler
    cfgs, sds, u = setup
    m = _build(lib, cfgs, sds)
    rows = lambda: [ln.flow._graph_rows for ln in list(m._lane_q.queue)]
    assert m.flow_graph_rows == 1 and rows() == [1]
    m.set_lanes(2)
    assert rows() == [1, 1] and m.flow.clone()._graph_rows == 1
    req = {k: u[k] for k in ("text", "flow_embedding", "llm_embedding", "prompt_text", "llm_prompt_speech_token", "flow_prompt_speech_token", "prompt_speech_feat")}
    alone = torch.cat([o["tts_speech"] for o in m.tts(stream=True, **req)], 1)
    import ctypes as C

    def captured(flow):
        v = C.c_int64(0)
        flow.lib.cv_flow_get_stat(flow._h, b"graph_captures", C.byref(v))
        return v.value
    assert sum(captured(ln.flow) for ln in list(m._lane_q.queue)) == 0
    sch = StreamScheduler(m, slots=2, step_chunk=4)
    try:
        assert m.flow_graph_rows == 3000 and rows() == [3000, 3000]
        served = torch.cat([o["tts_speech"] for o in sch.submit(stream=True, **req)], 1)
    finally:
        sch.shutdown()
    assert m.flow_graph_rows == 1 and rows() == [1, 1]
    assert served.shape == alone.shape and torch.isfinite(served).all()
