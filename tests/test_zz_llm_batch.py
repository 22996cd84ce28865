"""Lock-step batched decode (llm_batch_kernels.h, skinny GEMMs on the exact-fp32 MFMA): a sequence decoded in a batch must yield exactly
the tokens it yields alone - and exactly the oracle's - whatever its slot and whatever the other slots hold.  Runs under the CPU emulator (guard-page memory) and on the MI355X."""
import pytest
import torch

from cosyvoice_amd.llm import Qwen2LM
from oracle import llm as OL
from cosyvoice_amd import synthetic as W


def _req(cfg, seed, n_text, n_prompt_text, n_prompt_tok):
    u = W.synthetic_utterance(cfg, W.tiny()[1], n_prompt_tok=n_prompt_tok, n_prompt_text=n_prompt_text, n_text=n_text, seed=seed)
    return dict(text=u["text"], prompt_text=u["prompt_text"], prompt_speech_token=u["llm_prompt_speech_token"])


@pytest.mark.parametrize("use_graph", [True, False])
def test_batch_matches_single_and_oracle(lib, use_graph):
    if lib.emulated and not use_graph:
        pytest.skip("eager launches replay the same closures as the captured graph under the emulator")
    cfg = W.tiny()[0]
    sd = W.make_llm(cfg)
    lm = Qwen2LM(sd, cfg, lib=lib, max_len=160, sampling="greedy", decode_chunk=5, use_graph=use_graph)
    # different prompt lengths, different generated lengths (some stop on eos early, some run to max_len), one slot without a speech prompt
    reqs = [_req(cfg, 1986, 6, 5, 11), _req(cfg, 7, 4, 3, 20), _req(cfg, 11, 5, 0, 0), _req(cfg, 23, 3, 2, 33), _req(cfg, 5, 6, 4, 9)]
    got = lm.inference_batch(reqs, max_token_text_ratio=4, min_token_text_ratio=1)
    assert len(got) == len(reqs)
    t = lambda n: torch.tensor([n], dtype=torch.int32)
    lens = set()
    for r, g in zip(reqs, got):
        want = OL.inference(sd, cfg, r["text"], r["prompt_text"], r["prompt_speech_token"], max_token_text_ratio=4, min_token_text_ratio=1)
        alone = list(lm.inference(text=r["text"], text_len=t(r["text"].shape[1]), prompt_text=r["prompt_text"], prompt_text_len=t(r["prompt_text"].shape[1]),
                                  prompt_speech_token=r["prompt_speech_token"], prompt_speech_token_len=t(r["prompt_speech_token"].shape[1]),
                                  max_token_text_ratio=4, min_token_text_ratio=1))
        assert g == alone == want
        lens.add(len(g))
    assert len(lens) > 1                                             # the slots really finished at different steps
    # a second batch of another size on the same handle, then the single-sequence path again
    got2 = lm.inference_batch(reqs[:2], max_token_text_ratio=3, min_token_text_ratio=1)
    for r, g in zip(reqs[:2], got2):
        assert g == OL.inference(sd, cfg, r["text"], r["prompt_text"], r["prompt_speech_token"], max_token_text_ratio=3, min_token_text_ratio=1)


@pytest.mark.experiments
def test_deep_down_projection_variant(lib, monkeypatch):
    """CV_DOWN_DEEP=1: the down projection of the batched step as ONE launch (skinny_deep_kernel: 16-wave workgroups over the whole K, a ring of
    k-tiles per wave) instead of split-K partials + sum_partials_kernel.  Measured slower on the MI355X and off by default (llm.hip), but a kept
    alternative stays a tested one: same tokens as the oracle."""
    monkeypatch.setenv("CV_DOWN_DEEP", "1")
    cfg = W.tiny()[0]
    sd = W.make_llm(cfg)
    lm = Qwen2LM(sd, cfg, lib=lib, max_len=160, sampling="greedy", decode_chunk=5)
    reqs = [_req(cfg, 1986, 6, 5, 11), _req(cfg, 7, 4, 3, 20), _req(cfg, 23, 3, 2, 33)]
    got = lm.inference_batch(reqs, max_token_text_ratio=4, min_token_text_ratio=1)
    for r, g in zip(reqs, got):
        assert g == OL.inference(sd, cfg, r["text"], r["prompt_text"], r["prompt_speech_token"], max_token_text_ratio=4, min_token_text_ratio=1)


def test_row_major_skinny_variant(lib):
    """Option batch_packed = 0: the batched GEMMs on the row-major weight tensors (skinny_mfma_kernel, round 2) instead of the fragment-ordered copies +
    wave-private activation staging (skinny_pk_kernel, round 3, the default).  Same products in the same order: the same tokens, and the oracle's."""
    import ctypes as C
    cfg = W.tiny()[0]
    sd = W.make_llm(cfg)
    reqs = [_req(cfg, 1986, 6, 5, 11), _req(cfg, 7, 4, 3, 20), _req(cfg, 23, 3, 2, 33)]
    outs = []
    for packed in (1, 0):
        lm = Qwen2LM(sd, cfg, lib=lib, max_len=160, sampling="greedy", decode_chunk=5)
        lib.cv_llm_set_option(lm._h, b"batch_packed", C.c_int32(packed))
        outs.append(lm.inference_batch(reqs, max_token_text_ratio=4, min_token_text_ratio=1))
    assert outs[0] == outs[1]
    for r, g in zip(reqs, outs[0]):
        assert g == OL.inference(sd, cfg, r["text"], r["prompt_text"], r["prompt_speech_token"], max_token_text_ratio=4, min_token_text_ratio=1)


def test_batch_of_eight_and_long_context(lib):
    cfg = W.tiny()[0]
    sd = W.make_llm(cfg)
    lm = Qwen2LM(sd, cfg, lib=lib, max_len=512, sampling="greedy", decode_chunk=8)
    # contexts from 15 to ~420 keys: one and two attention passes (384 keys each), the second one with a partly filled last wave
    reqs = [_req(cfg, 100 + i, 3 + (i % 3), 2, 10 + 25 * i if i < 6 else 300 + 50 * (i - 5)) for i in range(8)]
    got = lm.inference_batch(reqs, max_token_text_ratio=3, min_token_text_ratio=2)
    for r, g in zip(reqs, got):
        assert g == OL.inference(sd, cfg, r["text"], r["prompt_text"], r["prompt_speech_token"], max_token_text_ratio=3, min_token_text_ratio=2)
    with pytest.raises(AssertionError):
        lm.inference_batch(reqs * 4 + reqs[:1])                       # 33 > MAX_NB


@pytest.mark.parametrize("nb", [17, 32])
def test_more_than_sixteen_slots(lib, nb):
    """Round 4: 17 .. 32 sequences per lock-step step - a second MFMA column tile per weight fragment (skinny_pk2_kernel): every slot still yields exactly the
    tokens of its request alone (= the oracle's), whatever its column tile, with different prompt lengths and early stops; then a smaller batch and a
    continuous-batching queue over 20 slots on the same handle."""
    cfg = W.tiny()[0]
    sd = W.make_llm(cfg)
    lm = Qwen2LM(sd, cfg, lib=lib, max_len=160, sampling="greedy", decode_chunk=6)
    reqs = [_req(cfg, 300 + i, 2 + (i % 4), 2 + (i % 3), 4 + 5 * (i % 7)) for i in range(nb)]
    want = [OL.inference(sd, cfg, r["text"], r["prompt_text"], r["prompt_speech_token"], max_token_text_ratio=3, min_token_text_ratio=1) for r in reqs]
    got = lm.inference_batch(reqs, max_token_text_ratio=3, min_token_text_ratio=1)
    assert got == want and len({len(g) for g in got}) > 1
    assert lm.inference_batch(reqs[:5], max_token_text_ratio=3, min_token_text_ratio=1) == want[:5]
    if nb == 32:
        q = dict(lm.inference_queue(reqs, slots=20, max_token_text_ratio=3, min_token_text_ratio=1))
        assert [q[i] for i in range(nb)] == want


@pytest.mark.parametrize("heads,kv_heads", [(14, 2), (6, 2)])
def test_grouped_query_attention_variants(lib, heads, kv_heads, monkeypatch):
    """The batched decode attention in all its forms, at CosyVoice2's group shape (7 heads per kv head) and at 3 per group, with contexts that cross the 192-key pass
    boundary of the per-head kernels and the 16-key tiles of the MFMA kernel:
      * round 5 default: attn_decode_batch_mfma_kernel (workgroup = (sequence, kv head, key slice), the group's heads as MFMA columns) + attn_merge_batch_kernel,
        with the slice count chosen by the launch rule, forced to 1 (no merge launch), 3 (slices that hold no key) and with 8-wave workgroups;
      * CV_ATTN_BATCH=0: one workgroup per head (rounds 2-4), and its CV_ATTN_BATCH_GQA=1 form (one workgroup per (sequence, kv head), measured slower).
    Every form yields the oracle's tokens in every slot."""
    import dataclasses
    cfg = dataclasses.replace(W.tiny()[0], heads=heads, kv_heads=kv_heads)                  # (hidden stays 128: the projections are 128 -> 64 * heads)
    sd = W.make_llm(cfg)
    reqs = [_req(cfg, 500 + i, 3, 2, (188, 7, 40, 195)[i]) for i in range(4)]          # prompt + text + generated tokens: ~200 / ~20 / ~50 / ~210 keys
    want = [OL.inference(sd, cfg, r["text"], r["prompt_text"], r["prompt_speech_token"], max_token_text_ratio=4, min_token_text_ratio=2) for r in reqs]
    knobs = [{"CV_ATTN_BATCH": "1"}, {"CV_ATTN_BATCH": "1", "CV_ATTN_BATCH_SLICES": "1"}, {"CV_ATTN_BATCH": "1", "CV_ATTN_BATCH_SLICES": "3", "CV_ATTN_BATCH_WAVES": "8"},
             {"CV_ATTN_BATCH": "0"}, {"CV_ATTN_BATCH": "0", "CV_ATTN_BATCH_GQA": "1"}, {}]      # {}: the launch rule (few slots, short contexts: the per-head form)
    if not lib.experiments:
        knobs = [k for k in knobs if "CV_ATTN_BATCH_GQA" not in k]                  # (the grouped form is a CV_BUILD_EXPERIMENTS kernel)
    if heads == 6:
        knobs = knobs[:2] + knobs[3:4]
    for env in knobs:
        for k in ("CV_ATTN_BATCH", "CV_ATTN_BATCH_SLICES", "CV_ATTN_BATCH_WAVES", "CV_ATTN_BATCH_GQA"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)                                                # read when the step of a handle is captured
        lm = Qwen2LM(sd, cfg, lib=lib, max_len=256, sampling="greedy", decode_chunk=5)
        assert lm.inference_batch(reqs, max_token_text_ratio=4, min_token_text_ratio=2) == want, env


def test_attention_form_changes_with_the_context(lib):
    """The decode attention of a lock-step batch is chosen per decode call from the slot count and the longest live context (llm.hip batch_decode: per-head VALU form for
    few slots at short contexts, MFMA form + merge launch otherwise; one captured graph each, both kept).  12 slots whose contexts grow across the rule's threshold (416)
    during the request: the step switches form between two decode chunks - tokens stay the oracle's, and a second batch on the handle re-uses both graphs."""
    cfg = W.tiny()[0]
    sd = W.make_llm(cfg)
    lm = Qwen2LM(sd, cfg, lib=lib, max_len=512, sampling="greedy", decode_chunk=6)
    reqs = [_req(cfg, 700 + i, 4, 2, 395 + (i % 3)) for i in range(12)]               # contexts ~404 .. ~420 and growing by up to 16 tokens
    want = [OL.inference(sd, cfg, r["text"], r["prompt_text"], r["prompt_speech_token"], max_token_text_ratio=4, min_token_text_ratio=3) for r in reqs]
    for _ in range(2):
        assert lm.inference_batch(reqs, max_token_text_ratio=4, min_token_text_ratio=3) == want
    assert max(len(w) for w in want) >= 12


def test_continuous_batching(lib):
    """inference_queue: 7 requests through 3 slots - finished slots are re-filled while the others keep decoding; every request gets
    exactly the tokens it gets alone."""
    cfg = W.tiny()[0]
    sd = W.make_llm(cfg)
    lm = Qwen2LM(sd, cfg, lib=lib, max_len=160, sampling="greedy", decode_chunk=4)
    reqs = [_req(cfg, 200 + i, 2 + (i % 4), 2 + (i % 2), 5 + 7 * (i % 3)) for i in range(7)]
    got = dict(lm.inference_queue(reqs, slots=3, max_token_text_ratio=4, min_token_text_ratio=1))
    assert sorted(got) == list(range(7))
    for i, r in enumerate(reqs):
        assert got[i] == OL.inference(sd, cfg, r["text"], r["prompt_text"], r["prompt_speech_token"], max_token_text_ratio=4, min_token_text_ratio=1)
    assert list(lm.inference_queue([], slots=3)) == []


@pytest.mark.parametrize("sampling", ["greedy", "ras"])
def test_decode_groups(lib, sampling):
    """Round 5: a lock-step batch of at least group_min_slots sequences is cut into `decode_groups` independent chains, each on a sibling handle (the same weight
    tensors, its own KV cache), stream and host thread.  Slots are independent, so every request gets exactly the tokens of the uncut batch - also under 'ras'
    sampling, where the chains key a request's sampler stream by the number it would have had on the one handle - through inference_batch and inference_queue."""
    cfg = W.tiny()[0]
    sd = W.make_llm(cfg)
    reqs = [_req(cfg, 300 + i, 2 + (i % 4), 2 + (i % 2), 5 + 7 * (i % 3)) for i in range(7)]
    one = Qwen2LM(sd, cfg, lib=lib, max_len=160, sampling=sampling, decode_chunk=4, decode_groups=1)
    two = Qwen2LM(sd, cfg, lib=lib, max_len=160, sampling=sampling, decode_chunk=4, decode_groups=2, queue_groups=2, group_min_slots=4)
    want_b = one.inference_batch(reqs[:5], max_token_text_ratio=4, min_token_text_ratio=1)
    want_q = dict(one.inference_queue(reqs, slots=4, max_token_text_ratio=4, min_token_text_ratio=1))
    got_b = two.inference_batch(reqs[:5], max_token_text_ratio=4, min_token_text_ratio=1)
    assert len(two._siblings) == 1 and two._siblings[0]._h.value != two._h.value          # the cut really happened
    got_q = dict(two.inference_queue(reqs, slots=4, max_token_text_ratio=4, min_token_text_ratio=1))
    assert got_b == want_b
    assert sorted(got_q) == list(range(7)) and got_q == want_q
    if sampling == "greedy":
        for r, g in zip(reqs[:5], got_b):
            assert g == OL.inference(sd, cfg, r["text"], r["prompt_text"], r["prompt_speech_token"], max_token_text_ratio=4, min_token_text_ratio=1)
    small = two.inference_batch(reqs[:3], max_token_text_ratio=4, min_token_text_ratio=1)   # below group_min_slots: the one handle
    assert small == one.inference_batch(reqs[:3], max_token_text_ratio=4, min_token_text_ratio=1)
    if sampling == "greedy" and lib.emulated:                   # (host-side state only; added after the round's last GPU session, so it has run under the emulator only)
        # two server threads batch on the SAME handle at once (gRPC workers share the model, runtime/python/grpc/server.py:69): each cut call gets its own tokens and the
        # cut stays a property of the call - the handle's `decode_groups` is what it was (it used to be parked at 1 while a cut batch ran)
        import threading
        res = {}
        ths = [threading.Thread(target=lambda k=k: res.__setitem__(k, two.inference_batch(reqs[k:k + 5], max_token_text_ratio=4, min_token_text_ratio=1))) for k in (0, 2)]
        [t.start() for t in ths]
        [t.join() for t in ths]
        assert res[0] == want_b and res[2] == one.inference_batch(reqs[2:7], max_token_text_ratio=4, min_token_text_ratio=1)
        assert two.decode_groups == 2 and two.queue_groups == 2 and len(two._siblings) == 1


def test_model_tts_batch(lib):
    """CosyVoice2Model.tts_batch == tts() per request (same tokens from the batched LM, same flow / HiFT)."""
    import dataclasses
    from cosyvoice_amd.model import CosyVoice2Model
    lc, fc, hc = W.tiny()
    fc = dataclasses.replace(fc, n_timesteps=1)                       # one Euler step: the emulator run stays short
    m = CosyVoice2Model.from_state_dicts(W.make_llm(lc), W.make_flow(fc), W.make_hift(hc), (lc, fc, hc), lib=lib, max_len=160, sampling="greedy")
    inf = m.hift.inference
    m.hift.inference = lambda speech_feat, cache_source=None: inf(speech_feat, cache_source, noise=torch.zeros(speech_feat.shape[2] * 480, 9))
    # both LM entry points with a small max ratio (llm_job / tts_batch use the reference defaults 20 / 2): 5 tokens per utterance
    inf_b, inf_1 = m.llm.inference_batch, m.llm.inference
    m.llm.inference_batch = lambda reqs: inf_b(reqs, max_token_text_ratio=5, min_token_text_ratio=2)
    m.llm.inference = lambda **kw: inf_1(**{**kw, "max_token_text_ratio": 5, "min_token_text_ratio": 2})
    us = [W.synthetic_utterance(lc, fc, n_prompt_tok=5 + 2 * i, n_prompt_text=2, n_text=1, seed=40 + i) for i in range(2)]
    keys = ("text", "flow_embedding", "llm_embedding", "prompt_text", "llm_prompt_speech_token", "flow_prompt_speech_token", "prompt_speech_feat")
    reqs = [{k: u[k] for k in keys} for u in us]
    alone = [next(iter(m.tts(**r, stream=False)))["tts_speech"] for r in reqs]
    got = m.tts_batch(reqs)
    for a, g in zip(alone, got):
        assert torch.equal(g["tts_speech"], a)
    # tts_queue: LM with continuous batching on the LLM thread / stream, vocoding of finished sequences on the caller's
    m.llm.inference_queue = (lambda f: lambda reqs_, slots=8: f(reqs_, slots=slots, max_token_text_ratio=5, min_token_text_ratio=2))(m.llm.inference_queue)
    seen = dict(m.tts_queue(reqs + reqs[:1], slots=2))
    assert sorted(seen) == [0, 1, 2]
    for i, a in enumerate(alone + alone[:1]):
        assert torch.equal(seen[i]["tts_speech"], a)
    # admission order (round 4): "longest_first" hands the LM the requests by decreasing length bound (text ids x max ratio), "fifo" as listed; the indices that come
    # back are the caller's, each job carries its own request.  (The LM and the vocoder are stubbed here: what is under test is the bookkeeping.)
    admitted = []

    def fake_queue(reqs_, slots=8):
        admitted.append([int(r["text"].shape[1]) * r.get("max_token_text_ratio", 20) for r in reqs_])
        for j, r in enumerate(reqs_):
            yield j, [int(r.get("max_token_text_ratio", 20))] * 3                      # "tokens" that name the request they belong to
    m.llm.inference_queue = fake_queue
    m._vocode_all = lambda groups, speed: ((i, {"req": r, "tokens": toks}) for jobs in groups for i, r, toks in jobs)
    mixed = [dict(reqs[1], max_token_text_ratio=6), dict(reqs[0]), dict(reqs[0], max_token_text_ratio=7)]       # bounds 6, 20 (the default ratio), 7 (one text id each)
    for order, want in (("fifo", [6, 20, 7]), ("longest_first", [20, 7, 6])):
        out = dict(m.tts_queue(mixed, slots=2, order=order))
        assert admitted[-1] == want and sorted(out) == [0, 1, 2]
        assert all(out[i]["req"] is mixed[i] and out[i]["tokens"] == [int(mixed[i].get("max_token_text_ratio", 20))] * 3 for i in range(3))


def test_queue_cursor_hands_every_chain_a_length_bucket_first():
    """_Cursor(n, first=[...]) (round 6): chain k's first admissions come from a block of its own in the (length-sorted) request list, then from the shared remainder,
    then from what another chain has not started - every index exactly once whatever the interleaving; cancel() stops admissions."""
    from cosyvoice_amd.llm import _Cursor
    cur = _Cursor(10, first=[3, 3])
    a, b = cur.view(0), cur.view(1)
    assert [a.take(), b.take(), a.take(), a.take(), b.take()] == [0, 3, 1, 2, 4]     # own blocks first: [0, 3) and [3, 6)
    assert a.take() == 6 and b.take() == 5 and b.take() == 7                          # chain 0's block is spent: the remainder; chain 1 finishes its block, then the remainder
    assert [a.take(), b.take(), a.take()] == [8, 9, None]
    cur = _Cursor(5, first=[4, 4])                                                    # fewer requests than slots: the blocks are cut to what exists, an idle chain helps itself
    a, b = cur.view(0), cur.view(1)
    got = [b.take(), b.take(), a.take(), b.take(), a.take(), a.take(), b.take()]
    assert sorted(x for x in got if x is not None) == [0, 1, 2, 3, 4] and got[0] == 4 and got.count(None) == 2
    cur = _Cursor(6, first=[2, 2])
    assert cur.view(1).take() == 2
    cur.cancel()
    assert cur.view(0).take() is None and cur.take() is None
    plain = _Cursor(3)
    assert [plain.take(), plain.take(), plain.take(), plain.take()] == [0, 1, 2, None]


def test_queue_cuts_itself_into_chains_of_sixteen(lib):
    """Round 6: inference_queue with queue_groups = 0 (the default) cuts `slots` >= 32 into chains of sixteen slots on sibling handles (2 at 32 .. 47, 3 from 48) with a
    length bucket of the sorted request list per chain, and pins the decode attention to the form with one key partition per sequence (batch_attn = 2) for the
    duration of the call; every request gets the tokens of the plain one-chain queue and of the oracle, the handle's own attention rule is back afterwards."""
    cfg = W.tiny()[0]
    sd = W.make_llm(cfg)
    reqs = [_req(cfg, 900 + i, 2 + (i % 3), 2, 4 + 5 * (i % 4)) for i in range(35)]
    reqs.sort(key=lambda r: -int(r["text"].shape[1]))                                # tts_queue's order: longest bound first
    lm = Qwen2LM(sd, cfg, lib=lib, max_len=96, sampling="greedy", decode_chunk=4)
    assert lm.queue_groups == 0 and lm.queue_invariant and lm.batch_attn == -1
    got = dict(lm.inference_queue(reqs, slots=34, max_token_text_ratio=3, min_token_text_ratio=1))
    assert len(lm._siblings) == 1                                                    # 34 slots asked for -> two chains of sixteen
    one = Qwen2LM(sd, cfg, lib=lib, max_len=96, sampling="greedy", decode_chunk=4, queue_groups=1)
    one.queue_invariant = False
    want = dict(one.inference_queue(reqs, slots=8, max_token_text_ratio=3, min_token_text_ratio=1))
    assert sorted(got) == list(range(35)) and got == want and not one._siblings
    for i in (0, 17, 34):
        r = reqs[i]
        assert got[i] == OL.inference(sd, cfg, r["text"], r["prompt_text"], r["prompt_speech_token"], max_token_text_ratio=3, min_token_text_ratio=1)
    small = dict(lm.inference_queue(reqs[:5], slots=4, max_token_text_ratio=3, min_token_text_ratio=1))       # below 32 slots: one chain (still the pinned attention form)
    assert small == {i: want[i] for i in range(5)}
