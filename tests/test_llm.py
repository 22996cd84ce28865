"""Stage B2 parity: HIP Qwen2LM (prefill GEMM path + graph-replayed decode path + on-device sampling) vs the oracle."""
import numpy as np
import pytest
import torch

from cosyvoice_amd.llm import CosyVoice3LM, Qwen2LM
from oracle import llm as OL
from oracle import sampling as OS
from cosyvoice_amd import synthetic as W


@pytest.fixture(scope="module")
def tiny_sd():
    cfg = W.tiny()[0]
    return cfg, W.make_llm(cfg)


def _utt(cfg, n_text=6, n_prompt_text=5, n_prompt_tok=11, seed=1986):
    return W.synthetic_utterance(cfg, W.tiny()[1], n_prompt_tok=n_prompt_tok, n_prompt_text=n_prompt_text, n_text=n_text, seed=seed)


def _kw(u):
    t = lambda n: torch.tensor([n], dtype=torch.int32)
    return dict(text=u["text"], text_len=t(u["text"].shape[1]), prompt_text=u["prompt_text"], prompt_text_len=t(u["prompt_text"].shape[1]),
                prompt_speech_token=u["llm_prompt_speech_token"], prompt_speech_token_len=t(u["llm_prompt_speech_token"].shape[1]),
                embedding=u["llm_embedding"])


@pytest.mark.parametrize("use_graph", [True, False])
def test_greedy_tokens_bit_exact(lib, tiny_sd, use_graph):
    """north_star: speech-token ids bit-exact under greedy decode."""
    cfg, sd = tiny_sd
    u = _utt(cfg)
    lm = Qwen2LM(sd, cfg, lib=lib, max_len=128, sampling="greedy", decode_chunk=5, use_graph=use_graph)
    got = list(lm.inference(**_kw(u), max_token_text_ratio=4, min_token_text_ratio=2))
    want = OL.inference(sd, cfg, u["text"], u["prompt_text"], u["llm_prompt_speech_token"], max_token_text_ratio=4, min_token_text_ratio=2)
    assert got == want
    assert len(got) >= 12


@pytest.mark.parametrize("splits", [4, 8, 16])
def test_teacher_forced_logits(lib, tiny_sd, splits):
    """First-step logits after prefill, and logits after each decode step, against the oracle's log-probs (short contexts:
    with 4 slices the last ones are empty for the first steps)."""
    cfg, sd = tiny_sd
    u = _utt(cfg, seed=7)
    lm = Qwen2LM(sd, cfg, lib=lib, max_len=128, sampling="greedy", attn_splits=splits)
    trace = {}
    want = OL.inference(sd, cfg, u["text"], u["prompt_text"], u["llm_prompt_speech_token"], max_token_text_ratio=2, min_token_text_ratio=2, trace=trace)
    lm_input = lm.build_lm_input(u["text"], u["prompt_text"], u["llm_prompt_speech_token"])
    ref_in = OL.build_lm_input(sd, cfg, u["text"], u["prompt_text"], u["llm_prompt_speech_token"])
    torch.testing.assert_close(lm_input.cpu(), ref_in, rtol=0, atol=0)
    lm.prefill(lm_input)
    sp = lm.make_sampling(12, 12)
    for i in range(len(want)):
        toks, fin = lm.decode(1, sp)
        logp = lm.last_logits().log_softmax(-1)
        torch.testing.assert_close(logp, trace["logp"][i], rtol=1e-4, atol=1e-4)
        assert toks == [want[i]]


def test_ras_sampling_matches_oracle_logic(lib, tiny_sd):
    """RAS decision logic on device == oracle restatement of utils/common.py:138-167 given the same uniform variates."""
    cfg, sd = tiny_sd
    u = _utt(cfg, seed=3)
    us = np.random.default_rng(5).random(400).astype(np.float32)
    lm = Qwen2LM(sd, cfg, lib=lib, max_len=128, sampling="ras", decode_chunk=7)
    lm.set_uniforms(us)
    got = list(lm.inference(**_kw(u), max_token_text_ratio=5, min_token_text_ratio=2))
    step = {"i": 0, "fallbacks": 0}

    def sampler(scores, decoded, k):
        i = step["i"]; step["i"] += 1
        before = scores.clone()
        t = OS.ras_sampling(scores, decoded, k, u=(float(us[2 * i]), float(us[2 * i + 1])))
        step["fallbacks"] += int(not torch.equal(before, scores))
        return t
    want = OL.inference(sd, cfg, u["text"], u["prompt_text"], u["llm_prompt_speech_token"], sampling_fn=sampler, max_token_text_ratio=5, min_token_text_ratio=2)
    assert got == want
    assert len(set(got)) > 3


def test_eos_stops_and_min_len(lib, tiny_sd):
    """A stop id ends the stream without being emitted; eos is suppressed below min_len (llm/llm.py:150-160,543-545)."""
    cfg, sd = tiny_sd
    sd2 = dict(sd)
    b = sd["llm_decoder.bias"].clone(); b[cfg.speech_token_size] = 50.0       # eos always wins once allowed
    sd2["llm_decoder.bias"] = b
    u = _utt(cfg)
    lm = Qwen2LM(sd2, cfg, lib=lib, max_len=128, sampling="greedy")
    got = list(lm.inference(**_kw(u), max_token_text_ratio=10, min_token_text_ratio=1.5))
    want = OL.inference(sd2, cfg, u["text"], u["prompt_text"], u["llm_prompt_speech_token"], max_token_text_ratio=10, min_token_text_ratio=1.5)
    assert got == want and len(got) == 9 and all(t < cfg.speech_token_size for t in got)


def test_kv_capacity_is_checked(lib, tiny_sd):
    cfg, sd = tiny_sd
    u = _utt(cfg)
    lm = Qwen2LM(sd, cfg, lib=lib, max_len=32, sampling="greedy")
    with pytest.raises(ValueError):
        list(lm.inference(**_kw(u), max_token_text_ratio=20))


@pytest.mark.parametrize("splits", [4, 8, 16])
def test_long_context_multi_pass_attention(lib, tiny_sd, splits):
    """Contexts > 48 keys per slice take the multi-pass branch of attn_decode_kernel (and > 64-row tiles of the prefill
    attention); splits = key-range slices per head whose partials the o_proj GEMV merges (default 8)."""
    cfg, sd = tiny_sd
    u = _utt(cfg, n_text=3, n_prompt_text=2, n_prompt_tok=900, seed=11)
    lm = Qwen2LM(sd, cfg, lib=lib, max_len=1024, sampling="greedy", decode_chunk=4, attn_splits=splits)
    got = list(lm.inference(**_kw(u), max_token_text_ratio=4, min_token_text_ratio=2))
    trace = {}
    want = OL.inference(sd, cfg, u["text"], u["prompt_text"], u["llm_prompt_speech_token"], max_token_text_ratio=4, min_token_text_ratio=2, trace=trace)
    assert got == want and len(got) >= 1
    lm.prefill(lm.build_lm_input(u["text"], u["prompt_text"], u["llm_prompt_speech_token"]))
    sp = lm.make_sampling(6, 12)
    for i in range(min(2, len(trace["logp"]))):               # the new position attends over all previous keys: multi-pass slices
        lm.decode(1, sp)
        torch.testing.assert_close(lm.last_logits().log_softmax(-1), trace["logp"][i], rtol=1e-4, atol=1e-4)


def test_no_speech_prompt(lib, tiny_sd):
    """prompt_speech_token_len == 0 (cross-lingual / instruct calls): lm_input = [sos | text | task_id] (llm/llm.py:489-494)."""
    cfg, sd = tiny_sd
    u = _utt(cfg, n_text=5, n_prompt_text=0, n_prompt_tok=0, seed=4)
    lm = Qwen2LM(sd, cfg, lib=lib, max_len=128, sampling="greedy")
    got = list(lm.inference(**_kw(u), max_token_text_ratio=3, min_token_text_ratio=1))
    want = OL.inference(sd, cfg, u["text"], u["prompt_text"], u["llm_prompt_speech_token"], max_token_text_ratio=3, min_token_text_ratio=1)
    assert got == want and len(got) >= 1


def test_profile_chain_leaves_handle_usable(lib, tiny_sd):
    """cv_llm_profile_chain (bench.py's live per-kernel timing) replays one kernel class of the decode step; it clobbers the running
    request's activations, so the next request must start with a prefill - after which tokens are bit-exact again."""
    import ctypes as C
    from cosyvoice_amd._lib import stream_ptr
    cfg, sd = tiny_sd
    u = _utt(cfg)
    lm = Qwen2LM(sd, cfg, lib=lib, max_len=128, sampling="greedy", decode_chunk=5)
    want = OL.inference(sd, cfg, u["text"], u["prompt_text"], u["llm_prompt_speech_token"], max_token_text_ratio=4, min_token_text_ratio=2)
    assert list(lm.inference(**_kw(u), max_token_text_ratio=4, min_token_text_ratio=2)) == want
    lm.prefill(lm.build_lm_input(u["text"], u["prompt_text"], u["llm_prompt_speech_token"]))
    lm.decode(2, lm.make_sampling(6, 12))
    for cat in range(6):
        ms, n = C.c_float(), C.c_int32()
        lib.cv_llm_profile_chain(lm._h, cat, 2, C.byref(ms), C.byref(n), stream_ptr(lib))
        assert n.value == (1 if cat == 5 else cfg.layers) * 2 and ms.value >= 0.0
    assert list(lm.inference(**_kw(u), max_token_text_ratio=4, min_token_text_ratio=2)) == want


def test_cosyvoice3_lm(lib):
    """SURVEY.md §8 row a17 (LM part): CosyVoice3LM on the same device kernels - special embeddings from speech_embedding rows, bias-free
    260-way head, 200 stop ids, index speech_token_size masked while below min_len.  Greedy ids bit-exact vs the oracle (which is pinned
    to the real reference class by tests/test_oracle_golden.py), on the golden utterance and on seeds that stop on a special id."""
    import os
    cfg = W.tiny_cv3_llm()
    sd = W.make_llm(cfg)
    lm = CosyVoice3LM(sd, cfg, lib=lib, max_len=160, sampling="greedy", decode_chunk=6)
    g = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "llm_cv3_tiny.npz")).items()}
    t = lambda n: torch.tensor([n], dtype=torch.int32)
    kw = dict(text=g["text"], text_len=t(5), prompt_text=g["prompt_text"], prompt_text_len=t(4), prompt_speech_token=g["prompt_speech_token"],
              prompt_speech_token_len=t(9))
    got = list(lm.inference(**kw, max_token_text_ratio=5, min_token_text_ratio=3))
    assert got == g["tokens"].tolist()                                  # == the real reference's tokens
    lm_input = lm.build_lm_input(g["text"], g["prompt_text"], g["prompt_speech_token"])
    torch.testing.assert_close(lm_input.cpu(), OL.build_lm_input(sd, cfg, g["text"], g["prompt_text"], g["prompt_speech_token"]), rtol=0, atol=0)
    stopped = 0
    for seed in range(3):                                               # other utterances: some end on one of the 200 stop ids before max_len
        u = W.synthetic_utterance(cfg, W.tiny()[1], n_prompt_tok=7, n_prompt_text=3, n_text=4, seed=100 + seed)
        u["text"][0, 0] = cfg.endofprompt_id
        want = OL.inference(sd, cfg, u["text"], u["prompt_text"], u["llm_prompt_speech_token"], max_token_text_ratio=8, min_token_text_ratio=1)
        got = list(lm.inference(**_kw(u), max_token_text_ratio=8, min_token_text_ratio=1))
        assert got == want and all(tok < cfg.speech_token_size for tok in got)
        stopped += len(want) < 32
    assert stopped >= 1
    with pytest.raises(AssertionError):
        list(lm.inference(text=torch.zeros(1, 4, dtype=torch.int32), text_len=t(4), prompt_text=torch.zeros(1, 2, dtype=torch.int32), prompt_text_len=t(2),
                          prompt_speech_token=g["prompt_speech_token"], prompt_speech_token_len=t(9)))


def test_inference_bistream(lib, tiny_sd):
    """Text arriving as a generator (llm/llm.py:551-661): prompt/text 5:15 mixing, forced and sampled-free fill handling, append-mode
    prefill on top of the cache, stop-token id read back from the device, final decode to eos.  Greedy ids bit-exact vs the oracle
    (pinned to the real reference by tests/test_oracle_golden.py) for the golden chunking and for other chunkings / prompt sizes."""
    import os
    cfg, base = tiny_sd
    g = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "llm_bistream_tiny.npz")).items()}
    sd = W.bistream_fixture(base, cfg, float(g["eos_bias"]))
    lm = Qwen2LM(sd, cfg, lib=lib, max_len=256, sampling="greedy", decode_chunk=7)
    t = lambda n: torch.tensor([n], dtype=torch.int32)
    chunks = [g["chunk%d" % i] for i in range(5)]
    got = list(lm.inference_bistream(text=iter(chunks), prompt_text=g["prompt_text"], prompt_text_len=t(4), prompt_speech_token=g["prompt_speech_token"],
                                     prompt_speech_token_len=t(20)))
    assert got == g["tokens"].tolist()                                   # == the real reference
    gen = torch.Generator().manual_seed(9)
    for sizes, n_prompt in (((5, 5, 5), 15), ((1, 1, 9, 2), 7), ((12,), 0), ((2, 2), 31)):
        ch = [torch.randint(0, cfg.text_vocab, (1, n), generator=gen, dtype=torch.int32) for n in sizes]
        pt = torch.randint(0, cfg.text_vocab, (1, 3), generator=gen, dtype=torch.int32)
        ps = torch.randint(0, cfg.speech_token_size, (1, n_prompt), generator=gen, dtype=torch.int32)
        want = OL.inference_bistream(sd, cfg, ch, pt, ps)
        got = list(lm.inference_bistream(text=iter(ch), prompt_text=pt, prompt_text_len=t(3), prompt_speech_token=ps, prompt_speech_token_len=t(n_prompt)))
        assert got == want, (sizes, n_prompt)
    # the offline path still works on the same handle afterwards
    u = _utt(cfg)
    assert list(lm.inference(**_kw(u), max_token_text_ratio=3, min_token_text_ratio=1)) == \
        OL.inference(sd, cfg, u["text"], u["prompt_text"], u["llm_prompt_speech_token"], max_token_text_ratio=3, min_token_text_ratio=1)


@pytest.mark.experiments
@pytest.mark.parametrize("splits,n_prompt", [(8, 11), (4, 300)])
def test_fused_qkv_attention_variant(lib, tiny_sd, splits, n_prompt):
    """Option fused_qkv_attn = 1 (qkv_attn_kernel: RMSNorm + q / k / v rows + RoPE + split attention over the cached keys in one launch, the new
    token joined in the o_proj merge) and head_rows = 2 are alternative decode configurations (measured slower on MI355X, kept selectable):
    same tokens as the oracle, short and multi-pass contexts."""
    import ctypes as C
    cfg, sd = tiny_sd
    u = _utt(cfg, n_prompt_tok=n_prompt)
    lm = Qwen2LM(sd, cfg, lib=lib, max_len=512, sampling="greedy", decode_chunk=6, attn_splits=splits)
    lib.cv_llm_set_option(lm._h, b"fused_qkv_attn", C.c_int32(1))
    lib.cv_llm_set_option(lm._h, b"head_rows", C.c_int32(2))
    got = list(lm.inference(**_kw(u), max_token_text_ratio=2, min_token_text_ratio=2))
    trace = {}
    want = OL.inference(sd, cfg, u["text"], u["prompt_text"], u["llm_prompt_speech_token"], max_token_text_ratio=2, min_token_text_ratio=2, trace=trace)
    assert got == want
    lm.prefill(lm.build_lm_input(u["text"], u["prompt_text"], u["llm_prompt_speech_token"]))
    sp = lm.make_sampling(6, 12)
    for i in range(min(3, len(want) + 1)):                    # step i attends over the prompt (several passes per slice at 300 keys) + i new keys
        lm.decode(1, sp)
        torch.testing.assert_close(lm.last_logits().log_softmax(-1), trace["logp"][i], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("n_prompt", [9, 120])
def test_prefill_weight_stationary_rows_path(lib, tiny_sd, n_prompt):
    """Round 3, option prefill_rows = 1: prompts of up to 160 rows run their GEMMs weight-stationary on the fragment-ordered copies (skinny_rows_kernel:
    RMSNorm in the prologue, SiLU * up in the epilogue, rows of X walked in groups of 16 - two groups with a ragged last one, nine groups) instead of the
    tiled GEMMs (the default; measured equally fast on the MI355X).  Both give the oracle's first-step log-probabilities and greedy tokens, and the K / V
    they leave serve the same decode."""
    import ctypes as C
    cfg, sd = tiny_sd
    u = _utt(cfg, n_prompt_tok=n_prompt)
    trace = {}
    want = OL.inference(sd, cfg, u["text"], u["prompt_text"], u["llm_prompt_speech_token"], max_token_text_ratio=2, min_token_text_ratio=2, trace=trace)
    lm = Qwen2LM(sd, cfg, lib=lib, max_len=256, sampling="greedy", decode_chunk=5)
    logits = []
    for rows in (1, 0):
        lib.cv_llm_set_option(lm._h, b"prefill_rows", C.c_int32(rows))
        x = lm.build_lm_input(u["text"], u["prompt_text"], u["llm_prompt_speech_token"])
        assert x.shape[0] <= 160
        lm.prefill(x)
        lm.decode(1, lm.make_sampling(6, 12))
        logits.append(lm.last_logits().clone())
        torch.testing.assert_close(logits[-1].log_softmax(-1), trace["logp"][0], rtol=1e-4, atol=1e-4)
        assert list(lm.inference(**_kw(u), max_token_text_ratio=2, min_token_text_ratio=2)) == want
    torch.testing.assert_close(logits[0], logits[1], rtol=1e-4, atol=1e-4)
    assert not torch.equal(logits[0], logits[1])                  # (two different summation orders really ran)


@pytest.mark.experiments
@pytest.mark.parametrize("rblocks,waves,n_prompt", [(4, 8, 11), (8, 8, 430), (4, 16, 430)])
def test_fused_attention_oproj_variant(lib, tiny_sd, rblocks, waves, n_prompt):
    """Option fused_attn_oproj = 1 (round 3, attn_oproj_kernel): attention and the o_proj GEMV in one launch, o_proj split by head, the per-head
    contributions summed into the residual by gate / up's prologue (gemv_norm_kernel<.., OPART>) - 4 launches per layer instead of 5.  Same tokens
    as the oracle and the same log-probabilities within the usual bound, for a one-pass context and for one whose key slices take several passes
    (430 + keys over 8 slices of 40 or 16 slices of 24 per pass), 4 and 8 row blocks per head; the KV cache it appends to serves the plain chain afterwards."""
    import ctypes as C
    cfg, sd = tiny_sd
    u = _utt(cfg, n_prompt_tok=n_prompt)
    lm = Qwen2LM(sd, cfg, lib=lib, max_len=512, sampling="greedy", decode_chunk=6)
    lib.cv_llm_set_option(lm._h, b"fused_attn_oproj", C.c_int32(1))
    lib.cv_llm_set_option(lm._h, b"oproj_rblocks", C.c_int32(rblocks))
    lib.cv_llm_set_option(lm._h, b"oproj_waves", C.c_int32(waves))
    got = list(lm.inference(**_kw(u), max_token_text_ratio=2, min_token_text_ratio=2))
    trace = {}
    want = OL.inference(sd, cfg, u["text"], u["prompt_text"], u["llm_prompt_speech_token"], max_token_text_ratio=2, min_token_text_ratio=2, trace=trace)
    assert got == want
    lm.prefill(lm.build_lm_input(u["text"], u["prompt_text"], u["llm_prompt_speech_token"]))
    sp = lm.make_sampling(6, 12)
    for i in range(min(3, len(want) + 1)):
        if i == 2:                                                 # the third step on the plain five-launch chain, over the cache the fused launches appended to
            lib.cv_llm_set_option(lm._h, b"fused_attn_oproj", C.c_int32(0))
        lm.decode(1, sp)
        torch.testing.assert_close(lm.last_logits().log_softmax(-1), trace["logp"][i], rtol=1e-4, atol=1e-4)


@pytest.mark.experiments
@pytest.mark.parametrize("mode,shift", [(1, 0), (2, 0), (1, 3)])
def test_decode_weight_prefetch_is_only_a_hint(lib, tiny_sd, mode, shift):
    """Option prefetch = 1 / 2 (round 3): extra workgroups of the short decode kernels (qkv / attention / o_proj) read the weights the gate / up and
    down GEMVs behind them will stream (llm_kernels.h PrefetchArgs).  Nothing but loads: tokens and log-probabilities are those of the plain chain,
    with and without the graph, whatever consumer the fetchers are pointed at."""
    import ctypes as C
    cfg, sd = tiny_sd
    u = _utt(cfg, n_prompt_tok=37)
    lm = Qwen2LM(sd, cfg, lib=lib, max_len=256, sampling="greedy", decode_chunk=5)
    ref = list(lm.inference(**_kw(u), max_token_text_ratio=2, min_token_text_ratio=2))
    lm.prefill(lm.build_lm_input(u["text"], u["prompt_text"], u["llm_prompt_speech_token"]))
    lm.decode(2, lm.make_sampling(6, 12)); ref_logits = lm.last_logits().clone()
    lib.cv_llm_set_option(lm._h, b"prefetch", C.c_int32(mode))
    lib.cv_llm_set_option(lm._h, b"prefetch_shift", C.c_int32(shift))
    if mode == 2:                                                  # and the head as seven-wave workgroups (option head_waves): same rows, same FMA order
        lib.cv_llm_set_option(lm._h, b"head_waves", C.c_int32(7))
    assert list(lm.inference(**_kw(u), max_token_text_ratio=2, min_token_text_ratio=2)) == ref
    lm.prefill(lm.build_lm_input(u["text"], u["prompt_text"], u["llm_prompt_speech_token"]))
    lm.decode(2, lm.make_sampling(6, 12))
    assert torch.equal(lm.last_logits(), ref_logits)
    assert ref == OL.inference(sd, cfg, u["text"], u["prompt_text"], u["llm_prompt_speech_token"], max_token_text_ratio=2, min_token_text_ratio=2)

