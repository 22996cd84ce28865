"""SURVEY.md section 8 row f4 (second half): CosyVoice-300M on the hand-written kernels (cosyvoice_amd/cosyvoice1_hip.py) against golden vectors made
by the REAL reference classes (tests/golden/make_golden_cv1.py --k: TransformerLM, MaskedDiffWithXvec, cli.model.CosyVoiceModel at configs.tiny_cv1_k(),
the 22.05 kHz HiFTGenerator at configs.tiny_cv1()'s vocoder) on the same seeded weights and the same host-RNG seeds.  Every test runs under the emulator
(`-m "not gpu"`) and on the MI355X (`-m gpu`).  The vocoder and the two CosyVoiceModel tests live in files of their own (test_zzz_cosyvoice1_hip_*.py):
under the emulator each is minutes of work, and pytest-xdist hands out whole files.  (File names: they sort after the CosyVoice2 / CosyVoice3 suites.)"""
import torch

from cosyvoice_amd import cosyvoice1 as C1
from cosyvoice_amd import cosyvoice1_hip as CK
from cosyvoice_amd import synthetic as W
from cv1k_common import CFG, HCFG, build_flow, gold, greedy, t


def test_group_norm_matches_torch(lib):
    """cv_group_norm (+ Mish, + per-channel add) against torch.nn.functional.group_norm on the channel-first view, incl. a group that spans the whole utterance."""
    K = CK.Kernels(lib)
    g = torch.Generator().manual_seed(3)
    for B, T, Cc, G in ((2, 37, 32, 8), (1, 211, 80, 1), (2, 9000, 8, 2)):
        x = torch.randn(B, T, Cc, generator=g) * 2 + 0.7
        gamma, beta, add = torch.randn(Cc, generator=g), torch.randn(Cc, generator=g), torch.randn(Cc, generator=g)
        want = torch.nn.functional.mish(torch.nn.functional.group_norm(x.transpose(1, 2), G, gamma, beta)).transpose(1, 2) + add
        got = K.group_norm(K.put(x), B, T, Cc, G, K.put(gamma), K.put(beta), act="mish", col_add=K.put(add)).cpu()
        torch.testing.assert_close(got, want, rtol=2e-5, atol=2e-5)


def test_conv_index_forms_match_torch(lib):
    """Strided Conv1d, ConvTranspose1d (polyphase) and `same` Conv1d of the U-Net as cv_gemm_conv index forms, batch of two channel-last sequences."""
    K = CK.Kernels(lib)
    g = torch.Generator().manual_seed(4)
    F = torch.nn.functional
    for T in (21, 22):
        x = torch.randn(2, T, 32, generator=g)
        w, b = torch.randn(32, 32, 3, generator=g) * 0.1, torch.randn(32, generator=g)
        want = F.conv1d(x.transpose(1, 2), w, b, stride=2, padding=1).transpose(1, 2)
        got, t_out = K.conv_stride(K.put(x), K.mat(w.permute(0, 2, 1).reshape(32, -1), b), 2, T, 32, k=3, stride=2, pad=1)
        assert t_out == want.shape[1]
        torch.testing.assert_close(got.cpu(), want, rtol=1e-5, atol=1e-5)
        wt = torch.randn(32, 24, 4, generator=g) * 0.1
        bt = torch.randn(24, generator=g)
        want = F.conv_transpose1d(x.transpose(1, 2), wt, bt, stride=2, padding=1).transpose(1, 2)
        got, t_out = K.conv_transpose(K.put(x), K.tconv_mat(wt, bt, 2), 2, T, 32, 24, k=4, stride=2, pad=1)
        assert t_out == want.shape[1] == 2 * T
        torch.testing.assert_close(got.cpu(), want, rtol=1e-5, atol=1e-5)
        want = F.relu(F.conv1d(x.transpose(1, 2), w, b, padding=1)).transpose(1, 2)
        torch.testing.assert_close(K.conv(K.put(x), K.conv_mat(w, b), 2, T, pad=1, act="relu").cpu(), want, rtol=1e-5, atol=1e-5)


def test_transformer_lm_tokens_match_reference(lib):
    g = gold("cv1k_llm")
    kw = dict(text=g["text"], text_len=t(7), prompt_text=g["prompt_text"], prompt_text_len=t(4), prompt_speech_token=g["prompt_speech_token"],
              prompt_speech_token_len=t(9), embedding=g["embedding"])
    sd = W.make_cv1_llm(CFG)
    lm = CK.TransformerLM(sd, text_heads=CFG.text_heads, llm_heads=CFG.llm_heads, sampling=greedy, lib=lib)
    ids = torch.cat([g["prompt_text"], g["text"]], 1).reshape(-1)
    torch.testing.assert_close(lm.encode_text(ids).cpu(), g["text_encoded"], rtol=1e-4, atol=1e-4)          # causal ConformerEncoder + affine
    assert list(lm.inference(max_token_text_ratio=6, min_token_text_ratio=2, **kw)) == g["tokens_greedy"].tolist()      # prefill + stepped KV cache
    e0 = torch.zeros(1, 0, dtype=torch.int32)
    sft = dict(kw, prompt_text=e0, prompt_text_len=t(0), prompt_speech_token=e0, prompt_speech_token_len=t(0))
    assert list(lm.inference(max_token_text_ratio=5, min_token_text_ratio=2, **sft)) == g["tokens_sft"].tolist()
    lm.sampling = C1.ras_sampling                                  # repetition-aware sampling on the host RNG: same seed -> the reference's tokens
    torch.manual_seed(7)
    got = list(lm.inference(max_token_text_ratio=6, min_token_text_ratio=2, **kw))
    assert got == g["tokens_ras"].tolist() and 14 <= len(got) <= 42
    # the torch-eager plumbing (configs[0]) agrees at this configuration too
    ref = C1.TransformerLM(sd, text_heads=CFG.text_heads, llm_heads=CFG.llm_heads, sampling=greedy)
    assert list(ref.inference(max_token_text_ratio=6, min_token_text_ratio=2, **kw)) == g["tokens_greedy"].tolist()


def test_kv_state_grows_in_place(lib):
    """forward_chunk's cache rows are reallocated (doubling) when a request outgrows them: stepping row by row equals one causal pass."""
    sd = W.make_cv1_llm(CFG)
    K = CK.Kernels(lib)
    enc = CK.EspnetEncoder(sd, "llm.", CFG.llm_heads, "transformer", kern=K)
    x = torch.randn(9, CFG.llm_dim, generator=torch.Generator().manual_seed(1))
    xd = K.put(x)
    whole, _ = enc.forward_chunk(xd, None)
    state = CK._KVState(K, enc.n_layers, enc.d, 4)                 # cap 4 < 9 rows
    rows = []
    for i in range(9):
        y, state = enc.forward_chunk(xd[i:i + 1], state)
        rows.append(y.cpu().clone())                                # (a replayed step returns the recorded step's output buffer: consume it before the next call)
    assert state.cap >= 9 and state.len == 9
    torch.testing.assert_close(torch.cat(rows), whole.cpu(), rtol=1e-4, atol=1e-4)
    ref, _ = C1.EspnetEncoder(sd, "llm.", CFG.llm_heads, "transformer").forward_chunk(x, None)
    torch.testing.assert_close(whole.cpu(), ref, rtol=1e-4, atol=1e-4)


def test_flow_inference_with_flow_cache_matches_reference(lib):
    g = gold("cv1k_flow")
    flow = build_flow(lib)
    cache = torch.zeros(1, 80, 0, 2)
    for name, n in (("a", 50), ("b", 30)):                         # 50 tokens: head / middle / tail interpolation; "b" runs on "a"'s flow cache
        torch.manual_seed(40 + n)
        feat, cache = flow.inference(token=g["token_" + name], token_len=t(n), prompt_token=g["prompt_token"], prompt_token_len=t(12), prompt_feat=g["prompt_feat"],
                                     prompt_feat_len=t(25), embedding=g["embedding"], flow_cache=cache)
        assert feat.shape == g["feat_" + name].shape == (1, 80, int(n / 50 * 22050 / 256))
        torch.testing.assert_close(feat.cpu(), g["feat_" + name], rtol=1e-3, atol=1e-3)
        torch.testing.assert_close(cache.cpu(), g["cache_" + name], rtol=1e-4, atol=1e-4)


def test_load_takes_reference_state_dict_files(lib, tmp_path):
    """CosyVoiceModel.load(llm.pt, flow.pt, hift.pt) (cli/model.py:65-73) into the kernel-backed stages; an inference_sft-shaped request end to end."""
    torch.save(W.make_cv1_llm(CFG), tmp_path / "llm.pt")
    torch.save(W.make_cv1_flow(CFG), tmp_path / "flow.pt")
    torch.save({"generator." + k: v for k, v in W.make_hift(HCFG).items()}, tmp_path / "hift.pt")
    m = CK.CosyVoiceModel()
    m.load(str(tmp_path / "llm.pt"), str(tmp_path / "flow.pt"), str(tmp_path / "hift.pt"), hift_cfg=HCFG, lib=lib, text_heads=CFG.text_heads, llm_heads=CFG.llm_heads,
           enc_heads=CFG.flow_heads, est_heads=CFG.est_heads)
    m.llm.sampling = greedy
    g = gold("cv1k_llm")
    inf = m.llm.inference
    m.llm.inference = lambda **kw: inf(**dict(kw, max_token_text_ratio=3, min_token_text_ratio=3))
    torch.manual_seed(1)
    out = next(iter(m.tts(text=g["text"], flow_embedding=g["embedding"], llm_embedding=g["embedding"], stream=False)))["tts_speech"]
    assert out.shape == (1, int(21 / 50 * 22050 / 256) * 256) and torch.isfinite(out).all() and float(out.abs().max()) > 0
    assert not m.fp16 and not m.llm.w16 and m.flow.precision == "fp32"
    # fp16=True, the reference's switch for this model (cli/cosyvoice.py:27-56): the LM's matrices as bf16, the estimator in bf16 mode
    h = CK.CosyVoiceModel()
    h.load(str(tmp_path / "llm.pt"), str(tmp_path / "flow.pt"), str(tmp_path / "hift.pt"), hift_cfg=HCFG, lib=lib, fp16=True, text_heads=CFG.text_heads,
           llm_heads=CFG.llm_heads, enc_heads=CFG.flow_heads, est_heads=CFG.est_heads)
    assert h.fp16 and h.llm.w16 and h.flow.precision == "bf16" and h.flow.estimator.precision == "bf16"
    h.llm.sampling = greedy
    inf16 = h.llm.inference
    h.llm.inference = lambda **kw: inf16(**dict(kw, max_token_text_ratio=3, min_token_text_ratio=3))
    torch.manual_seed(1)
    out16 = next(iter(h.tts(text=g["text"], flow_embedding=g["embedding"], llm_embedding=g["embedding"], stream=False)))["tts_speech"]
    assert out16.shape == out.shape and torch.isfinite(out16).all() and float(out16.abs().max()) > 0


def test_split3_weights_option(lib):
    """Kernels(split3=True): every weight GEMM with both operands split on the bf16 matrix pipe (gemm_conv.h WX3) - the same tokens and the same mel to fp32
    rounding as the fp32 MFMA chain."""
    g = gold("cv1k_llm")
    kw = dict(text=g["text"], text_len=t(7), prompt_text=g["prompt_text"], prompt_text_len=t(4), prompt_speech_token=g["prompt_speech_token"],
              prompt_speech_token_len=t(9), embedding=g["embedding"])
    lm = CK.TransformerLM(W.make_cv1_llm(CFG), text_heads=CFG.text_heads, llm_heads=CFG.llm_heads, sampling=greedy, lib=lib, split3=True)
    assert lm.affine.w3 is not None and lm.affine.w3.dtype == torch.bfloat16
    assert list(lm.inference(max_token_text_ratio=6, min_token_text_ratio=2, **kw)) == g["tokens_greedy"].tolist()
    g = gold("cv1k_flow")
    flow = CK.MaskedDiffWithXvec(W.make_cv1_flow(CFG), enc_heads=CFG.flow_heads, est_heads=CFG.est_heads, input_frame_rate=CFG.input_frame_rate, lib=lib, split3=True, estimator="operators")
    torch.manual_seed(90)
    feat, _ = flow.inference(token=g["token_a"], token_len=t(50), prompt_token=g["prompt_token"], prompt_token_len=t(12), prompt_feat=g["prompt_feat"],
                             prompt_feat_len=t(25), embedding=g["embedding"], flow_cache=torch.zeros(1, 80, 0, 2))
    torch.testing.assert_close(feat.cpu(), g["feat_a"], rtol=1e-3, atol=1e-3)


def test_launch_tapes_are_bit_identical_to_eager_sequencing(lib):
    """LaunchTape replays (the estimator of Euler steps 2..n, the LM decode step) against the same launches sequenced from scratch (Kernels.use_tapes = False):
    the same kernels on the same operands, so the mel and the decode rows are equal bit for bit."""
    g = gold("cv1k_flow")
    flow = build_flow(lib, "operators")                         # (the handle sequences its launches in C++: no tape)
    out = {}
    for tapes in (True, False):
        flow.k.use_tapes = tapes
        torch.manual_seed(41)
        feat, cache = flow.inference(token=g["token_b"], token_len=t(30), prompt_token=g["prompt_token"], prompt_token_len=t(12), prompt_feat=g["prompt_feat"],
                                     prompt_feat_len=t(25), embedding=g["embedding"], flow_cache=torch.zeros(1, 80, 0, 2))
        out[tapes] = (feat.cpu().clone(), cache.cpu().clone())
    assert torch.equal(out[True][0], out[False][0]) and torch.equal(out[True][1], out[False][1])
    K = CK.Kernels(lib)
    enc = CK.EspnetEncoder(W.make_cv1_llm(CFG), "llm.", CFG.llm_heads, "transformer", kern=K)
    xd = K.put(torch.randn(7, CFG.llm_dim, generator=torch.Generator().manual_seed(2)))
    rows = {}
    for tapes in (True, False):
        K.use_tapes = tapes
        y, state = enc.forward_chunk(xd[:3], None)                  # a 3-row prompt, then four single steps
        rows[tapes] = [y.cpu().clone()]
        for i in range(3, 7):
            y, state = enc.forward_chunk(xd[i:i + 1], state)
            rows[tapes].append(y.cpu().clone())
        assert (getattr(state, "plan", None) is not None) == tapes
    assert all(torch.equal(a, b) for a, b in zip(rows[True], rows[False]))


def test_concurrent_requests_on_one_stage_object_are_serialised(lib, monkeypatch):
    """Two threads inside one MaskedDiffWithXvec (what two concurrent tts() calls of one CosyVoiceModel do): the stage lock keeps one request's recording from
    swallowing the other's launches - each result equals the request alone.  (The CFM noise comes from the global host RNG, whose draws would depend on the
    interleaving: for this test it is a fixed tensor per request length.)"""
    import threading
    g = gold("cv1k_flow")
    flow = build_flow(lib, "operators")
    gen = torch.Generator().manual_seed(17)
    reqs = {n: torch.randint(0, 40, (1, n), generator=gen, dtype=torch.int32) for n in (24, 30)}
    noise = {25 + int(n / 50 * 22050 / 256): torch.randn(1, 80, 25 + int(n / 50 * 22050 / 256), generator=gen) for n in reqs}

    class _Torch:                                                   # the module's `torch` with randn answering from the table
        def __getattr__(self, name):
            return getattr(torch, name)

        @staticmethod
        def randn(*shape, **kw):
            return noise[shape[-1]].clone()
    monkeypatch.setattr(CK, "torch", _Torch())

    def run(n, out):
        feat, _ = flow.inference(token=reqs[n], token_len=t(n), prompt_token=g["prompt_token"], prompt_token_len=t(12), prompt_feat=g["prompt_feat"],
                                 prompt_feat_len=t(25), embedding=g["embedding"], flow_cache=torch.zeros(1, 80, 0, 2))
        out[n] = feat.cpu().clone()

    alone, both = {}, {}
    for n in reqs:
        run(n, alone)
    ths = [threading.Thread(target=run, args=(n, both)) for n in reqs]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    assert set(both) == set(reqs) and all(torch.equal(alone[n], both[n]) for n in reqs)


def test_fused_decode_step_matches_the_launch_per_operator_step(lib):
    """Round 4: cv_lm1_step (csrc/lm1.hip - the whole decode row as 3 + 5 launches per layer inside one hipGraph) against forward_chunk + decoder sequenced from the
    host (3 + 8 launches per layer).  GEMVs and LayerNorms repeat the operators' arithmetic; the one-query attention sums in another order -> logits to fp32
    rounding, the cache rows it leaves behind likewise; graph replay vs kernel-by-kernel launches of the same step: equal bit for bit.  A cache that grows (and is
    rebound) mid-request, and two requests stepping alternately on one handle."""
    import ctypes as C
    sd = W.make_cv1_llm(CFG)
    lm = CK.TransformerLM(sd, text_heads=CFG.text_heads, llm_heads=CFG.llm_heads, sampling=greedy, lib=lib)
    assert lm.step is not None and lm.fused_step
    K, enc = lm.k, lm.llm
    assert lm.step.stat("launches_per_step") == 3 + 5 * enc.n_layers
    gen = torch.Generator().manual_seed(5)
    xd = K.put(torch.randn(16, enc.embed.k, generator=gen))

    def run(fused, graph=1, cap=None, n_prompt=5):
        lm.step.lib.cv_lm1_set_option(lm.step.h, b"graph", C.c_int32(graph))
        state = CK._KVState(K, enc.n_layers, enc.d, cap) if cap else None
        y, state = enc.forward_chunk(xd[:n_prompt], state)
        out = []
        for i in range(n_prompt, 16):
            if fused:
                out.append(enc.fused_step(lm.step, xd[i:i + 1], state).cpu().clone())
            else:
                y, state = enc.forward_chunk(xd[i:i + 1], state)
                out.append(K.linear(y[-1:], lm.decoder, 1).reshape(-1).cpu().clone())
        return torch.stack(out), [r[:state.len].cpu().clone() for r in state.rows]

    want, rows_w = run(False)
    got, rows_g = run(True)
    torch.testing.assert_close(got, want, rtol=2e-4, atol=2e-4)
    for a, b in zip(rows_g, rows_w):
        torch.testing.assert_close(a, b, rtol=2e-4, atol=2e-4)
    assert lm.step.stat("graph_replays") == 11
    eager, _ = run(True, graph=0)
    assert torch.equal(eager, got)
    small, _ = run(True, cap=6)                                     # the cache doubles twice on the way to 16 rows: rebound each time
    assert torch.equal(small, got)
    # two requests taking turns on the one handle (the stage lock is per step): each sees its own cache
    sa, sb = None, None
    _, sa = enc.forward_chunk(xd[:5], sa)
    _, sb = enc.forward_chunk(xd[:3], sb)
    la, lb = [], []
    for i in range(5, 9):
        la.append(enc.fused_step(lm.step, xd[i:i + 1], sa).cpu().clone())
        lb.append(enc.fused_step(lm.step, xd[i - 2:i - 1], sb).cpu().clone())
    assert torch.equal(torch.stack(la), got[:4])
    alone_b, _ = run(True, n_prompt=3)
    assert torch.equal(torch.stack(lb), alone_b[:4])
    # the generator end to end: same tokens with and without the fused step
    g = gold("cv1k_llm")
    kw = dict(text=g["text"], text_len=t(7), prompt_text=g["prompt_text"], prompt_text_len=t(4), prompt_speech_token=g["prompt_speech_token"],
              prompt_speech_token_len=t(9), embedding=g["embedding"])
    lm.fused_step = False
    slow = list(lm.inference(max_token_text_ratio=6, min_token_text_ratio=2, **kw))
    lm.fused_step = True
    assert list(lm.inference(max_token_text_ratio=6, min_token_text_ratio=2, **kw)) == slow == g["tokens_greedy"].tolist()


def test_device_resident_decode_loop(lib):
    """Round 5: TransformerLM.inference with the loop on the device (cv_lm1_decode_begin / cv_lm1_decode: sampler + embedding row of the sampled token next to the step's 73
    launches; tokens come back per chunk, no logits per token).  sampling="greedy": exactly the tokens of the REAL class's golden (and of the host-sampler loop), with a
    prompt, without one (inference_sft shape), with chunks that do not divide the length and with eos arriving mid-chunk.  sampling="ras" with INJECTED uniform variates:
    the decisions of the reference's rule (oracle.sampling: nucleus prefix, repetition window, full-distribution fallback) replayed on the host from the same variates."""
    from oracle import sampling as OS
    g = gold("cv1k_llm")
    kw = dict(text=g["text"], text_len=t(7), prompt_text=g["prompt_text"], prompt_text_len=t(4), prompt_speech_token=g["prompt_speech_token"],
              prompt_speech_token_len=t(9), embedding=g["embedding"])
    sd = W.make_cv1_llm(CFG)
    for chunk in (64, 5):
        lm = CK.TransformerLM(sd, text_heads=CFG.text_heads, llm_heads=CFG.llm_heads, sampling="greedy", lib=lib, decode_chunk=chunk)
        assert lm.fused_step
        assert list(lm.inference(max_token_text_ratio=6, min_token_text_ratio=2, **kw)) == g["tokens_greedy"].tolist()
        e0 = torch.zeros(1, 0, dtype=torch.int32)
        sft = dict(kw, prompt_text=e0, prompt_text_len=t(0), prompt_speech_token=e0, prompt_speech_token_len=t(0))
        assert list(lm.inference(max_token_text_ratio=5, min_token_text_ratio=2, **sft)) == g["tokens_sft"].tolist()
        assert lm.step_stat("steps") > 0 and len(lm._loop_steps) == 1      # (a loop handle of the request's own, returned to the pool)
    # repetition-aware sampling from fixed uniforms: device loop == host loop driven by the oracle's restatement of the rule with the same variates
    u = torch.rand(2 * 64, generator=torch.Generator().manual_seed(21))
    lm = CK.TransformerLM(sd, text_heads=CFG.text_heads, llm_heads=CFG.llm_heads, sampling="ras", lib=lib, decode_chunk=7)
    lm._uniforms = u
    got = list(lm.inference(max_token_text_ratio=6, min_token_text_ratio=2, **kw))
    step = [0]

    def host_ras(scores, decoded, sampling):
        i = step[0]; step[0] += 1
        return OS.ras_sampling(scores.clone(), decoded, sampling, u=(float(u[2 * i]), float(u[2 * i + 1])))
    ref = CK.TransformerLM(sd, text_heads=CFG.text_heads, llm_heads=CFG.llm_heads, sampling=host_ras, lib=lib)
    want = list(ref.inference(max_token_text_ratio=6, min_token_text_ratio=2, **kw))
    assert got == want and 14 <= len(got) <= 42 and len(set(got)) > 3


def test_interleaved_device_loops_do_not_share_loop_state(lib):
    """the device-resident loop's state lives in a cv_lm1 handle and the stage lock is released between chunks - two requests whose generators are advanced
    alternately (chunks of 3 tokens) must each get exactly the tokens they get alone, with a host-sampler request stepping in between as well; the handles go back
    to the pool when a generator ends or is closed early."""
    g = gold("cv1k_llm")
    kw = dict(text=g["text"], text_len=t(7), prompt_text=g["prompt_text"], prompt_text_len=t(4), prompt_speech_token=g["prompt_speech_token"],
              prompt_speech_token_len=t(9), embedding=g["embedding"])
    e0 = torch.zeros(1, 0, dtype=torch.int32)
    sft = dict(kw, prompt_text=e0, prompt_text_len=t(0), prompt_speech_token=e0, prompt_speech_token_len=t(0))
    sd = W.make_cv1_llm(CFG)
    lm = CK.TransformerLM(sd, text_heads=CFG.text_heads, llm_heads=CFG.llm_heads, sampling="greedy", lib=lib, decode_chunk=3)
    a = lm.inference(max_token_text_ratio=6, min_token_text_ratio=2, **kw)
    b = lm.inference(max_token_text_ratio=5, min_token_text_ratio=2, **sft)
    got_a, got_b, live = [], [], [a, b]
    while live:
        for gen, out in ((a, got_a), (b, got_b)):
            if gen in live:
                try:
                    for _ in range(2):                              # two tokens at a time: the hand-over falls inside and between the 3-token chunks
                        out.append(next(gen))
                except StopIteration:
                    live.remove(gen)
    assert got_a == g["tokens_greedy"].tolist() and got_b == g["tokens_sft"].tolist()
    assert len(lm._loop_steps) == 2 and all(s.bound is None for s in lm._loop_steps)
    # an abandoned request returns its handle; a host-sampler model stepping between the chunks of a device loop does not disturb it (its own handle, rebound per call)
    c = lm.inference(max_token_text_ratio=6, min_token_text_ratio=2, **kw)
    first = [next(c), next(c)]
    c.close()
    assert first == g["tokens_greedy"].tolist()[:2] and len(lm._loop_steps) == 2
    d = lm.inference(max_token_text_ratio=6, min_token_text_ratio=2, **kw)
    got_d = [next(d)]
    lm.sampling, keep = greedy, lm.sampling                          # a python sampler: the operator-per-token path on lm.step, in the gap of d's open loop
    assert list(lm.inference(max_token_text_ratio=5, min_token_text_ratio=2, **sft)) == g["tokens_sft"].tolist()
    lm.sampling = keep
    got_d += list(d)
    assert got_d == g["tokens_greedy"].tolist()
