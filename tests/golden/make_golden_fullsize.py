"""Golden vectors at the FULL model dimensions, made by the REAL reference classes (/root/reference) - build container only:

    python tests/golden/make_golden_fullsize.py [llm] [llm_ras] [llm_cv3] [cv1_llm] [flow] [hift] [dit] [causal_hift] [cv1_flow] [cv1_hift] [model] [model_cv3] [model_cv1] [mixed64]   (no argument: all but mixed64, ~8 min on 8 cores)

The other generators (make_golden.py, make_golden_cv1.py) run the reference at test dimensions, which pins the oracle's ARITHMETIC; the full-size parity
tests and every bench run then compare the kernels with the oracle's own full-size output (tests/golden/u10_oracle_tokens.json, cv3_u10_oracle_tokens.json,
oracle.flow / oracle.hift computed on the spot).  This script closes that last link: the same real classes, loaded (strict=True) with the seeded full-size state
dicts of cosyvoice_amd.synthetic, run the BENCHMARK requests themselves -

  llm      cosyvoice.llm.llm.Qwen2LM.inference (llm/llm.py:458-549) at CosyVoice2-0.5B dimensions on U10: all 250 greedy ids + the first log-prob rows
  llm_ras  the same class with the REAL ras_sampling on U10, its multinomial draws replaced by the stored variates of u10_ras_oracle_tokens.json: all 250 ids
  mixed64  the same class on the 64 utterances of bench.py's mixed64 workload (configs[3]): every id (not in the default list: ~15 min)
  llm_cv3  CosyVoice3LM.inference (llm/llm.py:664-706) at Fun-CosyVoice3-0.5B dimensions on bench.py's instruct request: all 250 ids
  cv1_llm  TransformerLM.inference (llm/llm.py:162-223) at CosyVoice-300M dimensions on bench.py's inference_sft request: all 500 ids
  flow     CausalMaskedDiffWithXvec.inference (flow/flow.py:235-281; estimator over the restated Matcha blocks of matcha_stub.py) on U10's 250 tokens: the
           mel [80, 500]; the estimator boundary at T = 674 (offline and streaming masks); the encoder at 337 tokens
  hift     HiFTGenerator.inference (hifigan/generator.py:557-569) at 24 kHz dimensions on 100 frames of that mel
  dit, causal_hift   CausalMaskedDiffWithDiT / DiT and CausalHiFTGenerator at Fun-CosyVoice3-0.5B dimensions (row a17)
  model    cli.model.CosyVoice2Model.tts on U10 with its default streaming settings, offline and streamed (scripted LLM = the 250 ids above): chunk lengths + waveforms
  model_cv3  cli.model.CosyVoice3Model.tts on the cosyvoice3 bench request, offline (its replay is an opt-in test: CV_TEST_FULL=1)
  model_cv1  cli.model.CosyVoiceModel.tts on the CosyVoice-300M bench request, offline
  cv1_flow, cv1_hift MaskedDiffWithXvec (U-Net ConditionalDecoder, flow cache) and the 22.05 kHz HiFTGenerator at CosyVoice-300M dimensions (rows a18 / f4)

and tests/test_fullsize_pinned.py (CPU, `-m "not gpu"`) holds the oracle - and the committed oracle token files the GPU runs are checked against - to them.
Weights are not stored (the factory regenerates them from the seed); the HiFT noise is the global torch RNG seeded right before the call, replayed by the test.
"""
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)
_argv, sys.argv = sys.argv, sys.argv[:1]                # make_golden_cv1 reads its mode from argv at import
import make_golden as MG  # noqa: E402  (installs the reference import stubs)
import make_golden_cv1 as MG1  # noqa: E402
sys.argv = _argv
from cosyvoice_amd import configs as CF, synthetic as W  # noqa: E402

N_GEN, N_TEXT, N_PROMPT_TEXT, N_PROMPT_TOK = 250, 30, 12, 87
HIFT_FRAMES = 100


def save(name, **arrs):
    out = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()}
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote %s: %s, %.0f KB" % (name, {k: v.shape for k, v in out.items()}, os.path.getsize(path) / 1024))


def _qwen_lm(cfg, cls_name):
    """The real Qwen2LM / CosyVoice3LM over a random-init Qwen2ForCausalLM of the configured size (the wrapper of make_golden.golden_llm: from_pretrained needs a
    checkpoint directory; the decode mask follows the semantics the pinned transformers 4.51.3 gives the reference's all-ones mask)."""
    from transformers import Qwen2Config, Qwen2ForCausalLM
    import cosyvoice.llm.llm as L

    class Enc(L.Qwen2Encoder):
        def __init__(self):
            torch.nn.Module.__init__(self)
            hc = Qwen2Config(vocab_size=cfg.text_vocab, hidden_size=cfg.hidden, intermediate_size=cfg.inter, num_hidden_layers=cfg.layers,
                             num_attention_heads=cfg.heads, num_key_value_heads=cfg.kv_heads, max_position_embeddings=4096, rms_norm_eps=cfg.rms_eps,
                             rope_theta=cfg.rope_theta, tie_word_embeddings=True, attention_dropout=0.0)
            self.model = Qwen2ForCausalLM(hc)

        def forward_one_step(self, xs, masks, cache=None):
            outs = self.model(inputs_embeds=xs, attention_mask=None if xs.shape[1] == 1 else masks[:, -1, :], output_hidden_states=True, return_dict=True,
                              use_cache=True, past_key_values=cache)
            return outs.hidden_states[-1], outs.past_key_values

    logps = []

    def greedy(scores, decoded, k):
        logps.append(scores.clone())
        return int(scores.argmax().item())

    lm = getattr(L, cls_name)(cfg.hidden, cfg.hidden, cfg.speech_token_size, Enc(), greedy)
    lm.load_state_dict(W.make_llm(cfg), strict=True)
    return lm.eval(), logps


def _margins(logps, mask_id):
    out = []
    for lp in logps:
        lp = lp.clone()
        lp[mask_id] = -float("inf")
        top2 = torch.topk(lp, 2).values
        out.append(float(top2[0] - top2[1]))
    return np.array(out, dtype=np.float32)


def _run_lm(lm, logps, u, cfg, n_prompt_tok, N_GEN=N_GEN):
    t = lambda n: torch.tensor([n], dtype=torch.int32)
    # The reference computes its length bounds as int(int32 TENSOR * python float) = float32 arithmetic (llm/llm.py:497-498): 30 * (250 / 30) is 249.99998 there and 249
    # tokens come out, where python's double arithmetic (the oracle, the product: DESIGN.md section 4) gives 250.  (N + 0.5) / n_text yields N under both rules.
    ratio = (N_GEN + 0.5) / N_TEXT
    assert int(t(N_TEXT) * ratio) == N_GEN == int(N_TEXT * ratio)
    t0 = time.time()
    toks = list(lm.inference(text=u["text"], text_len=t(u["text"].shape[1]), prompt_text=u["prompt_text"], prompt_text_len=t(u["prompt_text"].shape[1]),
                             prompt_speech_token=u["llm_prompt_speech_token"], prompt_speech_token_len=t(n_prompt_tok), embedding=u["llm_embedding"],
                             max_token_text_ratio=ratio, min_token_text_ratio=ratio))
    print("  %d tokens from the real class in %.0f s" % (len(toks), time.time() - t0))
    assert len(toks) == N_GEN
    return toks


def golden_llm():
    lc, fc, _ = W.cv2()
    lm, logps = _qwen_lm(lc, "Qwen2LM")
    u = W.synthetic_utterance(lc, fc, n_prompt_tok=N_PROMPT_TOK, n_prompt_text=N_PROMPT_TEXT, n_text=N_TEXT)
    toks = _run_lm(lm, logps, u, lc, N_PROMPT_TOK)
    # (the reference logs a row AFTER sampling_ids masked index speech_token_size in place: that column is -inf in the stored rows)
    save("fullsize_llm", tokens=np.array(toks, dtype=np.int32), logp=torch.stack([logps[i] for i in (0, 1, 2, 249)]), logp_steps=np.array([0, 1, 2, 249]),
         top2_margin=_margins(logps, lc.speech_token_size))


def golden_mixed64():
    """The 64 utterances of bench.py's mixed64 workload (BASELINE.json configs[3]: seeds 4000 + i, 125 / 250 / 375 / 500 tokens in equal mix; make_mixed64.py holds the
    oracle's ids): every id from the real Qwen2LM (about a quarter of an hour of CPU)."""
    lc, fc, _ = W.cv2()
    lm, logps = _qwen_lm(lc, "Qwen2LM")
    out = {}
    for i in range(64):
        n_gen = (125, 250, 375, 500)[i % 4]
        u = W.synthetic_utterance(lc, fc, n_prompt_tok=N_PROMPT_TOK, n_prompt_text=N_PROMPT_TEXT, n_text=N_TEXT, seed=4000 + i)
        del logps[:]
        out["tokens_%02d" % i] = np.array(_run_lm(lm, logps, u, lc, N_PROMPT_TOK, n_gen), dtype=np.int16)
        out["min_margin_%02d" % i] = _margins(logps, lc.speech_token_size).min()
    save("fullsize_mixed64", **out)


def golden_llm_ras():
    """U10 decoded by the real Qwen2LM with the REAL repetition-aware sampler (cosyvoice/utils/common.py:138-167: ras_sampling -> nucleus_sampling / random_sampling)
    at full size.  Its draws (`Tensor.multinomial` on the global RNG, which nothing else can reproduce) are replaced by an inverse CDF on the variates stored in
    u10_ras_oracle_tokens.json - two per step, nucleus draw then fallback draw - i.e. the decision logic runs from the reference on the reference's own probabilities;
    bench.py replays the same variates on the device (`self_check.ras`)."""
    from cosyvoice.utils.common import ras_sampling
    lc, fc, _ = W.cv2()
    lm, _ = _qwen_lm(lc, "Qwen2LM")
    gold = json.load(open(os.path.join(HERE, "u10_ras_oracle_tokens.json")))
    us, cur, step, n_fallback = gold["variates"], [], [0], [0]

    def fake_multinomial(self, n, replacement=False):
        u = cur.pop(0)
        cdf = torch.cumsum(self.double() / self.double().sum(), 0)
        return torch.searchsorted(cdf, torch.tensor([u], dtype=torch.float64), right=True).clamp(max=self.numel() - 1)

    def sampler(scores, decoded, sampling):
        cur[:] = [us[2 * step[0]], us[2 * step[0] + 1]]
        step[0] += 1
        t = ras_sampling(scores, decoded, sampling)
        n_fallback[0] += int(len(cur) == 0)                    # both variates used: the repetition fallback drew
        return t
    lm.sampling = sampler
    orig = torch.Tensor.multinomial
    torch.Tensor.multinomial = fake_multinomial
    try:
        u = W.synthetic_utterance(lc, fc, n_prompt_tok=N_PROMPT_TOK, n_prompt_text=N_PROMPT_TEXT, n_text=N_TEXT)
        toks = _run_lm(lm, [], u, lc, N_PROMPT_TOK)
    finally:
        torch.Tensor.multinomial = orig
    save("fullsize_llm_ras", tokens=np.array(toks, dtype=np.int32), fallback_draws=np.array(n_fallback[0]), distinct=np.array(len(set(toks))))


def golden_llm_cv3():
    lc, fc = CF.cv3_llm(), CF.cv3_flow()
    lm, logps = _qwen_lm(lc, "CosyVoice3LM")
    u = W.synthetic_utterance(lc, fc, n_prompt_tok=N_PROMPT_TOK, n_prompt_text=24, n_text=N_TEXT, seed=2025)      # bench.py cv3_workload / make_cv3_u10.py
    u["prompt_text"][0, 11] = lc.endofprompt_id
    u["llm_prompt_speech_token"] = torch.zeros(1, 0, dtype=torch.int32)
    toks = _run_lm(lm, logps, u, lc, 0)
    save("fullsize_llm_cv3", tokens=np.array(toks, dtype=np.int32), logp=torch.stack([logps[i] for i in (0, 1, 249)]), logp_steps=np.array([0, 1, 249]),
         top2_margin=_margins(logps, lc.speech_token_size))


def golden_cv1_llm():
    """bench.py cv1_workload's request (seed 300: 25 text ids, a speaker embedding, no prompts; the length forced to 500 tokens, greedy)."""
    cfg, hcfg = W.cv1()
    MG1.CFG, MG1.HCFG = cfg, hcfg
    logps = []

    def greedy(scores, decoded, sampling):
        logps.append(scores.clone())
        return int(scores.argmax().item())
    m = MG1.build_llm(greedy)
    g = torch.Generator().manual_seed(300)
    n_text, n_gen = 25, 500
    text = torch.randint(0, cfg.text_vocab, (1, n_text), generator=g, dtype=torch.int32)
    emb = torch.randn(1, cfg.spk_dim, generator=g)
    e0 = torch.zeros(1, 0, dtype=torch.int32)
    tl = lambda n: torch.tensor([n], dtype=torch.int32)
    t0 = time.time()
    toks = list(m.inference(text=text, text_len=tl(n_text), prompt_text=e0, prompt_text_len=tl(0), prompt_speech_token=e0, prompt_speech_token_len=tl(0), embedding=emb,
                            max_token_text_ratio=n_gen / n_text, min_token_text_ratio=n_gen / n_text))
    print("  %d tokens from the real TransformerLM in %.0f s" % (len(toks), time.time() - t0))
    assert len(toks) == n_gen
    save("fullsize_cv1_llm", text=text, embedding=emb, tokens=np.array(toks, dtype=np.int32), logp=torch.stack([logps[i] for i in (0, 1, 499)]), logp_steps=np.array([0, 1, 499]),
         top2_margin=_margins(logps, cfg.speech_token_size))


def golden_flow():
    lc, fc, _ = W.cv2()
    flow = MG.build_ref_flow(fc)
    u = W.synthetic_utterance(lc, fc, n_prompt_tok=N_PROMPT_TOK, n_prompt_text=N_PROMPT_TEXT, n_text=N_TEXT)
    tokens = json.load(open(os.path.join(HERE, "u10_oracle_tokens.json")))["tokens"]        # (== the real class's: fullsize_llm.npz, checked by the test)
    token = torch.tensor(tokens, dtype=torch.int32).unsqueeze(0)
    t = lambda n: torch.tensor([n], dtype=torch.int32)
    common = dict(prompt_token=u["flow_prompt_speech_token"], prompt_token_len=t(N_PROMPT_TOK), prompt_feat=u["prompt_speech_feat"], prompt_feat_len=t(2 * N_PROMPT_TOK),
                  embedding=u["flow_embedding"])
    t0 = time.time()
    mel_full, _ = flow.inference(token=token, token_len=t(N_GEN), streaming=False, finalize=True, **common)
    # the first streamed chunk of the request (cli/model.py:345-351: hop 25 + prompt pad 13 + pre_lookahead 3 tokens)
    n1 = 25 + 13 + 3
    mel_chunk, _ = flow.inference(token=token[:, :n1], token_len=t(n1), streaming=True, finalize=False, **common)
    print("  flow.inference x2 from the real class in %.0f s" % (time.time() - t0))
    g = torch.Generator().manual_seed(12)
    T = 2 * (N_PROMPT_TOK + N_GEN)
    x = torch.randn(2, 80, T, generator=g); mu = torch.randn(2, 80, T, generator=g); cond = torch.randn(2, 80, T, generator=g)
    spk = torch.randn(2, 80, generator=g); tt = torch.tensor([0.25, 0.25]); mask = torch.ones(2, 1, T)
    with torch.inference_mode():
        e_full = flow.decoder.estimator(x, mask, mu, tt, spk, cond, streaming=False)
        e_stream = flow.decoder.estimator(x, mask, mu, tt, spk, cond, streaming=True)
        tok_emb = flow.input_embedding(torch.cat([u["flow_prompt_speech_token"], token], 1).long())
        h_full, _ = flow.encoder(tok_emb, t(N_PROMPT_TOK + N_GEN), streaming=False)
    # the estimator inputs are regenerated by the test from the seed; of the outputs, row 0 in full and a strided slice of row 1 keep the file small
    save("fullsize_flow", mel_full=mel_full[0], mel_chunk=mel_chunk[0], est_full=e_full[0], est_full_row1=e_full[1, :, ::8], est_stream=e_stream[0],
         est_stream_row1=e_stream[1, :, ::8], enc_full=h_full[0, ::4])


def golden_hift():
    _, _, hc = W.cv2()
    hift = MG.build_ref_hift(hc)
    mel = torch.from_numpy(np.load(os.path.join(HERE, "fullsize_flow.npz"))["mel_full"])[None, :, 200:200 + HIFT_FRAMES].contiguous()
    with torch.inference_mode():
        f0 = hift.f0_predictor(mel)
        torch.manual_seed(99)
        speech, source = hift.inference(speech_feat=mel)
    # the test replays the draws (generator.py:245, :312): manual_seed(99); rand(1, 9) with column 0 zeroed; randn_like of the [1, 480 m, 9] transposed view
    save("fullsize_hift", mel=mel[0], f0=f0, speech=speech, source=source)


def golden_dit():
    """CausalMaskedDiffWithDiT.inference (flow/flow.py:369-414) + DiT (flow/DiT/dit.py:145-176; x_transformers' rotary restated in xtransformers_stub.py) at
    Fun-CosyVoice3-0.5B dimensions on bench.py's cosyvoice3 request (250 tokens of cv3_u10_oracle_tokens.json, flow prompt 87 tokens): the estimator boundary at
    T = 674 in both mask modes, and inference with 2 Euler steps (the reference hard-codes 10, flow/flow.py:409; the step count goes through the CFM's own
    n_timesteps argument as in make_golden.golden_dit - two steps keep the CPU test that replays this short)."""
    lc, fc = CF.cv3_llm(), CF.cv3_flow()
    flow = MG.build_ref_dit_flow(fc)
    u = W.synthetic_utterance(lc, fc, n_prompt_tok=N_PROMPT_TOK, n_prompt_text=24, n_text=N_TEXT, seed=2025)
    token = torch.tensor(json.load(open(os.path.join(HERE, "cv3_u10_oracle_tokens.json")))["tokens"], dtype=torch.int32).unsqueeze(0)
    t = lambda n: torch.tensor([n], dtype=torch.int32)
    orig_fwd = type(flow.decoder).forward

    def fwd(self, mu, mask, spks, cond, n_timesteps=10, **kw):
        return orig_fwd(self, mu=mu, mask=mask, spks=spks, cond=cond, n_timesteps=2, **kw)
    type(flow.decoder).forward = fwd
    t0 = time.time()
    try:
        mel, _ = flow.inference(token=token, token_len=t(N_GEN), prompt_token=u["flow_prompt_speech_token"], prompt_token_len=t(N_PROMPT_TOK),
                                prompt_feat=u["prompt_speech_feat"], prompt_feat_len=t(2 * N_PROMPT_TOK), embedding=u["flow_embedding"], streaming=False, finalize=True)
    finally:
        type(flow.decoder).forward = orig_fwd
    g = torch.Generator().manual_seed(14)
    T = 2 * (N_PROMPT_TOK + N_GEN)
    x = torch.randn(2, 80, T, generator=g); mu = torch.randn(2, 80, T, generator=g); cond = torch.randn(2, 80, T, generator=g)
    spk = torch.randn(2, 80, generator=g); tt = torch.tensor([0.25, 0.25]); mask = torch.ones(2, 1, T)
    with torch.inference_mode():
        e_full = flow.decoder.estimator(x, mask, mu, tt, spk, cond, streaming=False)
        e_stream = flow.decoder.estimator(x, mask, mu, tt, spk, cond, streaming=True)
    print("  DiT flow: inference (2 steps) + 2 estimator calls from the real class in %.0f s" % (time.time() - t0))
    save("fullsize_dit", mel_2steps=mel[0], est_full=e_full[0, :, ::2], est_full_row1=e_full[1, :, ::8], est_stream=e_stream[0, :, ::2], est_stream_row1=e_stream[1, :, ::8])


def golden_causal_hift():
    """CausalHiFTGenerator.inference (hifigan/generator.py:572-726; float64 f0 predictor, :716-717) at Fun-CosyVoice3-0.5B dimensions on 100 mel frames, one-shot and
    as a non-final chunk.  The generator's fixed noise buffers are construction-time draws (generator.py:223-226): they are set here from a seeded generator the test
    replays."""
    hc = CF.cv3_hift()
    hift = MG.build_ref_causal_hift(hc)
    g = torch.Generator().manual_seed(16)
    sg = hift.m_source.l_sin_gen
    sg.rand_ini = torch.rand(1, 9, generator=g); sg.rand_ini[:, 0] = 0
    sg.sine_waves = torch.rand(1, 480 * HIFT_FRAMES, 9, generator=g)
    mel = torch.from_numpy(np.load(os.path.join(HERE, "fullsize_dit.npz"))["mel_2steps"])[None, :, 150:150 + HIFT_FRAMES].contiguous()
    with torch.inference_mode():
        speech, source = hift.inference(speech_feat=mel, finalize=True)
        speech_c, source_c = hift.inference(speech_feat=mel[:, :, :60], finalize=False)
        f0 = hift.f0_predictor(mel.to(torch.float64), finalize=True).float()
    save("fullsize_causal_hift", mel=mel[0], f0=f0, speech=speech, source=source, speech_c=speech_c, source_c=source_c)


def golden_cv1_flow():
    """MaskedDiffWithXvec.inference (flow/flow.py:102-146: InterpolateRegulator, ConditionalCFM with its flow cache, the U-Net ConditionalDecoder flow/decoder.py:88-291 over
    the restated Matcha blocks) at CosyVoice-300M dimensions on bench.py's inference_sft request: the 500 ids of fullsize_cv1_llm.npz, no prompt.  The CFM noise is the
    global torch RNG (flow_matching.py:50), seeded right before the call."""
    cfg, hcfg = W.cv1()
    MG1.CFG, MG1.HCFG = cfg, hcfg
    flow = MG1.build_flow()
    g = np.load(os.path.join(HERE, "fullsize_cv1_llm.npz"))
    token, emb = torch.from_numpy(g["tokens"]).to(torch.int32).unsqueeze(0), torch.from_numpy(g["embedding"])
    t = lambda n: torch.tensor([n], dtype=torch.int32)
    e0 = torch.zeros(1, 0, dtype=torch.int32)
    t0 = time.time()
    torch.manual_seed(41)
    feat, cache = flow.inference(token=token, token_len=t(token.shape[1]), prompt_token=e0, prompt_token_len=t(0), prompt_feat=torch.zeros(1, 0, 80), prompt_feat_len=t(0),
                                 embedding=emb, flow_cache=torch.zeros(1, 80, 0, 2))
    print("  MaskedDiffWithXvec.inference from the real class in %.0f s" % (time.time() - t0))
    save("fullsize_cv1_flow", feat=feat[0], cache_tail=cache[0, :, -8:])


def golden_cv1_hift():
    """HiFTGenerator.inference at 22.05 kHz dimensions (upsample_rates [8, 8], SineGen type 1) on 100 frames of that mel; global RNG seeded before the call."""
    cfg, hcfg = W.cv1()
    MG1.CFG, MG1.HCFG = cfg, hcfg
    h = MG1.build_hift()
    feat = torch.from_numpy(np.load(os.path.join(HERE, "fullsize_cv1_flow.npz"))["feat"])[None, :, 300:300 + HIFT_FRAMES].contiguous()
    torch.manual_seed(77)
    speech, source = h.inference(speech_feat=feat)
    with torch.inference_mode():
        f0 = h.f0_predictor(feat)
    save("fullsize_cv1_hift", feat=feat[0], f0=f0, speech=speech, source=source)


def golden_model():
    """The whole benchmark request through the REAL cosyvoice.cli.model.CosyVoice2Model.tts (cli/model.py:328-394: token2wav, the streaming loop with its default hops
    25 -> 50 -> 100 and the 13-token prompt pad, mel / source / speech caches, fade_in_out) around the real full-size flow + HiFT, offline and streaming, with a scripted LLM
    that yields U10's 250 ids (= the real Qwen2LM's, fullsize_llm.npz).  SineGen2's additive noise (generator.py:312) is zeros for the run, the convention of every
    model-level comparison here (make_golden.golden_model).  Stored: the chunk lengths and every 8th sample of the two waveforms."""
    import cosyvoice.cli.model as M
    lc, fc, hc = W.cv2()
    flow, hift = MG.build_ref_flow(fc), MG.build_ref_hift(hc)
    u = W.synthetic_utterance(lc, fc, n_prompt_tok=N_PROMPT_TOK, n_prompt_text=N_PROMPT_TEXT, n_text=N_TEXT)
    tokens = [int(t) for t in np.load(os.path.join(HERE, "fullsize_llm.npz"))["tokens"]]

    class ScriptedLLM:
        def inference(self, **kw):
            yield from tokens
    orig_randn_like = torch.randn_like
    torch.randn_like = lambda t, **kw: torch.zeros_like(t)
    M.time.sleep = lambda s: None
    out = {}
    try:
        for stream in (False, True):
            t0 = time.time()
            m = M.CosyVoice2Model(ScriptedLLM(), flow, hift)
            with torch.inference_mode():
                chunks = [o["tts_speech"] for o in m.tts(text=u["text"], flow_embedding=u["flow_embedding"], llm_embedding=u["llm_embedding"], prompt_text=u["prompt_text"],
                                                        llm_prompt_speech_token=u["llm_prompt_speech_token"], flow_prompt_speech_token=u["flow_prompt_speech_token"],
                                                        prompt_speech_feat=u["prompt_speech_feat"], stream=stream)]
            key = "stream" if stream else "offline"
            out[key + "_n"] = np.array([c.shape[1] for c in chunks])
            out[key] = torch.cat(chunks, 1)[:, ::8]
            print("  tts(stream=%s) from the real class in %.0f s: chunks %s" % (stream, time.time() - t0, out[key + "_n"].tolist()))
    finally:
        torch.randn_like = orig_randn_like
    save("fullsize_model", **out)


def golden_model_cv3():
    """bench.py's cosyvoice3 request through the REAL cli.model.CosyVoice3Model.tts (cli/model.py:397-450: silent-token filter, accumulating mel cache, speech offsets)
    around the real full-size DiT flow (its hard-coded 10 Euler steps, flow/flow.py:409) + CausalHiFTGenerator, offline; scripted LLM = the 250 ids of the real CosyVoice3LM
    (fullsize_llm_cv3.npz).  The generator's fixed noise buffer is zeros (the convention of make_golden.golden_model_cv3).  The test that replays it costs minutes of
    single-threaded DiT arithmetic and is opt-in (CV_TEST_FULL=1)."""
    import cosyvoice.cli.model as M
    lc, fc, hc = CF.cv3_llm(), CF.cv3_flow(), CF.cv3_hift()
    flow, hift = MG.build_ref_dit_flow(fc), MG.build_ref_causal_hift(hc)
    hift.m_source.l_sin_gen.sine_waves = torch.zeros(1, 480 * 2 * N_GEN, 9)    # (build_ref_causal_hift shrinks the 300 s buffers to 2 s for its fixtures)
    u = W.synthetic_utterance(lc, fc, n_prompt_tok=N_PROMPT_TOK, n_prompt_text=24, n_text=N_TEXT, seed=2025)
    tokens = [int(t) for t in np.load(os.path.join(HERE, "fullsize_llm_cv3.npz"))["tokens"]]

    class ScriptedLLM:
        def inference(self, **kw):
            yield from tokens
    M.time.sleep = lambda s: None
    t0 = time.time()
    m = M.CosyVoice3Model(ScriptedLLM(), flow, hift)
    with torch.inference_mode():
        chunks = [o["tts_speech"] for o in m.tts(text=u["text"], flow_embedding=u["flow_embedding"], llm_embedding=u["llm_embedding"], prompt_text=u["prompt_text"],
                                                llm_prompt_speech_token=torch.zeros(1, 0, dtype=torch.int32), flow_prompt_speech_token=u["flow_prompt_speech_token"],
                                                prompt_speech_feat=u["prompt_speech_feat"], stream=False)]
    print("  CosyVoice3Model.tts offline from the real class in %.0f s: %s samples" % (time.time() - t0, [c.shape[1] for c in chunks]))
    save("fullsize_model_cv3", offline_n=np.array([c.shape[1] for c in chunks]), offline=torch.cat(chunks, 1)[:, ::8])


def golden_model_cv1():
    """bench.py's CosyVoice-300M request through the REAL cli.model.CosyVoiceModel.tts (cli/model.py:135-242) around the real full-size MaskedDiffWithXvec + 22.05 kHz
    HiFTGenerator, offline; scripted LLM = the 500 ids of the real TransformerLM (fullsize_cv1_llm.npz); the global RNG (CFM noise, HiFT noise) seeded before the call."""
    import cosyvoice.cli.model as M
    cfg, hcfg = W.cv1()
    MG1.CFG, MG1.HCFG = cfg, hcfg
    flow, hift = MG1.build_flow(), MG1.build_hift()
    g = np.load(os.path.join(HERE, "fullsize_cv1_llm.npz"))
    tokens, emb = [int(t) for t in g["tokens"]], torch.from_numpy(g["embedding"])

    class ScriptedLLM:
        def inference(self, **kw):
            yield from tokens
    M.time.sleep = lambda s: None
    m = M.CosyVoiceModel(ScriptedLLM(), flow, hift)
    torch.manual_seed(55)
    t0 = time.time()
    with torch.inference_mode():
        chunks = [o["tts_speech"] for o in m.tts(text=torch.from_numpy(g["text"]), flow_embedding=emb, llm_embedding=emb, stream=False)]
    print("  CosyVoiceModel.tts offline from the real class in %.0f s: %s samples" % (time.time() - t0, [c.shape[1] for c in chunks]))
    save("fullsize_model_cv1", offline_n=np.array([c.shape[1] for c in chunks]), offline=torch.cat(chunks, 1)[:, ::8])


if __name__ == "__main__":
    for w in ([a for a in sys.argv[1:] if not a.startswith("--")] or ["llm", "llm_ras", "llm_cv3", "cv1_llm", "flow", "hift", "dit", "causal_hift", "cv1_flow", "cv1_hift", "model", "model_cv3", "model_cv1"]):
        print(w)
        globals()["golden_" + w]()
