"""Generate tests/golden/*.npz by running the REAL reference (/root/reference) on seeded inputs.

Run in the build container only:  python tests/golden/make_golden.py
The reference modules are instantiated with small dims, loaded (strict=True) with cosyvoice_amd.synthetic's seeded state dicts —
which also validates the factory's key names/shapes against the reference — and their outputs are stored.  Weights are
NOT stored: cosyvoice_amd/synthetic.py regenerates them from the seed (numpy PCG64).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)
import ref_import  # noqa: E402

ref_import.install()
from cosyvoice_amd import synthetic as W  # noqa: E402


def save(name, **arrs):
    out = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()}
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, {k: v.shape for k, v in out.items()})


# ------------------------------------------------------------------------------------------------ LLM
def golden_llm():
    from transformers import Qwen2Config, Qwen2ForCausalLM
    import cosyvoice.llm.llm as L
    from cosyvoice.utils.common import ras_sampling

    cfg = W.tiny()[0]

    class Enc(L.Qwen2Encoder):                          # same forward code, random-init backbone instead of from_pretrained
        def __init__(self):
            torch.nn.Module.__init__(self)
            hc = Qwen2Config(vocab_size=cfg.text_vocab, hidden_size=cfg.hidden, intermediate_size=cfg.inter,
                             num_hidden_layers=cfg.layers, num_attention_heads=cfg.heads, num_key_value_heads=cfg.kv_heads,
                             max_position_embeddings=4096, rms_norm_eps=cfg.rms_eps, rope_theta=cfg.rope_theta,
                             tie_word_embeddings=True, attention_dropout=0.0)
            self.model = Qwen2ForCausalLM(hc)

        def forward_one_step(self, xs, masks, cache=None):
            # SURVEY.md §0 oracle trap: the reference's [1,1] decode mask is mis-handled by transformers 5.x; the intended
            # semantics (transformers 4.51.3 drops an all-ones mask) are plain causal attention over the cache.
            outs = self.model(inputs_embeds=xs, attention_mask=None if xs.shape[1] == 1 else masks[:, -1, :],
                              output_hidden_states=True, return_dict=True, use_cache=True, past_key_values=cache)
            return outs.hidden_states[-1], outs.past_key_values

    sd = W.make_llm(cfg)
    logps = []

    def greedy(scores, decoded, k):
        logps.append(scores.clone())
        return int(scores.argmax().item())

    lm = L.Qwen2LM(cfg.hidden, cfg.hidden, cfg.speech_token_size, Enc(), greedy)
    missing = lm.load_state_dict(sd, strict=True)
    lm.eval()
    u = W.synthetic_utterance(cfg, W.tiny()[1], n_prompt_tok=11, n_prompt_text=5, n_text=6)
    kw = dict(text=u["text"], text_len=torch.tensor([u["text"].shape[1]], dtype=torch.int32),
              prompt_text=u["prompt_text"], prompt_text_len=torch.tensor([u["prompt_text"].shape[1]], dtype=torch.int32),
              prompt_speech_token=u["llm_prompt_speech_token"], prompt_speech_token_len=torch.tensor([11], dtype=torch.int32),
              embedding=u["llm_embedding"])
    toks = list(lm.inference(**kw, max_token_text_ratio=4, min_token_text_ratio=4))
    # prefill hidden state of the reference (full-sequence forward) for a stage-level check
    with torch.inference_mode():
        text = torch.cat([u["prompt_text"], u["text"]], 1)
        lm_input = torch.cat([lm.llm_embedding.weight[0].reshape(1, 1, -1), lm.llm.model.model.embed_tokens(text),
                              lm.llm_embedding.weight[1].reshape(1, 1, -1), lm.speech_embedding(u["llm_prompt_speech_token"])], 1)
        hid, _ = lm.llm.forward_one_step(lm_input, masks=torch.tril(torch.ones((1, lm_input.shape[1], lm_input.shape[1]))).to(torch.bool))
    save("llm_tiny", text=u["text"], prompt_text=u["prompt_text"], prompt_speech_token=u["llm_prompt_speech_token"],
         tokens=np.array(toks), logp=torch.stack(logps[:8]), prefill_hidden=hid[0], lm_input=lm_input[0])

    # ras_sampling decision logic with the multinomial draw replaced by an inverse-CDF on queued uniforms
    us = list(np.random.default_rng(7).random(400))
    orig = torch.Tensor.multinomial

    def fake_multinomial(self, n, replacement=False):
        uu = us.pop(0)
        cdf = torch.cumsum(self.double() / self.double().sum(), 0)
        return torch.searchsorted(cdf, torch.tensor([uu], dtype=torch.float64), right=True).clamp(max=self.numel() - 1)

    torch.Tensor.multinomial = fake_multinomial
    try:
        g = torch.Generator().manual_seed(3)
        rows, outs, decs = [], [], []
        decoded = []
        for i in range(60):
            logp = (torch.randn(cfg.speech_token_size + 3, generator=g) * (2.0 if i % 2 else 0.3)).log_softmax(0)
            rows.append(logp.clone()); decs.append(list(decoded[-10:]) + [-1] * (10 - len(decoded[-10:])))
            n_before = len(us)
            t = ras_sampling(logp, decoded, 25)
            outs.append([t, n_before - len(us)])
            decoded.append(t)
    finally:
        torch.Tensor.multinomial = orig
    save("ras_sampling", logp=torch.stack(rows), out=np.array(outs), uniforms=np.random.default_rng(7).random(400), window=np.array(decs))


# ------------------------------------------------------------------------------------------------ flow
def build_ref_flow(cfg):
    from omegaconf import DictConfig
    from cosyvoice.flow.decoder import CausalConditionalDecoder
    from cosyvoice.flow.flow import CausalMaskedDiffWithXvec
    from cosyvoice.flow.flow_matching import CausalConditionalCFM
    from cosyvoice.transformer.upsample_encoder import UpsampleConformerEncoder
    enc = UpsampleConformerEncoder(output_size=cfg.dim, attention_heads=cfg.enc_heads, linear_units=cfg.ffn, num_blocks=cfg.enc_blocks,
                                   dropout_rate=0.1, positional_dropout_rate=0.1, attention_dropout_rate=0.1, normalize_before=True,
                                   input_layer="linear", pos_enc_layer_type="rel_pos_espnet", selfattention_layer_type="rel_selfattn",
                                   input_size=cfg.dim, use_cnn_module=False, macaron_style=False, static_chunk_size=cfg.chunk)
    enc.up_encoders = enc.up_encoders[: cfg.up_blocks]
    est = CausalConditionalDecoder(in_channels=4 * cfg.mel, out_channels=cfg.mel, channels=[cfg.est_ch], dropout=0.0, attention_head_dim=64,
                                   n_blocks=cfg.est_blocks, num_mid_blocks=cfg.est_mid, num_heads=cfg.est_heads, act_fn="gelu",
                                   static_chunk_size=cfg.chunk * 2, num_decoding_left_chunks=-1)
    cfm = CausalConditionalCFM(in_channels=240, n_spks=1, spk_emb_dim=80,
                               cfm_params=DictConfig({"sigma_min": 1e-6, "solver": "euler", "t_scheduler": "cosine",
                                                      "training_cfg_rate": 0.2, "inference_cfg_rate": cfg.cfg_rate, "reg_loss_type": "l1"}),
                               estimator=est)
    flow = CausalMaskedDiffWithXvec(input_size=cfg.dim, output_size=cfg.mel, spk_embed_dim=cfg.spk_dim, vocab_size=cfg.vocab,
                                    input_frame_rate=25, token_mel_ratio=2, pre_lookahead_len=cfg.pre_lookahead, encoder=enc, decoder=cfm)
    flow.load_state_dict(W.make_flow(cfg), strict=True)
    return flow.eval()


def golden_flow():
    cfg = W.ref_small_flow()
    flow = build_ref_flow(cfg)
    g = torch.Generator().manual_seed(11)
    n_p, n_t = 9, 16
    prompt_token = torch.randint(0, cfg.vocab, (1, n_p), generator=g, dtype=torch.int32)
    token = torch.randint(0, cfg.vocab, (1, n_t), generator=g, dtype=torch.int32)
    prompt_feat = torch.randn(1, 2 * n_p, 80, generator=g) * 2 - 5
    emb = torch.randn(1, cfg.spk_dim, generator=g)
    common = dict(prompt_token=prompt_token, prompt_token_len=torch.tensor([n_p], dtype=torch.int32), prompt_feat=prompt_feat,
                  prompt_feat_len=torch.tensor([2 * n_p], dtype=torch.int32), embedding=emb)
    mel_full, _ = flow.inference(token=token, token_len=torch.tensor([n_t], dtype=torch.int32), streaming=False, finalize=True, **common)
    mel_stream, _ = flow.inference(token=token, token_len=torch.tensor([n_t], dtype=torch.int32), streaming=True, finalize=False, **common)
    # estimator boundary B3 (flow_matching.py:126-128) on random inputs, both mask modes
    T = 37
    x = torch.randn(2, 80, T, generator=g); mu = torch.randn(2, 80, T, generator=g); cond = torch.randn(2, 80, T, generator=g)
    spk = torch.randn(2, 80, generator=g); t = torch.tensor([0.3, 0.3]); mask = torch.ones(2, 1, T)
    with torch.inference_mode():
        e_full = flow.decoder.estimator(x, mask, mu, t, spk, cond, streaming=False)
        e_stream = flow.decoder.estimator(x, mask, mu, t, spk, cond, streaming=True)
        # encoder boundary B4
        tok_emb = flow.input_embedding(torch.cat([prompt_token, token], 1).long())
        h_full, _ = flow.encoder(tok_emb, torch.tensor([n_p + n_t]), streaming=False)
        h_ctx, _ = flow.encoder(tok_emb[:, :-3], torch.tensor([n_p + n_t]), context=tok_emb[:, -3:], streaming=True)
    save("flow_small", prompt_token=prompt_token, token=token, prompt_feat=prompt_feat, embedding=emb, mel_full=mel_full,
         mel_stream=mel_stream, est_x=x, est_mu=mu, est_cond=cond, est_spk=spk, est_t=t, est_full=e_full, est_stream=e_stream,
         enc_full=h_full, enc_ctx=h_ctx)


# ------------------------------------------------------------------------------------------------ HiFT
def build_ref_hift(cfg):
    from cosyvoice.hifigan.f0_predictor import ConvRNNF0Predictor
    from cosyvoice.hifigan.generator import HiFTGenerator
    hift = HiFTGenerator(in_channels=cfg.mel, base_channels=cfg.base, nb_harmonics=cfg.harmonics, sampling_rate=cfg.sr,
                         nsf_alpha=cfg.nsf_alpha, nsf_sigma=cfg.nsf_sigma, nsf_voiced_threshold=cfg.voiced_thr,
                         upsample_rates=cfg.ups, upsample_kernel_sizes=cfg.up_k, istft_params={"n_fft": cfg.n_fft, "hop_len": cfg.hop},
                         resblock_kernel_sizes=cfg.res_k, resblock_dilation_sizes=[cfg.res_d] * 3,
                         source_resblock_kernel_sizes=cfg.src_k, source_resblock_dilation_sizes=[cfg.res_d] * 3,
                         lrelu_slope=cfg.lrelu, audio_limit=cfg.audio_limit,
                         f0_predictor=ConvRNNF0Predictor(num_class=1, in_channels=cfg.mel, cond_channels=cfg.f0_ch))
    hift.load_state_dict(W.make_hift(cfg), strict=True)
    return hift.eval()


def golden_hift():
    cfg = W.tiny()[2]
    hift = build_ref_hift(cfg)
    g = torch.Generator().manual_seed(12)
    m = 9
    mel = torch.randn(1, 80, m, generator=g) * 2 - 5
    with torch.inference_mode():
        f0 = hift.f0_predictor(mel)
        torch.manual_seed(99)
        speech, source = hift.inference(speech_feat=mel)
        # the global-RNG draws inference() consumed, in order (generator.py:245, :312; :374 is unused)
        torch.manual_seed(99)
        rand_ini = torch.rand(1, 9); rand_ini[:, 0] = 0
        noise = torch.randn_like(torch.empty(1, 9, 480 * m).transpose(1, 2))     # sine_waves is a transposed view: same strides, same fill order
        cache = torch.randn(1, 1, 480 * 2, generator=g) * 0.1
        torch.manual_seed(99)
        speech_c, source_c = hift.inference(speech_feat=mel, cache_source=cache)
    save("hift_tiny", mel=mel, f0=f0, speech=speech, source=source, rand_ini=rand_ini, noise=noise, cache=cache,
         speech_c=speech_c, source_c=source_c)


def golden_llm_cv3():
    """CosyVoice3LM (llm/llm.py:664-706): sos / task_id rows of speech_embedding, bias-free head over speech_token_size + 200 ids,
    200 stop ids, <|endofprompt|> required.  Same random-init Qwen2 backbone wrapper as golden_llm()."""
    from transformers import Qwen2Config, Qwen2ForCausalLM
    import cosyvoice.llm.llm as L

    cfg = W.tiny_cv3_llm()

    class Enc(L.Qwen2Encoder):
        def __init__(self):
            torch.nn.Module.__init__(self)
            hc = Qwen2Config(vocab_size=cfg.text_vocab, hidden_size=cfg.hidden, intermediate_size=cfg.inter,
                             num_hidden_layers=cfg.layers, num_attention_heads=cfg.heads, num_key_value_heads=cfg.kv_heads,
                             max_position_embeddings=4096, rms_norm_eps=cfg.rms_eps, rope_theta=cfg.rope_theta,
                             tie_word_embeddings=True, attention_dropout=0.0)
            self.model = Qwen2ForCausalLM(hc)

        def forward_one_step(self, xs, masks, cache=None):       # see golden_llm(): intended mask semantics
            outs = self.model(inputs_embeds=xs, attention_mask=None if xs.shape[1] == 1 else masks[:, -1, :],
                              output_hidden_states=True, return_dict=True, use_cache=True, past_key_values=cache)
            return outs.hidden_states[-1], outs.past_key_values

    sd = W.make_llm(cfg)
    logps = []

    def greedy(scores, decoded, k):
        logps.append(scores.clone())
        return int(scores.argmax().item())

    lm = L.CosyVoice3LM(cfg.hidden, cfg.hidden, cfg.speech_token_size, Enc(), greedy)
    lm.load_state_dict(sd, strict=True)
    lm.eval()
    # the reference hard-codes id 151646 (:479): the tiny text vocabulary is extended to hold it, the text carries it once
    u = W.synthetic_utterance(cfg, W.tiny()[1], n_prompt_tok=9, n_prompt_text=4, n_text=5, seed=77)
    u["prompt_text"][0, -1] = 151646
    kw = dict(text=u["text"], text_len=torch.tensor([u["text"].shape[1]], dtype=torch.int32),
              prompt_text=u["prompt_text"], prompt_text_len=torch.tensor([u["prompt_text"].shape[1]], dtype=torch.int32),
              prompt_speech_token=u["llm_prompt_speech_token"], prompt_speech_token_len=torch.tensor([9], dtype=torch.int32),
              embedding=u["llm_embedding"])
    toks = list(lm.inference(**kw, max_token_text_ratio=5, min_token_text_ratio=3))
    save("llm_cv3_tiny", text=u["text"], prompt_text=u["prompt_text"], prompt_speech_token=u["llm_prompt_speech_token"],
         tokens=np.array(toks), logp=torch.stack(logps[:8]))


def golden_llm_bistream():
    """Qwen2LM.inference_bistream (llm/llm.py:551-661) on the real reference class: text arrives in uneven chunks, prompt of 20 speech
    tokens (one 5:15 mix + a remainder), forced fill tokens, final decode to eos.  Greedy sampler restricted so that the random
    fixture model terminates: special ids other than eos / fill are never the argmax (a trained model does not emit them)."""
    from transformers import Qwen2Config, Qwen2ForCausalLM
    import cosyvoice.llm.llm as L

    cfg = W.tiny()[0]

    class Enc(L.Qwen2Encoder):
        def __init__(self):
            torch.nn.Module.__init__(self)
            hc = Qwen2Config(vocab_size=cfg.text_vocab, hidden_size=cfg.hidden, intermediate_size=cfg.inter,
                             num_hidden_layers=cfg.layers, num_attention_heads=cfg.heads, num_key_value_heads=cfg.kv_heads,
                             max_position_embeddings=4096, rms_norm_eps=cfg.rms_eps, rope_theta=cfg.rope_theta,
                             tie_word_embeddings=True, attention_dropout=0.0)
            self.model = Qwen2ForCausalLM(hc)

        def forward_one_step(self, xs, masks, cache=None):       # see golden_llm(): intended mask semantics (plain causal over the cache)
            real = cache.real if cache is not None else None
            outs = self.model(inputs_embeds=xs, attention_mask=None, output_hidden_states=True, return_dict=True, use_cache=True, past_key_values=real)
            return outs.hidden_states[-1], LegacyView(outs.past_key_values)

    class LegacyView:
        """inference_bistream reads `cache[0][0].size(2)` (the tuple cache of transformers 4.51) only to size a mask this wrapper's
        forward ignores; transformers 5.x returns a DynamicCache, so expose just that one shape."""
        def __init__(self, real):
            self.real = real

        def __getitem__(self, i):
            return (torch.empty(1, 1, self.real.get_seq_length(), 1),)

    sts = cfg.speech_token_size

    def greedy(scores, decoded, k):                              # the plain greedy sampler (what the device implements)
        return int(scores.argmax().item())

    g = torch.Generator().manual_seed(5)
    chunks = [torch.randint(0, cfg.text_vocab, (1, n), generator=g, dtype=torch.int32) for n in (3, 4, 6, 2, 7)]
    prompt_text = torch.randint(0, cfg.text_vocab, (1, 4), generator=g, dtype=torch.int32)
    prompt_sp = torch.randint(0, sts, (1, 20), generator=g, dtype=torch.int32)
    base = W.make_llm(cfg)
    for eos_bias in (1.0, 2.0, 3.0, 4.0, 5.0, 6.0, 8.0):       # smallest eos bias with which the random model ends its final phase
        sd = W.bistream_fixture(base, cfg, eos_bias)
        lm = L.Qwen2LM(cfg.hidden, cfg.hidden, sts, Enc(), greedy)
        lm.load_state_dict(sd, strict=True)
        lm.eval()
        toks, ok = [], True
        for t in lm.inference_bistream(text=iter(chunks), prompt_text=prompt_text, prompt_text_len=torch.tensor([4], dtype=torch.int32),
                                       prompt_speech_token=prompt_sp, prompt_speech_token_len=torch.tensor([20], dtype=torch.int32),
                                       embedding=torch.zeros(1, 192)):
            toks.append(int(t))
            if len(toks) > 150:
                ok = False
                break
        print("eos_bias", eos_bias, "->", len(toks), "tokens", "(terminated)" if ok else "(no eos)")
        if ok:
            break
    assert ok
    save("llm_bistream_tiny", prompt_text=prompt_text, prompt_speech_token=prompt_sp, tokens=np.array(toks), eos_bias=np.array(eos_bias),
         **{"chunk%d" % i: c for i, c in enumerate(chunks)})


def golden_glue():
    """fade_in_out + masks (cosyvoice/utils/common.py:170-178, utils/mask.py)."""
    from cosyvoice.utils.common import fade_in_out
    from cosyvoice.utils.mask import subsequent_chunk_mask
    g = torch.Generator().manual_seed(13)
    a = torch.randn(1, 9000, generator=g); b = torch.randn(1, 3840, generator=g)
    win = np.hamming(2 * 3840)
    out = fade_in_out(a, b, win)
    save("glue", fade_a=a, fade_b=b, fade_out=out, chunk_mask=subsequent_chunk_mask(23, 5).to(torch.uint8))


def build_ref_dit_flow(cfg):
    from omegaconf import DictConfig
    from cosyvoice.flow.DiT.dit import DiT
    from cosyvoice.flow.flow import CausalMaskedDiffWithDiT
    from cosyvoice.flow.flow_matching import CausalConditionalCFM
    from cosyvoice.transformer.upsample_encoder import PreLookaheadLayer
    est = DiT(dim=cfg.est_ch, depth=cfg.est_blocks, heads=cfg.est_heads, dim_head=64, ff_mult=cfg.est_mid, mel_dim=cfg.mel, mu_dim=cfg.mel, spk_dim=cfg.mel,
              out_channels=cfg.mel, static_chunk_size=2 * cfg.chunk, num_decoding_left_chunks=-1, dropout=0.0)
    cfm = CausalConditionalCFM(in_channels=240, n_spks=1, spk_emb_dim=80,
                               cfm_params=DictConfig({"sigma_min": 1e-6, "solver": "euler", "t_scheduler": "cosine",
                                                      "training_cfg_rate": 0.2, "inference_cfg_rate": cfg.cfg_rate, "reg_loss_type": "l1"}),
                               estimator=est)
    flow = CausalMaskedDiffWithDiT(input_size=cfg.dim, output_size=cfg.mel, spk_embed_dim=cfg.spk_dim, vocab_size=cfg.vocab, input_frame_rate=25,
                                   token_mel_ratio=2, pre_lookahead_len=cfg.pre_lookahead,
                                   pre_lookahead_layer=PreLookaheadLayer(in_channels=cfg.dim, channels=cfg.ffn, pre_lookahead_len=cfg.pre_lookahead), decoder=cfm)
    flow.load_state_dict(W.make_flow_dit(cfg), strict=True)
    return flow.eval()


def golden_dit():
    """a17: the REAL CausalMaskedDiffWithDiT / DiT / DiTBlock / PreLookaheadLayer (x_transformers' rotary restated in xtransformers_stub.py).  The
    reference hard-codes 10 Euler steps (flow/flow.py:409); the fixture uses cfg.n_timesteps through the CFM's own n_timesteps argument."""
    import cosyvoice.flow.flow as FF
    cfg = W.tiny_cv3_flow()
    flow = build_ref_dit_flow(cfg)
    g = torch.Generator().manual_seed(21)
    n_p, n_t = 6, 11
    prompt_token = torch.randint(0, cfg.vocab, (1, n_p), generator=g, dtype=torch.int32)
    token = torch.randint(0, cfg.vocab, (1, n_t), generator=g, dtype=torch.int32)
    prompt_feat = torch.randn(1, 2 * n_p, 80, generator=g) * 2 - 5
    emb = torch.randn(1, cfg.spk_dim, generator=g)
    orig_fwd = type(flow.decoder).forward

    def fwd(self, mu, mask, spks, cond, n_timesteps=10, **kw):
        return orig_fwd(self, mu=mu, mask=mask, spks=spks, cond=cond, n_timesteps=cfg.n_timesteps, **kw)
    type(flow.decoder).forward = fwd
    try:
        common = dict(prompt_token=prompt_token, prompt_token_len=torch.tensor([n_p], dtype=torch.int32), prompt_feat=prompt_feat,
                      prompt_feat_len=torch.tensor([2 * n_p], dtype=torch.int32), embedding=emb)
        mel_full, _ = flow.inference(token=token, token_len=torch.tensor([n_t], dtype=torch.int32), streaming=False, finalize=True, **common)
        mel_stream, _ = flow.inference(token=token, token_len=torch.tensor([n_t], dtype=torch.int32), streaming=True, finalize=False, **common)
    finally:
        type(flow.decoder).forward = orig_fwd
    T = 27
    x = torch.randn(2, 80, T, generator=g); mu = torch.randn(2, 80, T, generator=g); cond = torch.randn(2, 80, T, generator=g)
    spk = torch.randn(2, 80, generator=g); t = torch.tensor([0.3, 0.3]); mask = torch.ones(2, 1, T)
    with torch.inference_mode():
        e_full = flow.decoder.estimator(x, mask, mu, t, spk, cond, streaming=False)
        e_stream = flow.decoder.estimator(x, mask, mu, t, spk, cond, streaming=True)
    save("dit_tiny", prompt_token=prompt_token, token=token, prompt_feat=prompt_feat, embedding=emb, mel_full=mel_full, mel_stream=mel_stream,
         est_x=x, est_mu=mu, est_cond=cond, est_spk=spk, est_t=t, est_full=e_full, est_stream=e_stream)


def build_ref_causal_hift(cfg):
    from cosyvoice.hifigan.f0_predictor import CausalConvRNNF0Predictor
    from cosyvoice.hifigan.generator import CausalHiFTGenerator
    import cosyvoice.hifigan.generator as GEN
    # the generator allocates 300 s x 24 kHz uniform noise buffers at construction (generator.py:223-226, 355-356): shrink them for the fixture
    orig_rand = torch.rand
    torch.rand = lambda *shape, **kw: orig_rand(*[min(int(x), 48000) if i == 1 and len(shape) == 3 else x for i, x in enumerate(shape)], **kw)
    try:
        torch.manual_seed(1234)
        hift = CausalHiFTGenerator(in_channels=cfg.mel, base_channels=cfg.base, nb_harmonics=cfg.harmonics, sampling_rate=cfg.sr,
                                   nsf_alpha=cfg.nsf_alpha, nsf_sigma=cfg.nsf_sigma, nsf_voiced_threshold=cfg.voiced_thr,
                                   upsample_rates=cfg.ups, upsample_kernel_sizes=cfg.up_k, istft_params={"n_fft": cfg.n_fft, "hop_len": cfg.hop},
                                   resblock_kernel_sizes=cfg.res_k, resblock_dilation_sizes=[cfg.res_d] * 3,
                                   source_resblock_kernel_sizes=cfg.src_k, source_resblock_dilation_sizes=[cfg.res_d] * 3,
                                   lrelu_slope=cfg.lrelu, audio_limit=cfg.audio_limit, conv_pre_look_right=cfg.look_right,
                                   f0_predictor=CausalConvRNNF0Predictor(num_class=1, in_channels=cfg.mel, cond_channels=cfg.f0_ch))
    finally:
        torch.rand = orig_rand
    hift.load_state_dict(W.make_hift(cfg), strict=True)
    return hift.eval()


def golden_causal_hift():
    """a17: the REAL CausalHiFTGenerator + CausalConvRNNF0Predictor (float64 f0, fixed SineGen2 buffers): one-shot and a non-final chunk."""
    import dataclasses
    cfg = dataclasses.replace(W.tiny()[2], causal=True)
    hift = build_ref_causal_hift(cfg)
    g = torch.Generator().manual_seed(15)
    m = 17
    mel = torch.randn(1, 80, m, generator=g) * 2 - 5
    with torch.inference_mode():
        speech, source = hift.inference(speech_feat=mel, finalize=True)
        speech_c, source_c = hift.inference(speech_feat=mel[:, :, :13], finalize=False)
        f0 = hift.f0_predictor(mel.to(torch.float64), finalize=True).float()
    sg = hift.m_source.l_sin_gen
    save("causal_hift_tiny", mel=mel, f0=f0, speech=speech, source=source, speech_c=speech_c, source_c=source_c,
         rand_ini=sg.rand_ini, noise=sg.sine_waves[:, : 480 * m])


def golden_model_cv3():
    """a17 / B1 for CosyVoice3: the REAL cosyvoice.cli.model.CosyVoice3Model (token2wav with the accumulating mel cache + speech offsets, inherited
    streaming loop, silent-token filter) over the real tiny CausalMaskedDiffWithDiT + CausalHiFTGenerator and a scripted LLM."""
    import dataclasses
    import cosyvoice.cli.model as M
    lc, _, hc0 = W.tiny()
    fc = W.tiny_cv3_flow()
    hc = dataclasses.replace(hc0, causal=True)
    flow, hift = build_ref_dit_flow(fc), build_ref_causal_hift(hc)
    hift.m_source.l_sin_gen.sine_waves.zero_()                  # the fixed SineGen2 noise buffer is a construction-time draw: zeros keep the fixture small
    u = W.synthetic_utterance(lc, dataclasses.replace(fc, spk_dim=fc.spk_dim), n_prompt_tok=8, n_prompt_text=4, n_text=2, seed=21)
    g = torch.Generator().manual_seed(32)
    tokens = torch.randint(0, fc.vocab, (41,), generator=g).tolist()
    tokens[5:13] = [1, 2, 1, 2, 28, 1, 2, 1]                    # a run of silent tokens longer than 5: the filter drops the tail of it

    class ScriptedLLM:
        def inference(self, **kw):
            for t in tokens:
                yield t

    orig_fwd = type(flow.decoder).forward

    def fwd(self, mu, mask, spks, cond, n_timesteps=10, **kw):
        return orig_fwd(self, mu=mu, mask=mask, spks=spks, cond=cond, n_timesteps=fc.n_timesteps, **kw)
    type(flow.decoder).forward = fwd
    M.time.sleep = lambda s: None
    try:
        out = {}
        for stream in (False, True):
            m = M.CosyVoice3Model(ScriptedLLM(), flow, hift)
            m.token_hop_len, m.token_max_hop_len = 5, 20
            with torch.inference_mode():
                chunks = [o["tts_speech"] for o in m.tts(text=u["text"], flow_embedding=u["flow_embedding"], llm_embedding=u["llm_embedding"],
                                                        prompt_text=u["prompt_text"], llm_prompt_speech_token=u["llm_prompt_speech_token"],
                                                        flow_prompt_speech_token=u["flow_prompt_speech_token"], prompt_speech_feat=u["prompt_speech_feat"],
                                                        stream=stream)]
            key = "stream" if stream else "offline"
            out[key + "_n"] = np.array([c.shape[1] for c in chunks])
            out[key] = torch.cat(chunks, 1)
    finally:
        type(flow.decoder).forward = orig_fwd
    save("model_cv3_tiny", tokens=np.array(tokens), **out)


def golden_model():
    """B1 / a1 / a16: the REAL cosyvoice.cli.model.CosyVoice2Model (token2wav + the streaming tts loop with its hop doubling, mel / source /
    speech caches and fade_in_out) around the real tiny flow + HiFT modules and a scripted LLM.  Pins oracle/model.py.  SineGen2's additive
    noise (generator.py:312, `torch.randn_like`) is patched to zeros for the run - the same convention oracle.model.Pipeline and the HIP
    tests use (noise is an explicit argument there); `rand_ini` cannot reach the output of this path."""
    import dataclasses
    import cosyvoice.cli.model as M
    lc, _, hc = W.tiny()
    fc = dataclasses.replace(W.ref_small_flow(), chunk=5, n_timesteps=2)   # the reference encoder hard-codes 512 channels (upsample_encoder.py:238-241)
    flow, hift = build_ref_flow(fc), build_ref_hift(hc)
    u = W.synthetic_utterance(lc, fc, n_prompt_tok=8, n_prompt_text=4, n_text=2, seed=21)
    g = torch.Generator().manual_seed(31)
    tokens = torch.randint(0, fc.vocab, (37,), generator=g).tolist()

    class ScriptedLLM:
        def inference(self, **kw):
            for t in tokens:
                yield t

    # the reference hard-codes 10 Euler steps (flow/flow.py:278); the tiny fixtures use 2, like every other model-level test here
    orig_fwd = type(flow.decoder).forward

    def fwd(self, mu, mask, spks, cond, n_timesteps=10, **kw):
        return orig_fwd(self, mu=mu, mask=mask, spks=spks, cond=cond, n_timesteps=2, **kw)
    type(flow.decoder).forward = fwd
    orig_randn_like = torch.randn_like
    torch.randn_like = lambda t, **kw: torch.zeros_like(t)
    M.time.sleep = lambda s: None
    try:
        out = {}
        for stream in (False, True):
            m = M.CosyVoice2Model(ScriptedLLM(), flow, hift)
            m.token_hop_len, m.token_max_hop_len = 5, 20
            with torch.inference_mode():
                chunks = [o["tts_speech"] for o in m.tts(text=u["text"], flow_embedding=u["flow_embedding"], llm_embedding=u["llm_embedding"],
                                                        prompt_text=u["prompt_text"], llm_prompt_speech_token=u["llm_prompt_speech_token"],
                                                        flow_prompt_speech_token=u["flow_prompt_speech_token"], prompt_speech_feat=u["prompt_speech_feat"],
                                                        stream=stream)]
            key = "stream" if stream else "offline"
            out[key + "_n"] = np.array([c.shape[1] for c in chunks])
            out[key] = torch.cat(chunks, 1)
        # speed != 1 (cli/model.py:320-322)
        m = M.CosyVoice2Model(ScriptedLLM(), flow, hift)
        with torch.inference_mode():
            out["speed"] = next(iter(m.tts(text=u["text"], flow_embedding=u["flow_embedding"], llm_embedding=u["llm_embedding"], prompt_text=u["prompt_text"],
                                           llm_prompt_speech_token=u["llm_prompt_speech_token"], flow_prompt_speech_token=u["flow_prompt_speech_token"],
                                           prompt_speech_feat=u["prompt_speech_feat"], stream=False, speed=1.3)))["tts_speech"]
    finally:
        torch.randn_like = orig_randn_like
        type(flow.decoder).forward = orig_fwd
    save("model_tiny", tokens=np.array(tokens), **out)


def golden_estimator_module():
    """B3: cosyvoice_amd.flow.EstimatorModule (an nn.Module over the product estimator) dropped into the REAL ConditionalCFM.forward_estimator /
    solve_euler (flow_matching.py:71-153).  There is no GPU in the build container, so the product estimator is driven through the CPU-emulator
    build of the kernels; what is checked here is the nn.Module branch of forward_estimator, the in-place buffer reuse of solve_euler (:103-108)
    and the dtype / layout contract.  The test re-runs the same comparison against the stored result."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from emu.build_emu import build_emu
    from cosyvoice_amd._lib import Lib
    from cosyvoice_amd.flow import CausalMaskedDiffWithXvec, EstimatorModule
    cfg = W.ref_small_flow()
    flow = build_ref_flow(cfg)
    lib = Lib(build_emu(), allow_emulated=True)
    mine = CausalMaskedDiffWithXvec(W.make_flow(cfg), cfg, lib=lib)
    g = torch.Generator().manual_seed(17)
    T = 23
    mu = torch.randn(1, 80, T, generator=g); spks = torch.randn(1, 80, generator=g); cond = torch.randn(1, 80, T, generator=g)
    mask = torch.ones(1, 1, T)
    with torch.inference_mode():
        want, _ = flow.decoder(mu=mu, mask=mask, spks=spks, cond=cond, n_timesteps=3, streaming=False)
        flow.decoder.estimator = EstimatorModule(mine)               # the swap a maintainer would do
        assert isinstance(flow.decoder.estimator, torch.nn.Module)
        got, _ = flow.decoder(mu=mu, mask=mask, spks=spks, cond=cond, n_timesteps=3, streaming=False)
    err = (got - want).abs().max().item()
    print("EstimatorModule inside the real solve_euler: max |diff| = %.2e" % err)
    assert err < 1e-3, err
    save("estimator_module", mu=mu, spks=spks, cond=cond, out=want)


if __name__ == "__main__":
    which = sys.argv[1:] or ["llm", "flow", "hift", "glue"]
    for w in which:
        globals()["golden_" + w]()
