"""CosyVoice-300M on the kernels at its real dimensions (cosyvoice_amd/cosyvoice1_hip.py): where the time goes, on the MI355X.

    gpurun -- python tools/probe_cv1.py [split3] [graphs] [profile] [nofused] [nograph]

Per stage, with one synchronisation per measurement: LM prefill (text encoder + 132-row forward_chunk) and decode step (ms per token over 200 steps), one flow
pass (ms per Euler step at T = 861), one HiFT pass (861 frames), and the host's share: the same python sequencing with the launches replaced by no-ops
(`CV1_PROBE_DRY`: ctypes + allocation cost alone - what a C entry point + hipGraph per stage would remove).  `split3`: the weight GEMMs on the two-sided bf16 split
(Kernels(split3=True)) instead of the fp32 MFMA chain.  `profile`: one short pass of every stage only (for `rocprofv3 --kernel-trace --stats`).
`nofused`: the LM decode step as the launch-per-operator tape instead of cv_lm1_step (round 4, csrc/lm1.hip); `nograph`: cv_lm1_step launching kernel by kernel.
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cosyvoice_amd import cosyvoice1_hip as CK, synthetic as W   # noqa: E402

split3, profile, graphs = "split3" in sys.argv, "profile" in sys.argv, "graphs" in sys.argv
cfg, hcfg = W.cv1()
t0 = time.time()
sd_llm, sd_flow, sd_hift = W.make_cv1_llm(cfg), W.make_cv1_flow(cfg), W.make_hift(hcfg)
print("weights made in %.1f s; split3=%s" % (time.time() - t0, split3), flush=True)
greedy = lambda scores, decoded, sampling: int(scores.argmax().item())
lm = CK.TransformerLM(sd_llm, text_heads=cfg.text_heads, llm_heads=cfg.llm_heads, sampling=greedy, split3=split3)
flow = CK.MaskedDiffWithXvec(sd_flow, enc_heads=cfg.flow_heads, est_heads=cfg.est_heads, input_frame_rate=cfg.input_frame_rate, split3=split3)
hift = CK.HiFTGenerator(sd_hift, hcfg)
if lm.step is not None:
    import ctypes
    lm.fused_step = "nofused" not in sys.argv
    lm.set_step_option("graph", 0 if "nograph" in sys.argv else 1)      # (the probe's default is the graph; the library's is eager)
print("LM decode step: %s" % ("cv_lm1_step (%d launches, %s)" % (lm.step.stat("launches_per_step"), "kernel by kernel" if "nograph" in sys.argv else "one hipGraph") if lm.fused_step
                              else "launch-per-operator tape (3 + 8 launches per layer)"), flush=True)
flow.k.use_graphs = graphs                                   # `graphs`: the estimator tape of a solve as a hipGraph (LaunchTape.capture), opt-in until measured
tl = lambda n: torch.tensor([n], dtype=torch.int32)
g = torch.Generator().manual_seed(300)
text = torch.randint(0, cfg.text_vocab, (1, 25), generator=g, dtype=torch.int32)
ptext = torch.randint(0, cfg.text_vocab, (1, 17), generator=g, dtype=torch.int32)
pspeech = torch.randint(0, cfg.speech_token_size, (1, 87), generator=g, dtype=torch.int32)
emb = torch.randn(1, cfg.spk_dim, generator=g)
sync = torch.cuda.synchronize


def lm_run(n_gen):
    kw = dict(text=text, text_len=tl(25), prompt_text=ptext, prompt_text_len=tl(17), prompt_speech_token=pspeech, prompt_speech_token_len=tl(87), embedding=emb,
              max_token_text_ratio=n_gen / 25, min_token_text_ratio=n_gen / 25)
    sync(); t0 = time.perf_counter()
    it = lm.inference(**kw)
    first = next(it); sync(); t1 = time.perf_counter()
    toks = [first] + list(it); sync(); t2 = time.perf_counter()
    return 1e3 * (t1 - t0), 1e3 * (t2 - t1) / max(1, len(toks) - 1), toks


n_gen = 60 if profile else 200
lm_run(4)
pre, step, toks = lm_run(n_gen)
print("LM: prefill + first token %.2f ms (132 rows), decode %.3f ms per token over %d steps (context 132 -> %d)" % (pre, step, n_gen - 1, 131 + n_gen), flush=True)

n_tok = 500
token = torch.randint(0, cfg.speech_token_size, (1, n_tok), generator=g, dtype=torch.int32)
e0 = torch.zeros(1, 0, dtype=torch.int32)
fkw = dict(token=token, token_len=tl(n_tok), prompt_token=e0, prompt_token_len=tl(0), prompt_feat=torch.zeros(1, 0, 80), prompt_feat_len=tl(0), embedding=emb,
           flow_cache=torch.zeros(1, 80, 0, 2))
if profile:
    flow.n_timesteps = 1
mel, _ = flow.inference(**fkw)
sync(); t0 = time.perf_counter()
mel, _ = flow.inference(**fkw)
sync(); ft = 1e3 * (time.perf_counter() - t0)
print("flow: %.1f ms for T = %d, %d Euler steps (%.2f ms per step incl. encoder / regulator share); graph replays so far %d" % (ft, mel.shape[2], flow.n_timesteps, ft / flow.n_timesteps, flow.k.graph_replays), flush=True)
hift.inference(speech_feat=mel)
sync(); t0 = time.perf_counter()
hift.inference(speech_feat=mel)
sync(); ht = 1e3 * (time.perf_counter() - t0)
print("hift: %.2f ms for %d frames" % (ht, mel.shape[2]), flush=True)
audio_s = mel.shape[2] * 256 / 22050.0
if not profile:
    total = pre + step * (n_tok - 1) + ft + ht
    print("one 500-token utterance (%.2f s of audio), stages summed: %.0f ms -> %.1f audio-s/s (LM %.0f / flow %.0f / HiFT %.1f)" % (audio_s, total, 1e3 * audio_s / total,
          pre + step * (n_tok - 1), ft, ht), flush=True)
    # the host's share: the same sequencing with every library call a no-op
    import ctypes

    class _Dry:
        def __init__(self, lib):
            self._lib = lib

        def raw(self, name, restype=None):                      # what a LaunchTape records and replays
            return lambda *a: 0

        def __getattr__(self, name):
            if name.startswith("cv_"):
                return lambda *a: None
            return getattr(self._lib, name)
    real = lm.k.lib
    lm.fused_step = False                                       # (the host share of the launch-per-operator sequencing: what cv_lm1_step removed)
    for obj in (lm.k, lm.text_encoder.k, lm.llm.k):
        obj.lib = _Dry(real)
    _, dry_step, _ = lm_run(60)
    for obj in (lm.k, lm.text_encoder.k, lm.llm.k):
        obj.lib = real
    realf = flow.k.lib
    flow.k.lib = _Dry(realf)
    sync(); t0 = time.perf_counter(); flow.inference(**fkw); sync(); dry_flow = 1e3 * (time.perf_counter() - t0)
    flow.k.lib = realf
    print("host sequencing alone (library calls replaced by no-ops): LM %.3f ms per token, flow %.1f ms per pass" % (dry_step, dry_flow), flush=True)
