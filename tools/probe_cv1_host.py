"""Host sequencing cost of the kernel-backed CosyVoice-300M LM decode step (cosyvoice_amd/cosyvoice1_hip.py), measured WITHOUT a GPU: the real-dimension model's
python sequencing with every library launch replaced by a no-op (the emulator library only supplies the handle).  `eager` disables the LaunchTape replays.

    timeout 600 python tools/probe_cv1_host.py [eager] [prof]        ->  profiles/r3_cv1_host_sequencing.txt
"""
import sys, time, os, torch, cProfile, pstats
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from emu.build_emu import build_emu
from cosyvoice_amd._lib import Lib
from cosyvoice_amd import cosyvoice1_hip as CK, synthetic as W
torch.set_num_threads(2)
lib = Lib(build_emu(), allow_emulated=True)
class Dry:
    def __init__(s, l): s._l = l
    def raw(s, name, restype=None): return lambda *a: 0
    def __getattr__(s, n):
        if n.startswith("cv_"): return lambda *a: None
        return getattr(s._l, n)
cfg, _ = W.cv1()
import dataclasses
cfg = dataclasses.replace(cfg, text_vocab=2000)   # smaller text table: faster weight generation, same layers
sd = W.make_cv1_llm(cfg)
print("weights", flush=True)
dry = Dry(lib)
greedy = lambda s, d, k: 5
lm = CK.TransformerLM(sd, sampling=greedy, lib=dry)
lm.k.use_tapes = "eager" not in sys.argv
t = lambda n: torch.tensor([n], dtype=torch.int32)
g = torch.Generator().manual_seed(1)
text = torch.randint(0, 2000, (1, 25), generator=g, dtype=torch.int32)
e0 = torch.zeros(1, 0, dtype=torch.int32)
kw = dict(text=text, text_len=t(25), prompt_text=e0, prompt_text_len=t(0), prompt_speech_token=e0, prompt_speech_token_len=t(0), embedding=torch.randn(1,192))
def run(n):
    it = lm.inference(max_token_text_ratio=n/25, min_token_text_ratio=n/25, **kw)
    next(it); t0 = time.perf_counter(); k = sum(1 for _ in it); return 1e3*(time.perf_counter()-t0)/k
run(20)
print("host ms per token (no-op launches, %s): %.3f" % ("eager sequencing" if "eager" in sys.argv else "LaunchTape replays", run(200)))
if "prof" in sys.argv:
    pr = cProfile.Profile(); pr.enable(); run(100); pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(18)
