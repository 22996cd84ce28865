"""Dev probe (round 3): batch-1 decode with the next-kernel weight prefetch (cv_llm option "prefetch", llm_kernels.h PrefetchArgs) off / 1 / 2 and with
the XCD match broken on purpose (prefetch_shift): us per token over the U10 decode (prompt 131 rows, 250 tokens), tokens compared with the plain chain,
and the per-category chain times (cv_llm_profile_chain: the category's 24 launches as a dependent chain).
    gpurun -- python tools/probe_prefetch.py"""
import sys, time, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cosyvoice_amd import synthetic as W
from cosyvoice_amd.llm import Qwen2LM
from cosyvoice_amd.ops import stream_ptr

cfg = W.cv2()[0]
sd = W.make_llm(cfg)
u = W.synthetic_utterance(cfg, W.cv2()[1])
lm = Qwen2LM(sd, cfg, max_len=1024, sampling="greedy", decode_chunk=32)
ref = None
for mode, shift in ((0, 0), (1, 0), (2, 0), (1, 3), (2, 3), (0, 0), (1, 0)):
    lm.lib.cv_llm_set_option(lm._h, b"prefetch", C.c_int32(mode)); lm.lib.cv_llm_set_option(lm._h, b"prefetch_shift", C.c_int32(shift))
    best = 1e9; toks_all = None
    for rep in range(4):
        lm_input = lm.build_lm_input(u["text"], u["prompt_text"], u["llm_prompt_speech_token"])
        lm.prefill(lm_input); torch.cuda.synchronize(); t1 = time.time()
        sp = lm.make_sampling(250, 250)
        out = []
        while len(out) < 250:
            toks, fin = lm.decode(min(32, 250 - len(out) + 1), sp); out += toks
            if fin: break
        torch.cuda.synchronize(); t2 = time.time()
        best = min(best, (t2 - t1) * 1e6 / max(len(out), 1)); toks_all = out
    if ref is None: ref = toks_all
    cats = {}
    names = ["qkv", "attention", "o_proj", "gate_up", "down", "head"]
    for k, nm in enumerate(names):
        ms1, n1 = C.c_float(0), C.c_int32(0)
        lm.lib.cv_llm_profile_chain(lm._h, k, 20, C.byref(ms1), C.byref(n1), stream_ptr(lm.lib))
        cats[nm] = round(ms1.value * 1e3 / max(n1.value, 1), 2)
    print("prefetch=%d shift=%d  %.1f us/token  tokens_equal=%s  chains(us/launch)=%s" % (mode, shift, best, toks_all == ref, cats), flush=True)
