"""Dev probe (round 3): batch-1 decode with attention + o_proj in one launch (cv_llm option fused_attn_oproj; attn_oproj_kernel) against the five-launch layer:
us per token over the U10 decode (prompt 131 rows, 250 greedy tokens), tokens compared, per-category chain times.   gpurun -- python tools/probe_fused_o.py"""
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
for fused, rb, nw in ((0, 4, 8), (1, 4, 8), (1, 8, 8), (1, 4, 16), (1, 8, 16), (0, 4, 8), (1, 4, 8)):
    for k, v in ((b"fused_attn_oproj", fused), (b"oproj_rblocks", rb), (b"oproj_waves", nw)):
        lm.lib.cv_llm_set_option(lm._h, k, C.c_int32(v))
    best = 1e9
    for rep in range(4):
        lm.prefill(lm.build_lm_input(u["text"], u["prompt_text"], u["llm_prompt_speech_token"])); torch.cuda.synchronize(); t1 = time.time()
        sp = lm.make_sampling(250, 250)
        out = []
        while len(out) < 250:
            toks, fin = lm.decode(min(32, 250 - len(out) + 1), sp); out += toks
            if fin: break
        torch.cuda.synchronize(); best = min(best, (time.time() - t1) * 1e6 / max(len(out), 1))
    ref = ref or out
    cats = {}
    for k, nm in enumerate(["qkv", "attention", "o_proj", "gate_up", "down", "head"]):
        ms1, n1 = C.c_float(0), C.c_int32(0)
        lm.lib.cv_llm_profile_chain(lm._h, k, 20, C.byref(ms1), C.byref(n1), stream_ptr(lm.lib))
        cats[nm] = round(ms1.value * 1e3 / max(n1.value, 1), 2)
    print("fused_attn_oproj=%d rblocks=%d waves=%d  %.1f us/token  tokens_equal=%s  chains(us/launch)=%s" % (fused, rb, nw, best, out == ref, cats), flush=True)
