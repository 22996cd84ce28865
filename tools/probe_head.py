"""Dev probe (round 3): the LM head GEMV as 411 four-wave or 235 seven-wave workgroups (cv_llm option head_waves), us per token over the U10 decode and
the head's own chain time.   gpurun -- python tools/probe_head.py"""
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
for waves, rows in ((4, 1), (7, 1), (4, 2), (4, 1), (7, 1)):
    lm.lib.cv_llm_set_option(lm._h, b"head_waves", C.c_int32(waves)); lm.lib.cv_llm_set_option(lm._h, b"head_rows", C.c_int32(rows))
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
    ms1, n1 = C.c_float(0), C.c_int32(0)
    lm.lib.cv_llm_profile_chain(lm._h, 5, 20, C.byref(ms1), C.byref(n1), stream_ptr(lm.lib))
    print("head_waves=%d head_rows=%d  %.1f us/token  tokens_equal=%s  head chain %.2f us/launch" % (waves, rows, best, out == ref, ms1.value * 1e3 / max(n1.value, 1)), flush=True)
