"""Dev probe: LLM prefill (131 rows of U10, and 8 stacked prompts) per GEMM tile shape (CV_GEMM_FORCE_TILE is read at every launch), and the decode step
with the head GEMV as 411 four-wave (rows = 1) or 165 five-wave (rows = 2) workgroups."""
import ctypes as C
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cosyvoice_amd import synthetic as W
from cosyvoice_amd.llm import Qwen2LM

cfg = W.cv2()[0]
sd = W.make_llm(cfg)
u = W.synthetic_utterance(cfg, W.cv2()[1])
lm = Qwen2LM(sd, cfg, max_len=1200, sampling="greedy", decode_chunk=64)
x = lm.build_lm_input(u["text"], u["prompt_text"], u["llm_prompt_speech_token"])
x8 = torch.cat([x] * 8, 0).contiguous()


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for name, env in (("by size", {}), ("tile 4", {"CV_GEMM_FORCE_TILE": "4"}), ("waste<=1.25", {"CV_GEMM_MAX_WASTE": "1.25"}),
                  ("waste<=1.25 blocks>=720", {"CV_GEMM_MAX_WASTE": "1.25", "CV_GEMM_MIN_BLOCKS_F32": "720"}),
                  ("waste<=1.25 blocks>=1500", {"CV_GEMM_MAX_WASTE": "1.25", "CV_GEMM_MIN_BLOCKS_F32": "1500"}),
                  ("blocks>=720", {"CV_GEMM_MIN_BLOCKS_F32": "720"}), ("blocks>=1500", {"CV_GEMM_MIN_BLOCKS_F32": "1500"})):
    for k in ("CV_GEMM_FORCE_TILE", "CV_GEMM_MAX_WASTE", "CV_GEMM_MIN_BLOCKS_F32"):
        os.environ.pop(k, None)
    os.environ.update(env)
    print("%-26s prefill(131 rows) %.2f ms   prefill(8 x 131 rows stacked) %.2f ms" % (name, timed(lambda: lm.prefill(x)), timed(lambda: lm.prefill(x8))), flush=True)
for k in ("CV_GEMM_FORCE_TILE", "CV_GEMM_MAX_WASTE", "CV_GEMM_MIN_BLOCKS_F32"):
    os.environ.pop(k, None)
for rows in (1,):
    lm.lib.cv_llm_set_option(lm._h, b"head_rows", C.c_int32(rows))
    lm.prefill(x)
    sp = lm.make_sampling(250, 250)
    lm.decode(8, sp); torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 0
    while n < 192:
        toks, fin = lm.decode(64, sp); n += len(toks)
    torch.cuda.synchronize()
    print("head_rows=%d decode %.1f us/token" % (rows, (time.perf_counter() - t0) * 1e6 / n), flush=True)
