"""Dev probe: LLM prefill (131 rows of U10, and 8 stacked prompts) with the GEMMs on the exact three-term bf16 split (CV_GEMM_X3 unset) or on the fp32
MFMA chain (CV_GEMM_X3=0; read at every launch)."""
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


def timed(fn, reps=8):
    fn(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for rnd in range(2):
    for name, v in (("fp32 MFMA chain", "0"), ("three-term bf16 split", "1")):
        os.environ["CV_GEMM_X3"] = v
        print("%-22s prefill(131 rows) %.2f ms   prefill(8 x 131 rows stacked) %.2f ms" % (name, timed(lambda: lm.prefill(x)), timed(lambda: lm.prefill(x8))), flush=True)
