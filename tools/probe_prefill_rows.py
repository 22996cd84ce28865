"""Dev probe (round 3): LLM prefill of the U10 prompt (131 rows) on the weight-stationary rows path (skinny_rows_kernel, option prefill_rows = 1) against the tiled
GEMMs of round 2 (prefill_rows = 0): ms per prefill (best of 10, one synchronisation each), then the 250 greedy tokens of both against the committed oracle
tokens.   gpurun -- python tools/probe_prefill_rows.py"""
import sys, time, os, json, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cosyvoice_amd import synthetic as W
from cosyvoice_amd.llm import Qwen2LM

cfg = W.cv2()[0]
sd = W.make_llm(cfg)
u = W.synthetic_utterance(cfg, W.cv2()[1])
gold = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "u10_oracle_tokens.json")))["tokens"]
lm = Qwen2LM(sd, cfg, max_len=1024, sampling="greedy", decode_chunk=32)
t = lambda n: torch.tensor([n], dtype=torch.int32)
kw = dict(text=u["text"], text_len=t(30), prompt_text=u["prompt_text"], prompt_text_len=t(12), prompt_speech_token=u["llm_prompt_speech_token"],
          prompt_speech_token_len=t(87), embedding=None, max_token_text_ratio=250 / 30, min_token_text_ratio=250 / 30)
for rows in (1, 0, 1, 0):
    lm.lib.cv_llm_set_option(lm._h, b"prefill_rows", C.c_int32(rows))
    x = lm.build_lm_input(u["text"], u["prompt_text"], u["llm_prompt_speech_token"])
    best = 1e9
    for _ in range(10):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        lm.prefill(x)
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) * 1e3)
    toks = list(lm.inference(**kw))
    print("prefill_rows=%d  prefill(%d rows) %.3f ms   250 tokens equal the oracle's: %s" % (rows, x.shape[0], best, toks == gold), flush=True)
