"""CosyVoice-300M TransformerLM decode at its real dimensions on the MI355X: us per token of the device-resident loop (cv_lm1_decode) over fp32 and bf16 matrices
(TransformerLM weight_dtype, cv_lm1_use_bf16) with 1, 2 or 4 output rows per 16-lane group of the decode GEMVs (options "gemv_rows" / "gemv_rows16").

    gpurun -- python tools/probe_cv1_lm.py [tokens]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cosyvoice_amd import cosyvoice1_hip as CK, synthetic as W   # noqa: E402

cfg, _ = W.cv1()
sd = W.make_cv1_llm(cfg)
n_gen = int(sys.argv[1]) if len(sys.argv) > 1 else 400
g = torch.Generator().manual_seed(300)
text = torch.randint(0, cfg.text_vocab, (1, 25), generator=g, dtype=torch.int32)
emb = torch.randn(1, cfg.spk_dim, generator=g)
e0 = torch.zeros(1, 0, dtype=torch.int32)
tl = lambda n: torch.tensor([n], dtype=torch.int32)
kw = dict(text=text, text_len=tl(25), prompt_text=e0, prompt_text_len=tl(0), prompt_speech_token=e0, prompt_speech_token_len=tl(0), embedding=emb,
          max_token_text_ratio=n_gen / 25, min_token_text_ratio=n_gen / 25)
ref = {}
for dtype, opt in ((None, "gemv_rows"), (torch.bfloat16, "gemv_rows16")):
    lm = CK.TransformerLM(sd, text_heads=cfg.text_heads, llm_heads=cfg.llm_heads, sampling="greedy", weight_dtype=dtype)
    for rows in ((1, 0) if dtype is None else (0, 1)):      # bf16: 0 = the wide layout (lm1_gemv16_kernel, the default), 1 / 2 = lm1_gemv_kernel with that many rows per group
        lm.set_step_option("gemv16_wide" if dtype is not None else "gemv_wide", int(rows == 0))
        lm.set_step_option(opt, max(rows, 1))
        best = 1e9
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            toks = list(lm.inference(**kw))
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        ref.setdefault(dtype, toks)
        print("%-5s layout %d (0 = wide: whole rows per lane group; 1 = lm1_gemv_kernel): %.1f us per token (%d tokens incl. the prompt pass), tokens equal rows=1: %s" %
              ("bf16" if dtype else "fp32", rows, 1e6 * best / len(toks), len(toks), toks == ref[dtype]), flush=True)
