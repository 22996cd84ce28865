"""LM-only lock-step batched decode at CosyVoice2's dimensions (no flow / HiFT in the process): microseconds per step for NB slots, long contexts optional.
    python tools/probe_lm_batch.py [nb=32] [n_prompt_tok=87] [reps=2]
Used alone for the step time, and under `rocprofv3 --kernel-trace --stats` for clean per-kernel durations (bench extras overlap the LM with the vocoder)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cosyvoice_amd import synthetic as W            # noqa: E402
from cosyvoice_amd.configs import cv2               # noqa: E402
from cosyvoice_amd.llm import Qwen2LM               # noqa: E402


def main():
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    n_prompt = int(sys.argv[2]) if len(sys.argv) > 2 else 87
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    cfg = cv2()[0]
    lm = Qwen2LM(W.make_llm(cfg), cfg, max_len=1024, sampling="greedy", decode_chunk=64)
    u = W.synthetic_utterance(cfg, cv2()[1], n_prompt_tok=n_prompt, n_prompt_text=12, n_text=30)
    req = dict(text=u["text"], prompt_text=u["prompt_text"], prompt_speech_token=u["llm_prompt_speech_token"])
    ratio = 250 / 30
    lm.inference_batch([req] * nb, max_token_text_ratio=ratio, min_token_text_ratio=ratio)           # warm-up, graph capture
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        toks = lm.inference_batch([req] * nb, max_token_text_ratio=ratio, min_token_text_ratio=ratio)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / reps
    n = len(toks[0])
    print("lm_batch nb=%d context %d..%d: %.1f us per step (%d steps, prefill included), %.0f tokens/s, slots identical: %s"
          % (nb, 43 + n_prompt, 43 + n_prompt + n, 1e6 * el / n, n, nb * n / el, all(t == toks[0] for t in toks)), flush=True)


if __name__ == "__main__":
    main()
