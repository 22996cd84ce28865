"""Do independent lock-step decode chains overlap on the MI355X?  G Qwen2LM handles (each its own weights - a probe), each decoding NB slots of U10 on its own
stream / host thread at the same time; prints aggregate tokens/s.  Every launch of the batched step is latency-bound and fills the chip only partly
(profiles/r5_lm_batch_*.txt), so two chains side by side may cost less than their sum - unlike LM next to the vocoder (0.9 x the sum, profiles/r4_batch_serving_ab.txt).
    python tools/probe_lm_groups.py "1x32 2x16 4x8 2x32" """
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cosyvoice_amd import synthetic as W            # noqa: E402
from cosyvoice_amd.configs import cv2               # noqa: E402
from cosyvoice_amd.llm import Qwen2LM               # noqa: E402


def main():
    plans = " ".join(sys.argv[1:]).split() or "1x32 2x16 4x8 2x32".split()
    cfg = cv2()[0]
    sd = W.make_llm(cfg)
    gmax = max(int(p.split("x")[0]) for p in plans)
    lms = [Qwen2LM(sd, cfg, max_len=1024, sampling="greedy", decode_chunk=64) for _ in range(gmax)]
    streams = [torch.cuda.Stream() for _ in range(gmax)]
    u = W.synthetic_utterance(cfg, cv2()[1], n_prompt_tok=87, n_prompt_text=12, n_text=30)
    req = dict(text=u["text"], prompt_text=u["prompt_text"], prompt_speech_token=u["llm_prompt_speech_token"])
    ratio = 250 / 30
    ref = None
    for plan in plans:
        g, nb = (int(x) for x in plan.split("x"))
        outs = [None] * g

        def work(i, reps):
            with torch.cuda.stream(streams[i]):
                for _ in range(reps):
                    outs[i] = lms[i].inference_batch([req] * nb, max_token_text_ratio=ratio, min_token_text_ratio=ratio)
                torch.cuda.current_stream().synchronize()

        for reps in (1, 2):                                        # warm-up (graph capture per batch size), then the timed run
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ths = [threading.Thread(target=work, args=(i, reps)) for i in range(g)]
            [t.start() for t in ths]
            [t.join() for t in ths]
            torch.cuda.synchronize()
            el = (time.perf_counter() - t0) / reps
        n = len(outs[0][0])
        ref = ref or outs[0][0]
        same = all(t == ref for o in outs for t in o)
        print("groups %d x %d slots: %.1f ms per batch of %d x %d tokens, %.0f tokens/s aggregate, %.1f us per (group) step, tokens identical: %s"
              % (g, nb, 1e3 * el, g * nb, n, g * nb * n / el, 1e6 * el / n, same), flush=True)


if __name__ == "__main__":
    main()
