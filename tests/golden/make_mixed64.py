"""Generate tests/golden/mixed64_oracle_tokens.json: the CPU oracle's greedy speech tokens for the 64 utterances of the mixed64 workload
(bench.py `mixed_requests`, BASELINE.json configs[3]: seeds 4000..4063, generated lengths 125 / 250 / 375 / 500 in equal mix) at the REAL
CosyVoice2-0.5B dimensions with the seeded synthetic weights, plus per step the oracle's own top-1 / top-2 log-prob margin.

    python tests/golden/make_mixed64.py          (about 20 minutes on 8 cores; pure oracle, no /root/reference needed)

bench.py checks the tokens of its batched / continuous-batching paths against this file: every utterance must reproduce the oracle's tokens up to
the first step whose oracle margin is a near-tie (<= 1e-3 in log-prob; past such a step two correct fp32 implementations may legitimately part)."""
import json
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import llm as OL  # noqa: E402
from cosyvoice_amd import synthetic as W  # noqa: E402

N_TEXT, N_PROMPT_TEXT, N_PROMPT_TOK = 30, 12, 87


def main():
    lc, fc, _ = W.cv2()
    sd = W.make_llm(lc)
    out = {"workload": "mixed64: seeds 4000 + i, prompt 87 speech tokens, 12+30 text ids, 125/250/375/500 greedy tokens (i % 4), CosyVoice2-0.5B dims, synthetic weights (seed 1986)",
           "utterances": [], "torch": torch.__version__, "threads": torch.get_num_threads()}
    t0 = time.time()
    for i in range(64):
        n_gen = (125, 250, 375, 500)[i % 4]
        u = W.synthetic_utterance(lc, fc, n_prompt_tok=N_PROMPT_TOK, n_prompt_text=N_PROMPT_TEXT, n_text=N_TEXT, seed=4000 + i)
        trace = {}
        ratio = n_gen / N_TEXT
        with torch.inference_mode():
            toks = OL.inference(sd, lc, u["text"], u["prompt_text"], u["llm_prompt_speech_token"], max_token_text_ratio=ratio, min_token_text_ratio=ratio, trace=trace)
        assert len(toks) == n_gen
        margins = []
        for lp in trace["logp"]:
            lp = lp.clone()
            lp[lc.speech_token_size] = -float("inf")
            top2 = torch.topk(lp, 2).values
            margins.append(round(float(top2[0] - top2[1]), 6))
        out["utterances"].append({"index": i, "n_gen": n_gen, "tokens": toks, "min_margin": min(margins),
                                  "near_ties": [[k, m] for k, m in enumerate(margins) if m <= 2e-3]})
        print("utterance %2d: %d tokens, min margin %.3e  (%.0f s)" % (i, n_gen, min(margins), time.time() - t0), flush=True)
    with open(os.path.join(HERE, "mixed64_oracle_tokens.json"), "w") as f:
        json.dump(out, f)
    print("wrote mixed64_oracle_tokens.json")


if __name__ == "__main__":
    main()
