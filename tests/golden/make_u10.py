"""Generate tests/golden/u10_oracle_tokens.json: the CPU oracle's greedy speech tokens for the benchmark utterance U10 (SURVEY.md section 8d,
BASELINE.json configs[1]) at the REAL CosyVoice2-0.5B dimensions with the seeded synthetic weights - 250 tokens, context 131 -> 381 - plus,
per step, the oracle's own top-1 / top-2 log-prob margin (a free-running comparison is only meaningful up to the first near-tie).

    python tests/golden/make_u10.py            (about a minute on 8 cores; pure oracle, no /root/reference needed)

bench.py checks the tokens its timed path produced against this file (outside the timed region); tests/test_zz_fullsize.py uses it for the
free-running first-divergence check.  The weights are regenerated from the seed (numpy PCG64), so the file is valid on any machine."""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import llm as OL  # noqa: E402
from cosyvoice_amd import synthetic as W  # noqa: E402

N_GEN, N_TEXT, N_PROMPT_TEXT, N_PROMPT_TOK = 250, 30, 12, 87


def main():
    lc, fc, _ = W.cv2()
    sd = W.make_llm(lc)
    u = W.synthetic_utterance(lc, fc, n_prompt_tok=N_PROMPT_TOK, n_prompt_text=N_PROMPT_TEXT, n_text=N_TEXT)
    trace = {}
    ratio = N_GEN / N_TEXT
    with torch.inference_mode():
        toks = OL.inference(sd, lc, u["text"], u["prompt_text"], u["llm_prompt_speech_token"], max_token_text_ratio=ratio, min_token_text_ratio=ratio, trace=trace)
    assert len(toks) == N_GEN
    margins = []
    for lp in trace["logp"]:
        lp = lp.clone()
        lp[lc.speech_token_size] = -float("inf")            # eos is masked below min_len (= every step here)
        top2 = torch.topk(lp, 2).values
        margins.append(round(float(top2[0] - top2[1]), 6))
    out = {"workload": "U10: seed 1986, prompt 87 speech tokens, 12+30 text ids, 250 greedy tokens, CosyVoice2-0.5B dims, synthetic weights (seed 1986)",
           "tokens": toks, "top2_margin": margins, "min_margin": min(margins), "torch": torch.__version__, "threads": torch.get_num_threads()}
    with open(os.path.join(HERE, "u10_oracle_tokens.json"), "w") as f:
        json.dump(out, f)
    print("wrote u10_oracle_tokens.json: %d tokens, min top-2 margin %.3e, 10 smallest %s" % (len(toks), min(margins), sorted(margins)[:10]))


if __name__ == "__main__":
    main()
