"""Generate tests/golden/cv3_u10_oracle_tokens.json: the CPU oracle's greedy speech tokens for the Fun-CosyVoice3-0.5B instruct request bench.py's
`cosyvoice3` extra synthesises (BASELINE.json configs[4] shape: seed 2025, prompt text of 24 ids with <|endofprompt|> at position 11, no LLM speech
prompt, 30 text ids, 250 tokens) at the real CosyVoice3LM dimensions with the seeded synthetic weights, plus the per-step top-2 margins.

    python tests/golden/make_cv3_u10.py          (about 15 s on 8 cores; pure oracle)"""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import llm as OL  # noqa: E402
from cosyvoice_amd import configs as CF, synthetic as W  # noqa: E402

N_GEN, N_TEXT, N_PROMPT_TOK = 250, 30, 87


def main():
    lc, fc = CF.cv3_llm(), CF.cv3_flow()
    sd = W.make_llm(lc)
    u = W.synthetic_utterance(lc, fc, n_prompt_tok=N_PROMPT_TOK, n_prompt_text=24, n_text=N_TEXT, seed=2025)
    u["prompt_text"][0, 11] = lc.endofprompt_id
    e0 = torch.zeros(1, 0, dtype=torch.int32)
    trace = {}
    ratio = N_GEN / N_TEXT
    with torch.inference_mode():
        toks = OL.inference(sd, lc, u["text"], u["prompt_text"], e0, max_token_text_ratio=ratio, min_token_text_ratio=ratio, trace=trace)
    assert len(toks) == N_GEN
    margins = []
    for lp in trace["logp"]:
        lp = lp.clone()
        lp[lc.speech_token_size] = -float("inf")
        top2 = torch.topk(lp, 2).values
        margins.append(round(float(top2[0] - top2[1]), 6))
    out = {"workload": "Fun-CosyVoice3-0.5B instruct request of bench.py cv3_workload (seed 2025), 250 greedy tokens, synthetic weights",
           "tokens": toks, "top2_margin": margins, "min_margin": min(margins), "torch": torch.__version__}
    with open(os.path.join(HERE, "cv3_u10_oracle_tokens.json"), "w") as f:
        json.dump(out, f)
    print("wrote cv3_u10_oracle_tokens.json: %d tokens, min top-2 margin %.3e, 10 smallest %s" % (len(toks), min(margins), sorted(margins)[:10]))


if __name__ == "__main__":
    main()
