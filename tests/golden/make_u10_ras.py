"""Generate tests/golden/u10_ras_oracle_tokens.json: the CPU oracle's REPETITION-AWARE-SAMPLED speech tokens for the benchmark utterance U10 at the real
CosyVoice2-0.5B dimensions (seeded synthetic weights), drawn with FIXED uniform variates (two per step: nucleus draw, fallback draw; numpy PCG64 seed 11, a pair
re-drawn while it would put the step's decision within 1e-3 of flipping - the accepted variates are stored in the file) - a 250-token sequence that does not
fall into the short loop greedy decoding of random weights ends in ("the headline parity workload is degenerate").

    python tests/golden/make_u10_ras.py        (about two minutes on 8 cores; pure oracle, no /root/reference needed)

Per step the file also holds how far the decision was from flipping: the distance of the variate that decided it to the nearest edge of its CDF interval,
and of the nucleus mass to top_p where that decides membership (>= 1e-3 by construction, ~100 x the fp32 summation noise of a 6 761-way softmax).  bench.py
replays the workload on the device with the same variates (outside the timed region, nothing from oracle/ imported there): the device's tokens must be these."""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import llm as OL, sampling as OS  # noqa: E402
from cosyvoice_amd import synthetic as W  # noqa: E402

N_GEN, N_TEXT, N_PROMPT_TEXT, N_PROMPT_TOK = 250, 30, 12, 87
SEED = int(os.environ.get("U10_RAS_SEED", "11"))
OUT = os.environ.get("U10_RAS_OUT", os.path.join(HERE, "u10_ras_oracle_tokens.json"))


def margin(scores, decoded, u0, u1, top_p=0.8, top_k=25, win=10, tau=0.1):
    """distance of this step's decision to the nearest alternative (in probability mass)"""
    p, idx = scores.softmax(0).double().sort(descending=True, stable=True)
    cum, n = 0.0, 0
    while n < len(p) and cum < top_p and n < top_k:
        cum += float(p[n]); n += 1
    m = abs(cum - top_p) if n < top_k else 1.0                 # one element more / fewer in the nucleus
    if n > 1:
        m = min(m, abs(cum - float(p[n - 1]) - top_p))
    cdf = torch.cumsum(p[:n] / p[:n].sum(), 0)
    k = min(int(torch.searchsorted(cdf, torch.tensor(float(u0), dtype=torch.float64), right=True)), n - 1)
    lo = float(cdf[k - 1]) if k else 0.0
    m = min(m, float(u0) - lo, float(cdf[k]) - float(u0)) if k < n - 1 else min(m, float(u0) - lo)
    top = int(idx[k])
    if sum(1 for t in decoded[-win:] if t == top) >= win * tau:          # the fallback draw over everything but `top`
        s2 = scores.clone(); s2[top] = -float("inf")
        c2 = torch.cumsum(s2.softmax(0).double(), 0)
        j = min(int(torch.searchsorted(c2, torch.tensor(float(u1), dtype=torch.float64), right=True)), len(c2) - 1)
        m = min(m, float(u1) - (float(c2[j - 1]) if j else 0.0), float(c2[j]) - float(u1))
    return max(m, 0.0)


def main():
    lc, fc, _ = W.cv2()
    sd = W.make_llm(lc)
    u = W.synthetic_utterance(lc, fc, n_prompt_tok=N_PROMPT_TOK, n_prompt_text=N_PROMPT_TEXT, n_text=N_TEXT)
    rng = np.random.default_rng(SEED)
    us, margins, fallbacks, redraws = [], [], [0], [0]

    def sampler(scores, decoded, sampling):
        while True:                                             # a pair that leaves the decision within 1e-3 of flipping is drawn again
            u0, u1 = (np.float32(x) for x in rng.random(2))
            m = margin(scores.clone(), decoded, u0, u1)
            if m >= 1e-3:
                break
            redraws[0] += 1
        us.extend([float(u0), float(u1)]); margins.append(m)
        before = scores.clone()
        t = OS.ras_sampling(scores, decoded, sampling, u=(float(u0), float(u1)))
        fallbacks[0] += int(not torch.equal(before, scores))
        return t
    ratio = N_GEN / N_TEXT
    with torch.inference_mode():
        toks = OL.inference(sd, lc, u["text"], u["prompt_text"], u["llm_prompt_speech_token"], sampling_fn=sampler, max_token_text_ratio=ratio, min_token_text_ratio=ratio)
    assert len(toks) == N_GEN, len(toks)
    out = {"workload": "U10-RAS: U10 (seed 1986, prompt 87 speech tokens, 12+30 text ids, CosyVoice2-0.5B dims, synthetic weights) decoded with ras_sampling "
                       "(top_p 0.8, top_k 25, win 10, tau_r 0.1), two fp32 uniform variates per step (stored; numpy default_rng(%d), %d pairs re-drawn for margin)" % (SEED, redraws[0]),
           "variates": us + [0.5, 0.5, 0.5, 0.5], "tokens": toks, "margin": [round(m, 7) for m in margins], "min_margin": min(margins),
           "distinct": len(set(toks)), "fallback_draws": fallbacks[0], "torch": torch.__version__, "threads": torch.get_num_threads()}
    with open(OUT, "w") as f:
        json.dump(out, f)
    print("wrote u10_ras_oracle_tokens.json: %d tokens, %d distinct, %d fallback draws, min margin %.3e, 8 smallest %s"
          % (len(toks), len(set(toks)), fallbacks[0], min(margins), sorted(margins)[:8]))


if __name__ == "__main__":
    main()
