"""Boundary B2, the device sampler's own random stream: a chi-square test of the RAS sampler on its counter RNG against the distribution the reference's rule
defines (oracle/sampling.py).  Split from test_llm.py so that the CPU suite's files balance over the pytest-xdist workers (it is the slowest LLM test under the emulator)."""
import numpy as np
import pytest
import torch

from cosyvoice_amd.llm import Qwen2LM
from oracle import sampling as OS
from test_llm import tiny_sd  # noqa: F401  (module-scoped fixture: tiny config + seeded state dict)


def _device_rng_draws(lib, cfg, sd, logits, n_draws, tau_r, per_request=256):
    """Tokens drawn by the device sampler (sample_kernel on its own counter RNG, no injected uniforms) from a FIXED distribution: the head's weight
    is zeroed and its bias carries `logits`, so every step samples softmax(logits) whatever the hidden state; each request gets a fresh RNG key
    (seed + request counter, Qwen2LM.make_sampling)."""
    sd2 = dict(sd)
    sd2["llm_decoder.weight"] = torch.zeros_like(sd["llm_decoder.weight"])
    sd2["llm_decoder.bias"] = logits.clone()
    lm = Qwen2LM(sd2, cfg, lib=lib, max_len=per_request + 16, sampling="ras", decode_chunk=per_request, tau_r=tau_r, seed=20260921)
    x = lm.lib.hook(torch.zeros(3, cfg.hidden, dtype=torch.float32, device=lm.device))
    out = []
    while len(out) < n_draws:
        lm.prefill(x)
        toks, _ = lm.decode(per_request, lm.make_sampling(per_request + 1, per_request + 1))    # eos masked throughout (step < min_len)
        assert len(toks) == per_request
        out += toks
    return torch.tensor(out[:n_draws])


def _chi_square_p(counts, probs):
    """Pearson goodness-of-fit p-value; cells with an expectation < 5 are pooled into one."""
    from scipy import stats
    n = counts.sum().item()
    exp = probs.double() * n
    big = exp >= 5
    obs_c = torch.cat([counts[big].double(), counts[~big].double().sum().reshape(1)])
    exp_c = torch.cat([exp[big], exp[~big].sum().reshape(1)])
    if exp_c[-1] < 1e-9:
        assert obs_c[-1] == 0
        obs_c, exp_c = obs_c[:-1], exp_c[:-1]
    chi2 = ((obs_c - exp_c) ** 2 / exp_c).sum().item()
    return chi2, len(exp_c) - 1, float(stats.chi2.sf(chi2, len(exp_c) - 1))


def test_ras_device_rng_distribution(lib, tiny_sd):
    """Distribution-level check of the device sampler on ITS OWN RNG (the logic test above injects the uniforms): >= 10 000 draws per branch
    against the probabilities oracle/sampling.py's rules imply (utils/common.py:138-167).
      tau_r = 2   -> the repetition test can never fire: iid draws from the renormalised top-p / top-k prefix (nucleus_sampling)
      tau_r = 0   -> it always fires: the nucleus pick a is masked and the token is drawn from the rest of the FULL softmax (random_sampling),
                     P(t) = sum_{a != t} P_nucleus(a) p_t / (1 - p_a)
    Pearson chi-square, p > 1e-3 (fixed RNG keys: the outcome is deterministic); plus a serial-correlation check of consecutive draws."""
    cfg, sd = tiny_sd
    V = cfg.speech_token_size + cfg.n_special
    g = torch.Generator().manual_seed(77)
    logits = torch.randn(V, generator=g) * 1.5
    logits[: 12] += 3.0                                            # a dozen likely ids: the 0.8 nucleus is ~10 wide, the tail is long
    logits[cfg.speech_token_size:] = -30.0                         # stop ids practically impossible (eos itself is masked below min_len)
    p = logits.clone(); p[cfg.speech_token_size] = -float("inf")
    p = p.softmax(0)
    # expected nucleus distribution, by the oracle's own rule
    sv, si = p.sort(descending=True, stable=True)
    keep, cum = [], 0.0
    for i in range(V):
        if cum < 0.8 and len(keep) < 25:
            cum += sv[i].item(); keep.append(i)
        else:
            break
    pn = torch.zeros(V, dtype=torch.float64); pn[si[keep]] = sv[keep].double() / sv[keep].double().sum()
    n = 10240 if not lib.emulated else 1536            # the emulator samples ~10 steps / s: the >= 10 000-draw statement is the hardware run's
    draws = _device_rng_draws(lib, cfg, sd, logits, n, tau_r=2.0)
    assert set(draws.tolist()) <= set(si[keep].tolist()), "a nucleus draw fell outside the top-p / top-k prefix"
    chi2, dof, pv = _chi_square_p(torch.bincount(draws, minlength=V), pn)
    print("nucleus branch: chi2 %.1f / %d dof, p = %.3g" % (chi2, dof, pv))
    assert pv > 1e-3, (chi2, dof, pv)
    # serial independence of the counter RNG: the lag-1 contingency of (is the most likely id) x (next is the most likely id)
    top = si[0].item()
    a = (draws[:-1] == top).double(); b = (draws[1:] == top).double()
    corr = ((a - a.mean()) * (b - b.mean())).mean() / (a.std() * b.std())
    assert abs(corr.item()) < 4.0 / (n ** 0.5), corr.item()
    # fallback branch
    pd = p.double()
    pf = torch.zeros(V, dtype=torch.float64)
    for a_id in si[keep].tolist():
        rest = pd.clone(); rest[a_id] = 0.0
        pf += pn[a_id] * rest / rest.sum()
    draws = _device_rng_draws(lib, cfg, sd, logits, n, tau_r=0.0)
    chi2, dof, pv = _chi_square_p(torch.bincount(draws, minlength=V), pf)
    print("fallback branch: chi2 %.1f / %d dof, p = %.3g" % (chi2, dof, pv))
    assert pv > 1e-3, (chi2, dof, pv)
