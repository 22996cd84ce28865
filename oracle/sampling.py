"""Oracle: repetition-aware sampling (TEST INFRASTRUCTURE).

Restates cosyvoice/utils/common.py:138-167 (ras_sampling / nucleus_sampling / random_sampling).  The reference draws
with torch.multinomial on the global RNG, which no other implementation can reproduce bit-for-bit; the restatement takes
the uniform variates as an argument and inverts the CDF, so the *decision logic* (stable descending sort, top-p 0.8 /
top-k 25 prefix with the reference's 'add while cum < top_p' rule, repetition window, full-distribution fallback with the
repeated id masked) is comparable exactly.  With `u=None` it falls back to torch.multinomial like the reference.
"""
import torch


def _draw(prob, u):
    if u is None:
        return int(prob.multinomial(1, replacement=True).item())
    cdf = torch.cumsum(prob.double() / prob.double().sum(), 0)
    idx = int(torch.searchsorted(cdf, torch.tensor(float(u), dtype=torch.float64), right=True).item())
    return min(idx, prob.numel() - 1)


def nucleus_sampling(weighted_scores, top_p=0.8, top_k=25, u=None):
    prob, indices = [], []
    cum_prob = 0.0
    sorted_value, sorted_idx = weighted_scores.softmax(dim=0).sort(descending=True, stable=True)
    for i in range(len(sorted_idx)):
        if cum_prob < top_p and len(prob) < top_k:          # common.py:153-158
            cum_prob += sorted_value[i]
            prob.append(sorted_value[i])
            indices.append(sorted_idx[i])
        else:
            break
    prob = torch.tensor(prob).to(weighted_scores)
    indices = torch.tensor(indices, dtype=torch.long)
    return int(indices[_draw(prob, u)].item())


def random_sampling(weighted_scores, decoded_tokens, sampling, u=None):
    return _draw(weighted_scores.softmax(dim=0), u)


def ras_sampling(weighted_scores, decoded_tokens, sampling, top_p=0.8, top_k=25, win_size=10, tau_r=0.1, u=(None, None)):
    top_ids = nucleus_sampling(weighted_scores, top_p=top_p, top_k=top_k, u=u[0])
    rep_num = (torch.tensor(decoded_tokens[-win_size:], dtype=torch.long) == top_ids).sum().item()
    if rep_num >= win_size * tau_r:
        weighted_scores[top_ids] = -float("inf")             # in place, like the reference (common.py:142)
        top_ids = random_sampling(weighted_scores, decoded_tokens, sampling, u=u[1])
    return top_ids
