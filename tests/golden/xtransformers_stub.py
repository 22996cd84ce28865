"""Stand-in for the two symbols of `x_transformers` (pinned x-transformers==2.11.24, requirements.txt:39; not installed here) that the DiT imports
(flow/DiT/dit.py:17, flow/DiT/modules.py:20): `RotaryEmbedding` (only `forward_from_seq_len`) and `apply_rotary_pos_emb`.  Restated from the
published x_transformers 2.x source: PARITY UNPINNED at this boundary - nothing under /root/reference pins it."""
import torch
from torch import nn


class RotaryEmbedding(nn.Module):
    def __init__(self, dim, base=10000):
        super().__init__()
        inv_freq = 1.0 / (base ** (torch.arange(0, dim, 2).float() / dim))
        self.register_buffer("inv_freq", inv_freq, persistent=False)

    def forward_from_seq_len(self, seq_len):
        t = torch.arange(seq_len, device=self.inv_freq.device)
        return self.forward(t)

    def forward(self, t):
        freqs = torch.einsum("i , j -> i j", t.type_as(self.inv_freq), self.inv_freq)
        freqs = torch.stack((freqs, freqs), dim=-1).reshape(*freqs.shape[:-1], -1)      # '... d r -> ... (d r)': every frequency twice, interleaved
        return freqs, 1.0


def rotate_half(x):
    x = x.reshape(*x.shape[:-1], -1, 2)                                                # '... (d r) -> ... d r', r = 2
    x1, x2 = x.unbind(dim=-1)
    return torch.stack((-x2, x1), dim=-1).reshape(*x.shape[:-2], -1)


def apply_rotary_pos_emb(t, freqs, scale=1):
    rot_dim, seq_len, orig_dtype = freqs.shape[-1], t.shape[-2], t.dtype
    freqs = freqs[-seq_len:, :]
    t_left, t_right = t[..., :rot_dim], t[..., rot_dim:]                                # partial rotary: only the first rot_dim channels turn
    t_left = (t_left * freqs.cos() * scale) + (rotate_half(t_left) * freqs.sin() * scale)
    return torch.cat((t_left, t_right), dim=-1).type(orig_dtype)
