"""Third-party stand-ins (tests/golden/*_stub.py) against independent implementations that ARE installed.

`x_transformers` (rotary helpers of the CosyVoice3 DiT, flow/DiT/modules.py:20) is absent from this image; tests/golden/xtransformers_stub.py restates its published source.
Hugging Face `transformers` carries the same rotary convention in its GPT-J port (frequencies duplicated pairwise, channels rotated in interleaved pairs - the convention
x_transformers' `rotate_half` / `apply_rotary_pos_emb` implement with their '... (d r) -> ... d r' rearranges).  Agreement rules out an error in the restated arithmetic;
that x_transformers 2.x uses THIS convention remains a statement about its published source."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))


def test_rotary_stub_equals_the_gptj_port():
    gptj = pytest.importorskip("transformers.models.gptj.modeling_gptj")
    import xtransformers_stub as X
    g = torch.Generator().manual_seed(5)
    T, H, D = 37, 3, 64
    q = torch.randn(1, H, T, D, generator=g)
    freqs, scale = X.RotaryEmbedding(D).forward_from_seq_len(T)
    ours = X.apply_rotary_pos_emb(q, freqs, scale)
    sincos = gptj.create_sinusoidal_positions(T, D)                      # [T, D]: sin | cos, one column per frequency
    sin, cos = sincos[None, :, : D // 2], sincos[None, :, D // 2:]
    theirs = gptj.apply_rotary_pos_emb(q.transpose(1, 2), sin, cos).transpose(1, 2)       # GPT-J keeps [batch, time, head, dim]
    assert torch.allclose(ours, theirs, rtol=0, atol=2e-6), (ours - theirs).abs().max().item()
    # partial rotary (the DiT turns 64 channels of a wider head): channels beyond rot_dim pass through
    wide = torch.randn(1, H, T, D + 32, generator=g)
    out = X.apply_rotary_pos_emb(wide, freqs, scale)
    assert torch.equal(out[..., D:], wide[..., D:])
