"""The large-M kernel set of the flow estimator (round 4, csrc/flow_big.h): bit identity with the small-tile path, tile by tile and pass by pass.  In a file of its own:
under the emulator every variant is a minute of work, and pytest-xdist hands out whole files."""
import pytest
import torch

from cosyvoice_amd.flow import CausalMaskedDiffWithXvec
from cosyvoice_amd import synthetic as W


@pytest.mark.parametrize("tile0,tile1", [(1, 1), (2, 2), (3, 3), (0, 0), (3, 4), (4, 4)])
def test_big_m_kernels_are_bit_identical(lib, tile0, tile1):
    """bf16 mode, round 4: the large-M kernel set (flow_big.h: LayerNorm once per row -> bf16, 128 x 128 / 128 x 64 / 64 x 64 GEMM tiles over a swizzled LDS ring,
    attention with 32 queries per wave) is selected by the ROW COUNT of a pass, so it has to compute every element exactly as the small-tile path does - an
    utterance's mel must not depend on what it shared a pass with.  Forced on (big_rows = attn2_rows = 1) against forced off (0), bit for bit, on the emulator
    AND on the hardware: two heads, T not a multiple of 4 / 16 / 64 / 128 (V^T pairs straddling requests, ragged row and query tiles), several key tiles,
    both mask modes, CFG batch rows, K = 128 (two k stages) and K = 512 (FF2: eight)."""
    import ctypes as C
    import dataclasses
    cfg = dataclasses.replace(W.tiny()[1], est_ch=128, est_heads=2, est_mid=1, chunk=13)
    sd = W.make_flow(cfg)
    flow = CausalMaskedDiffWithXvec(sd, cfg, lib=lib, precision="bf16")
    g = torch.Generator().manual_seed(14)

    def opt(big, attn2, cap=0, glds=0, epi=1):
        for k, v in (("big_rows", big), ("attn2_rows", attn2), ("big_tile0", tile0), ("big_tile1", tile1), ("big_grid_cap", cap), ("big_glds", glds), ("big_lds_epi", epi)):
            lib.cv_flow_set_option(flow._h, k.encode(), C.c_int32(v))
    try:
        for T in ((45, 150) if lib.emulated else (45, 150, 281)):       # (the third length only on the hardware: the emulator pays T^2 per attention)
            x = torch.randn(2, 80, T, generator=g); mu = torch.randn(2, 80, T, generator=g); cond = torch.randn(2, 80, T, generator=g)
            spk = torch.randn(2, 80, generator=g); t = torch.tensor([0.3, 0.3]); mask = torch.ones(2, 1, T)
            for streaming in (False, True):
                outs = []
                # cap = 3 / 1: the persistent form proper - three workgroups (one) walk all the tiles of a launch, the stage pipeline running across tile boundaries
                # glds = 1: the stages by LDS-DMA (global_load_lds) instead of registers + ds_write
                # epi = 0: per-lane stores from the accumulator layout instead of the row-wise stores through LDS (the default of the one-tile-per-workgroup form)
                for big, attn2, cap, glds, epi in ((0, 0, 0, 0, 1), (1, 0, 0, 0, 1), (0, 1, 0, 0, 1), (1, 1, 0, 0, 1), (1, 0, 3, 0, 1), (1, 1, 1, 0, 1), (1, 0, 0, 1, 1), (1, 0, 2, 1, 1),
                                                   (1, 0, 0, 0, 0), (1, 0, 0, 1, 0)):
                    opt(big, attn2, cap, glds, epi)
                    outs.append(flow.decoder.estimator(x, mask, mu, t, spk, cond, streaming=streaming).cpu().clone())
                assert torch.isfinite(outs[0]).all() and outs[0].abs().max() > 0
                for k in range(1, len(outs)):
                    assert torch.equal(outs[0], outs[k]), (T, streaming, k, (outs[0] - outs[k]).abs().max().item())
    finally:
        opt(5000, 0)
        lib.cv_flow_set_option(flow._h, b"big_tile0", C.c_int32(0)); lib.cv_flow_set_option(flow._h, b"big_tile1", C.c_int32(0)); lib.cv_flow_set_option(flow._h, b"big_grid_cap", C.c_int32(0))


def test_big_m_pass_equals_single_passes(lib):
    """The contract the thresholds rely on, end to end: a padded pass over utterances of different lengths with the large-M kernels forced on gives every
    utterance the mel of its own single pass with them off (cv_flow_inference_ragged, bf16 mode, streaming and not)."""
    import ctypes as C
    cfg = W.tiny()[1]
    cfg = __import__("dataclasses").replace(cfg, n_timesteps=2)
    sd = W.make_flow(cfg)
    flow = CausalMaskedDiffWithXvec(sd, cfg, lib=lib, precision="bf16")
    g = torch.Generator().manual_seed(15)
    items = []
    for n_t, n_p in ((40, 9), (33, 5), (37, 7)):
        items.append(dict(token=torch.randint(0, cfg.vocab, (1, n_t), generator=g, dtype=torch.int32), prompt_token=torch.randint(0, cfg.vocab, (1, n_p), generator=g, dtype=torch.int32),
                          prompt_feat=torch.randn(1, 2 * n_p, 80, generator=g) * 2 - 5, embedding=torch.randn(1, cfg.spk_dim, generator=g)))
    try:
        for k, v in (("big_rows", 0), ("attn2_rows", 0)):
            lib.cv_flow_set_option(flow._h, k.encode(), C.c_int32(v))
        single = [flow.inference_batch([it])[0].cpu().clone() for it in items]
        for k, v in (("big_rows", 1), ("attn2_rows", 1)):
            lib.cv_flow_set_option(flow._h, k.encode(), C.c_int32(v))
        together = [m.cpu().clone() for m in flow.inference_batch(items)]
        for a, b in zip(single, together):
            assert a.shape == b.shape and torch.equal(a, b), (a.shape, (a - b).abs().max().item())
    finally:
        lib.cv_flow_set_option(flow._h, b"big_rows", C.c_int32(5000)); lib.cv_flow_set_option(flow._h, b"attn2_rows", C.c_int32(0))


@pytest.mark.parametrize("est_blocks", [1, 3])
def test_band_kernel_is_bit_identical(lib, est_blocks):
    """Round 5: in a large pass everything between a block's attention and the next block's QKV GEMM is ONE launch per 64-row band (flow_band.h: out-projection +
    residual -> LayerNorm -> FF1 + GELU -> FF2 + residual -> the next block's LayerNorm as bf16 rows; FF1 -> FF2 in chunks with the accumulators kept across them).
    Same rounding points, same k order into one accumulator chain per element, same LayerNorm expressions: bit-identical to the five launches it replaces
    (fused_band = 0) and to the small-tile path (big_rows = 0) - on the emulator AND on the hardware.  est_blocks = 3: the chained form (the band's LayerNorm output
    feeds the next QKV GEMM) and the closing one; T not a multiple of 64 (ragged last band), bands that straddle the CFG batch rows, both mask modes."""
    import ctypes as C
    import dataclasses
    cfg = dataclasses.replace(W.tiny()[1], est_blocks=est_blocks, est_mid=1, chunk=13)
    sd = W.make_flow(cfg)
    flow = CausalMaskedDiffWithXvec(sd, cfg, lib=lib, precision="bf16")
    assert any(k.endswith(".band") for k in flow._tensors), "weights.pack_flow did not produce the band streams"
    g = torch.Generator().manual_seed(16)
    try:
        for T in (45, 150):
            x = torch.randn(2, 80, T, generator=g); mu = torch.randn(2, 80, T, generator=g); cond = torch.randn(2, 80, T, generator=g)
            spk = torch.randn(2, 80, generator=g); t = torch.tensor([0.3, 0.3]); mask = torch.ones(2, 1, T)
            for streaming in (False, True):
                outs = []
                # band_bm = 64 / 48 / 32 rows per band (four, three - a ragged LayerNorm pass - or two row tiles per weight fragment), 0 = by the row count of the pass;
                # band_qkv = 1: the band launch also runs the NEXT block's QKV GEMM (Q | K rows and V^T columns straight from the accumulators) - only with a next block
                # band_pipe = 1 / 2: the FF1 -> GELU -> FF2 chunks of a 48-row (and 32-row) band as a software pipeline (MFMA slices between the GELU pieces, two GELU tiles)
                variants = [(0, 0, 0, 0, 0), (1, 0, 0, 0, 0), (1, 1, 64, 0, 0), (1, 1, 32, 0, 0), (1, 1, 48, 0, 0), (1, 1, 48, 0, 1), (1, 1, 32, 0, 2)]
                if est_blocks > 1:
                    variants += [(1, 1, 64, 1, 0), (1, 1, 32, 1, 0), (1, 1, 48, 1, 0), (1, 1, 48, 1, 1), (1, 1, 32, 1, 2), (1, 1, 0, 1, 1)]
                for big, band, bm, qkv, pipe in variants:
                    lib.cv_flow_set_option(flow._h, b"big_rows", C.c_int32(big)); lib.cv_flow_set_option(flow._h, b"fused_band", C.c_int32(band))
                    lib.cv_flow_set_option(flow._h, b"band_bm", C.c_int32(bm)); lib.cv_flow_set_option(flow._h, b"band_qkv", C.c_int32(qkv)); lib.cv_flow_set_option(flow._h, b"band_pipe", C.c_int32(pipe))
                    outs.append(flow.decoder.estimator(x, mask, mu, t, spk, cond, streaming=streaming).cpu().clone())
                assert torch.isfinite(outs[0]).all() and outs[0].abs().max() > 0
                for k in range(1, len(outs)):
                    assert torch.equal(outs[0], outs[k]), (T, streaming, k, (outs[0] - outs[k]).abs().max().item())
    finally:
        lib.cv_flow_set_option(flow._h, b"big_rows", C.c_int32(5000)); lib.cv_flow_set_option(flow._h, b"fused_band", C.c_int32(1))
        lib.cv_flow_set_option(flow._h, b"band_bm", C.c_int32(0)); lib.cv_flow_set_option(flow._h, b"band_qkv", C.c_int32(1)); lib.cv_flow_set_option(flow._h, b"band_pipe", C.c_int32(2))


def test_band_kernel_at_the_real_width(lib):
    """The instantiation the MI355X runs (C = 256, 8 heads, FF = 1024: 8 waves, two column tiles per wave, the out-projection in two passes) against the five-launch
    form and the small-tile path, bit for bit, on a short time axis (one full band and a ragged one)."""
    import ctypes as C
    import dataclasses
    cfg = dataclasses.replace(W.tiny()[1], est_ch=256, est_heads=8, est_blocks=2, est_mid=1, chunk=13)
    sd = W.make_flow(cfg)
    flow = CausalMaskedDiffWithXvec(sd, cfg, lib=lib, precision="bf16")
    assert any(k.endswith(".band") for k in flow._tensors)
    g = torch.Generator().manual_seed(17)
    T = 37
    x = torch.randn(2, 80, T, generator=g); mu = torch.randn(2, 80, T, generator=g); cond = torch.randn(2, 80, T, generator=g)
    spk = torch.randn(2, 80, generator=g); t = torch.tensor([0.6, 0.6]); mask = torch.ones(2, 1, T)
    try:
        outs = []
        for big, band, bm, qkv, pipe in ((0, 0, 0, 0, 0), (1, 0, 0, 0, 0), (1, 1, 64, 0, 0), (1, 1, 32, 0, 0), (1, 1, 64, 1, 0), (1, 1, 32, 1, 0), (1, 1, 48, 1, 0), (1, 1, 48, 1, 1), (1, 1, 48, 0, 1), (1, 1, 32, 1, 2)):
            lib.cv_flow_set_option(flow._h, b"big_rows", C.c_int32(big)); lib.cv_flow_set_option(flow._h, b"fused_band", C.c_int32(band))
            lib.cv_flow_set_option(flow._h, b"band_bm", C.c_int32(bm)); lib.cv_flow_set_option(flow._h, b"band_qkv", C.c_int32(qkv)); lib.cv_flow_set_option(flow._h, b"band_pipe", C.c_int32(pipe))
            outs.append(flow.decoder.estimator(x, mask, mu, t, spk, cond, streaming=False).cpu().clone())
        assert torch.isfinite(outs[0]).all() and outs[0].abs().max() > 0
        for k in range(1, len(outs)):
            assert torch.equal(outs[0], outs[k]), (k, (outs[0] - outs[k]).abs().max().item())
    finally:
        lib.cv_flow_set_option(flow._h, b"big_rows", C.c_int32(5000)); lib.cv_flow_set_option(flow._h, b"fused_band", C.c_int32(1))
        lib.cv_flow_set_option(flow._h, b"band_bm", C.c_int32(0)); lib.cv_flow_set_option(flow._h, b"band_qkv", C.c_int32(1)); lib.cv_flow_set_option(flow._h, b"band_pipe", C.c_int32(2))


@pytest.mark.experiments
def test_eager_large_pass_runs_two_chains(lib):
    """Round 5: a pass that runs eager (graph_max_rows) takes the estimator's batch rows as TWO launch chains on two streams (option eager_streams = 2, the default) - the
    same kernels on the same rows, the band height chosen from the rows of the WHOLE pass: every utterance's mel is the one chain's, bit for bit, in an equal-shape pass
    and in a padded one, with the large-M kernel set and the band + QKV launch on."""
    import ctypes as C
    import dataclasses
    cfg = dataclasses.replace(W.tiny()[1], n_timesteps=2, est_blocks=2, est_mid=1)
    sd = W.make_flow(cfg)
    flow = CausalMaskedDiffWithXvec(sd, cfg, lib=lib, precision="bf16")
    g = torch.Generator().manual_seed(18)
    items = []
    for n_t, n_p in ((40, 9), (33, 5), (37, 7), (40, 9)):
        items.append(dict(token=torch.randint(0, cfg.vocab, (1, n_t), generator=g, dtype=torch.int32), prompt_token=torch.randint(0, cfg.vocab, (1, n_p), generator=g, dtype=torch.int32),
                          prompt_feat=torch.randn(1, 2 * n_p, 80, generator=g) * 2 - 5, embedding=torch.randn(1, cfg.spk_dim, generator=g)))
    opt = lambda **kw: [lib.cv_flow_set_option(flow._h, k.encode(), C.c_int32(v)) for k, v in kw.items()]
    opt(big_rows=1, graph_max_rows=1)                                # every pass is a large one and runs eager
    outs = {}
    for streams in (1, 2):
        opt(eager_streams=streams)
        outs[streams] = [m.cpu().clone() for m in flow.inference_batch(items)] + [m.cpu().clone() for m in flow.inference_batch([items[0], items[3]])]
    assert all(torch.isfinite(m).all() and m.abs().max() > 0 for m in outs[1])
    for a, b in zip(outs[1], outs[2]):
        assert a.shape == b.shape and torch.equal(a, b), (a.shape, (a - b).abs().max().item())
