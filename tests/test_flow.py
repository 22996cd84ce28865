"""Stages B3/B4/B5 parity: HIP flow encoder / estimator / full CFM inference vs the oracle (and the reference goldens)."""
import os

import numpy as np
import pytest
import torch

from cosyvoice_amd.flow import CausalMaskedDiffWithXvec
from oracle import flow as OF
from cosyvoice_amd import synthetic as W

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def tiny():
    cfg = W.tiny()[1]
    return cfg, W.make_flow(cfg)


def _inputs(cfg, n_p=7, n_t=13, seed=5):
    g = torch.Generator().manual_seed(seed)
    return dict(prompt_token=torch.randint(0, cfg.vocab, (1, n_p), generator=g, dtype=torch.int32),
                token=torch.randint(0, cfg.vocab, (1, n_t), generator=g, dtype=torch.int32),
                prompt_feat=torch.randn(1, 2 * n_p, 80, generator=g) * 2 - 5,
                embedding=torch.randn(1, cfg.spk_dim, generator=g))


@pytest.mark.parametrize("streaming,ctx", [(False, False), (True, True)])
def test_encoder(lib, tiny, streaming, ctx):
    cfg, sd = tiny
    flow = CausalMaskedDiffWithXvec(sd, cfg, lib=lib)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1, 31, cfg.dim, generator=g)
    if ctx:
        h, mask = flow.encoder(x[:, :-3], torch.tensor([31]), context=x[:, -3:], streaming=streaming)
        ref = OF.encoder(sd, cfg, x[:, :-3], x[:, -3:], streaming)
    else:
        h, mask = flow.encoder(x, torch.tensor([31]), streaming=streaming)
        ref = OF.encoder(sd, cfg, x, None, streaming)
    torch.testing.assert_close(h.cpu(), ref, rtol=2e-4, atol=2e-4)
    assert mask.shape == (1, 1, h.shape[1])


@pytest.mark.parametrize("streaming,T", [(False, 41), (True, 70)])
def test_estimator_boundary(lib, tiny, streaming, T):
    """B3: same call signature and layouts as flow.decoder.estimator; tolerance = the reference's own rtol 1e-2 / atol 1e-4
    (cosyvoice/bin/export_onnx.py:109), tightened to 2e-4."""
    cfg, sd = tiny
    flow = CausalMaskedDiffWithXvec(sd, cfg, lib=lib)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 80, T, generator=g); mu = torch.randn(2, 80, T, generator=g); cond = torch.randn(2, 80, T, generator=g)
    spk = torch.randn(2, 80, generator=g); t = torch.tensor([0.3, 0.7]); mask = torch.ones(2, 1, T)
    out = flow.decoder.estimator(x, mask, mu, t, spk, cond, streaming=streaming)
    ref = OF.estimator(sd, cfg, x, mask, mu, t, spk, cond, streaming)
    torch.testing.assert_close(out.cpu(), ref, rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize("streaming", [False, True])
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_estimator_padded_mask(lib, tiny, streaming, precision):
    """B3 with the reference's PADDED mask (flow/decoder.py:405-494 takes mask [B,1,T]; round 4): rows of different valid length in one
    call.  Valid frames = the oracle's with the same mask (which zeroes the padding before every block and masks it out of the attention); the padding comes
    out exactly zero; and a row's valid frames are BIT-identical to the same row called alone at its own length (the contract of the padded passes).  A mask
    that is not 'ones then zeros' is refused."""
    cfg, sd = tiny
    flow = CausalMaskedDiffWithXvec(sd, cfg, lib=lib, precision=precision)
    g = torch.Generator().manual_seed(12)
    T, lens = 70, (70, 52)
    x = torch.randn(2, 80, T, generator=g); mu = torch.randn(2, 80, T, generator=g); cond = torch.randn(2, 80, T, generator=g)
    spk = torch.randn(2, 80, generator=g); t = torch.tensor([0.3, 0.7])
    mask = torch.zeros(2, 1, T)
    for b, n in enumerate(lens):
        mask[b, 0, :n] = 1
    out = flow.decoder.estimator(x, mask, mu, t, spk, cond, streaming=streaming).cpu()
    assert torch.equal(out[1, :, lens[1]:], torch.zeros(80, T - lens[1]))
    if precision == "fp32":
        ref = OF.estimator(sd, cfg, x, mask, mu, t, spk, cond, streaming)
        torch.testing.assert_close(out, ref, rtol=2e-4, atol=2e-4)
    n = lens[1]                                              # row 1 alone at its own length, as both rows of an unpadded call
    pair = lambda a: torch.stack([a[1], a[1]])
    alone = flow.decoder.estimator(pair(x)[:, :, :n].contiguous(), torch.ones(2, 1, n), pair(mu)[:, :, :n].contiguous(), torch.tensor([0.7, 0.7]), pair(spk),
                                   pair(cond)[:, :, :n].contiguous(), streaming=streaming).cpu()
    assert torch.equal(out[1, :, :n], alone[1])
    bad = mask.clone(); bad[1, 0, 3] = 0
    with pytest.raises(NotImplementedError):
        flow.decoder.estimator(x, bad, mu, t, spk, cond, streaming=streaming)
    # the entry point itself refuses key counts outside 1 .. T (an empty row, a row longer than the tensor) and key counts without a mask
    import ctypes as C
    from cosyvoice_amd._lib import CosyVoiceAmdError
    dev = flow.device
    ten = [lib.hook(a.to(dev).contiguous()) for a in (x, mask, mu, t, spk, cond)]
    outb = lib.hook(torch.empty(2, 80, T, device=dev))
    p = lambda a: C.c_void_p(a.data_ptr())
    for kl, with_mask in (((T + 1, T), True), ((0, T), True), ((T, T), False)):
        with pytest.raises(CosyVoiceAmdError, match="key_len"):
            lib.cv_flow_estimator_masked(flow._h, p(ten[0]), p(ten[1]) if with_mask else None, (C.c_int32 * 2)(*kl), p(ten[2]), p(ten[3]), p(ten[4]), p(ten[5]),
                                         C.c_int32(T), C.c_int32(int(streaming)), p(outb), None)


def test_estimator_like_export_onnx_check(lib, tiny):
    """The reference's only numeric self-check (cosyvoice/bin/export_onnx.py:89-109): 10 random inputs, B = 2, T drawn from [16, 512],
    estimator output of the accelerated backend vs PyTorch at rtol 1e-2 / atol 1e-4 - here the HIP estimator vs the oracle, same
    tolerance, same input recipe (x, mu, cond ~ randn, mask = ones, t ~ rand, spks ~ randn).  The emulator runs 3 draws capped at T = 96."""
    cfg, sd = tiny
    flow = CausalMaskedDiffWithXvec(sd, cfg, lib=lib)
    g = torch.Generator().manual_seed(1986)
    n, t_hi = (3, 96) if lib.emulated else (10, 512)
    for i in range(n):
        T = int(torch.randint(16, t_hi + 1, (1,), generator=g))
        x = torch.randn(2, 80, T, generator=g); mu = torch.randn(2, 80, T, generator=g); cond = torch.randn(2, 80, T, generator=g)
        spk = torch.randn(2, 80, generator=g); t = torch.rand(2, generator=g); mask = torch.ones(2, 1, T)
        out = flow.decoder.estimator(x, mask, mu, t, spk, cond, streaming=bool(i % 2))
        ref = OF.estimator(sd, cfg, x, mask, mu, t, spk, cond, bool(i % 2))
        torch.testing.assert_close(out.cpu(), ref, rtol=1e-2, atol=1e-4)


@pytest.mark.parametrize("streaming,finalize", [(False, True), (True, False)])
def test_inference(lib, tiny, streaming, finalize):
    cfg, sd = tiny
    flow = CausalMaskedDiffWithXvec(sd, cfg, lib=lib, n_timesteps=4)
    u = _inputs(cfg)
    t = lambda n: torch.tensor([n], dtype=torch.int32)
    mel, _ = flow.inference(token=u["token"], token_len=t(13), prompt_token=u["prompt_token"], prompt_token_len=t(7), prompt_feat=u["prompt_feat"],
                            prompt_feat_len=t(14), embedding=u["embedding"], streaming=streaming, finalize=finalize)
    ref = OF.inference(sd, cfg, u["token"], u["prompt_token"], u["prompt_feat"], u["embedding"], streaming=streaming, finalize=finalize, n_timesteps=4)
    assert mel.shape == ref.shape
    torch.testing.assert_close(mel.cpu(), ref, rtol=1e-3, atol=1e-3)


def test_streaming_prefix_invariance(lib, tiny):
    """The reference's own invariance check (flow/flow.py:417-443 __main__): with chunk-causal masks the mel of a prefix does not
    change when more tokens arrive (what makes CosyVoice2's re-run-the-flow streaming correct, cli/model.py:292-303)."""
    cfg, sd = tiny
    flow = CausalMaskedDiffWithXvec(sd, cfg, lib=lib, n_timesteps=2)
    u = _inputs(cfg, n_p=25, n_t=60, seed=9)
    t = lambda n: torch.tensor([n], dtype=torch.int32)
    kw = dict(prompt_token=u["prompt_token"], prompt_token_len=t(25), prompt_feat=u["prompt_feat"], prompt_feat_len=t(50), embedding=u["embedding"], streaming=True)
    a, _ = flow.inference(token=u["token"][:, :28], token_len=t(28), finalize=False, **kw)          # 25+28-3 = 50 tokens = 2 chunks
    b, _ = flow.inference(token=u["token"], token_len=t(60), finalize=True, **kw)
    torch.testing.assert_close(a.cpu(), b.cpu()[:, :, : a.shape[2]], rtol=1e-4, atol=1e-4)


def test_matches_reference_golden(lib):
    """End to end against golden vectors produced by the REAL reference (dim 512, see tests/golden/make_golden.py)."""
    g = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(G, "flow_small.npz")).items()}
    cfg = W.ref_small_flow()
    flow = CausalMaskedDiffWithXvec(W.make_flow(cfg), cfg, lib=lib)
    mask = torch.ones(2, 1, g["est_x"].shape[-1])
    out = flow.decoder.estimator(g["est_x"], mask, g["est_mu"], g["est_t"], g["est_spk"], g["est_cond"], streaming=True)
    torch.testing.assert_close(out.cpu(), g["est_stream"], rtol=1e-2, atol=1e-4)
    t = lambda n: torch.tensor([n], dtype=torch.int32)
    mel, _ = flow.inference(token=g["token"], token_len=t(16), prompt_token=g["prompt_token"], prompt_token_len=t(9), prompt_feat=g["prompt_feat"],
                            prompt_feat_len=t(18), embedding=g["embedding"], streaming=False, finalize=True)
    torch.testing.assert_close(mel.cpu(), g["mel_full"], rtol=2e-3, atol=2e-3)


def test_graph_replay_is_identical_to_eager(lib, tiny):
    """cv_flow_inference captures the whole Euler solve into a hipGraph the 2nd time a (T, steps, streaming) key is seen and
    replays it afterwards: eager, capture+launch and replay must give the same bits, also after another shape ran in between."""
    cfg, sd = tiny
    flow = CausalMaskedDiffWithXvec(sd, cfg, lib=lib, n_timesteps=3)
    u = _inputs(cfg)
    t = lambda n: torch.tensor([n], dtype=torch.int32)
    kw = dict(prompt_token=u["prompt_token"], prompt_token_len=t(7), prompt_feat=u["prompt_feat"], prompt_feat_len=t(14), embedding=u["embedding"],
              streaming=False, finalize=True)
    outs = [flow.inference(token=u["token"], token_len=t(13), **kw)[0].cpu().clone() for _ in range(2)]
    other = flow.inference(token=u["token"][:, :9], token_len=t(9), **kw)[0].cpu().clone()            # different T in between
    outs += [flow.inference(token=u["token"], token_len=t(13), **kw)[0].cpu().clone() for _ in range(2)]
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    ref = OF.inference(sd, cfg, u["token"], u["prompt_token"], u["prompt_feat"], u["embedding"], streaming=False, finalize=True, n_timesteps=3)
    torch.testing.assert_close(outs[0], ref, rtol=1e-3, atol=1e-3)
    assert other.shape[2] == 18


def test_graph_cache_evicts_least_recently_used(lib, tiny):
    """Round 3: the Euler-solve graph cache holds `graph_cap` shapes (32; 2 here) and evicts the least recently used one when a new shape is captured - before, the
    ninth shape dropped ALL graphs and none of them was ever captured again (a serving scheduler's shared passes see ~16 shapes per lane).  Three shapes in rotation:
    results never change, an evicted shape is captured again on its second new sighting, a shape that stays in use is not."""
    import ctypes as C
    if not lib.emulated:
        pytest.skip("host logic of the stage driver (which graph is kept): exercised under the emulator; graph replay itself runs on hardware in the tests around it")
    cfg, sd = tiny
    flow = CausalMaskedDiffWithXvec(sd, cfg, lib=lib, n_timesteps=2)
    lib.cv_flow_set_option(flow._h, b"graph_cap", C.c_int32(2))
    u = _inputs(cfg)
    t = lambda n: torch.tensor([n], dtype=torch.int32)
    kw = dict(prompt_token=u["prompt_token"], prompt_token_len=t(7), prompt_feat=u["prompt_feat"], prompt_feat_len=t(14), embedding=u["embedding"],
              streaming=False, finalize=True)

    def stat(name):
        v = C.c_int64(0)
        lib.cv_flow_get_stat(flow._h, name, C.byref(v))
        return v.value
    run = lambda n: flow.inference(token=u["token"][:, :n], token_len=t(n), **kw)[0].cpu().clone()
    flow.inference(token=u["token"], token_len=t(13), **kw)                       # the workspaces reach their final size before anything is captured
    first = {}
    for n in (13, 11, 13, 11):                                                    # both shapes captured on their second sighting
        o = run(n); first.setdefault(n, o); assert torch.equal(o, first[n])
    assert stat(b"graphs_cached") == 2 and stat(b"graph_captures") == 2
    for n in (9, 13, 9):                                                          # third shape: captured on ITS second sighting, evicting 11 (13 was used since)
        o = run(n); first.setdefault(n, o); assert torch.equal(o, first[n])
    assert stat(b"graphs_cached") == 2 and stat(b"graph_captures") == 3
    assert torch.equal(run(13), first[13]) and stat(b"graph_captures") == 3      # 13 still replays its graph
    assert torch.equal(run(11), first[11]) and stat(b"graph_captures") == 3      # 11 was evicted: eager now (first new sighting) ...
    assert torch.equal(run(11), first[11]) and stat(b"graph_captures") == 4      # ... captured again on the second, evicting the least recently used of {9, 13}
    assert torch.equal(run(9), first[9]) and torch.equal(run(13), first[13]) and stat(b"graphs_cached") == 2


def test_no_prompt_and_single_token(lib, tiny):
    """Edge cases of flow.inference: empty prompt (mel_len1 = 0, cross-lingual style call) and the shortest legal input."""
    cfg, sd = tiny
    flow = CausalMaskedDiffWithXvec(sd, cfg, lib=lib, n_timesteps=2)
    g = torch.Generator().manual_seed(3)
    emb = torch.randn(1, cfg.spk_dim, generator=g)
    t = lambda n: torch.tensor([n], dtype=torch.int32)
    for n in (1, 6):
        tok = torch.randint(0, cfg.vocab, (1, n), generator=g, dtype=torch.int32)
        mel, _ = flow.inference(token=tok, token_len=t(n), prompt_token=torch.zeros(1, 0, dtype=torch.int32), prompt_token_len=t(0),
                                prompt_feat=torch.zeros(1, 0, 80), prompt_feat_len=t(0), embedding=emb, streaming=False, finalize=True)
        ref = OF.inference(sd, cfg, tok, torch.zeros(1, 0, dtype=torch.int32), torch.zeros(1, 0, 80), emb, streaming=False, finalize=True, n_timesteps=2)
        assert mel.shape == (1, 80, 2 * n)
        torch.testing.assert_close(mel.cpu(), ref, rtol=1e-3, atol=1e-3)
    with pytest.raises(ValueError):           # streaming call with fewer tokens than the look-ahead: nothing to generate
        flow.inference(token=torch.zeros(1, 2, dtype=torch.int32), token_len=t(2), prompt_token=torch.zeros(1, 0, dtype=torch.int32), prompt_token_len=t(0),
                       prompt_feat=torch.zeros(1, 0, 80), prompt_feat_len=t(0), embedding=emb, streaming=True, finalize=False)


# ---- "bf16" precision mode: Linear / Conv1d operands rounded to bf16, bf16 MFMA, fp32 accumulate ---------------------------
# What can and cannot be asserted.  Each product is exact given its rounded operands (tests/test_ops.py::test_linear_bf16_mfma),
# and a shallow path reproduces the rounding-mirroring oracle tightly (the encoder below).  Through the deep estimator the
# rounding noise is not a smooth function of the input: a 1e-6 relative perturbation of x (the size of the summation-order
# differences between any two fp32 implementations) re-draws ~5 % of the downstream rounding decisions, and the oracle's own bf16
# result moves by as much as the whole bf16-vs-fp32 gap (measured: mean 2.7e-3 vs 2.5e-3 on this fixture).  So there the check is
# statistical: the product's distance to the fp32 oracle must be the distance the mirroring oracle has - no more.
# Stated tolerance of the mode: log-mel max |mel_bf16 - mel_fp32| <= 5e-2 (mean <= 1e-2 added here) against the
# reference golden, and waveform SNR >= 30 dB through HiFT with an identical source (the reference accepts its own fp16/TensorRT
# estimator at rtol 1e-2 per call, cosyvoice/bin/export_onnx.py:109).
def test_bf16_mode_encoder_matches_rounding_oracle(lib, tiny):
    cfg, sd = tiny
    flow = CausalMaskedDiffWithXvec(sd, cfg, lib=lib, precision="bf16")
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1, 31, cfg.dim, generator=g)
    h, _ = flow.encoder(x, torch.tensor([31]), streaming=False)
    with OF.bf16_act():
        ref = OF.encoder(sd, cfg, x, None, False)
    ref32 = OF.encoder(sd, cfg, x, None, False)
    # (emulator: identical libm, max 7e-5.  MI355X: the device's exp/silu differ from the host's in the last ulp, ~1 % of the
    # outputs see a flipped rounding upstream, max 5e-3 measured)
    e_mirror, e_fp32 = (h.cpu() - ref).abs(), (h.cpu() - ref32).abs()
    assert e_mirror.max() < 1e-2, e_mirror.max().item()
    assert e_mirror.mean() < 0.25 * e_fp32.mean(), (e_mirror.mean().item(), e_fp32.mean().item())   # the mirror explains the bf16-vs-fp32 gap


def test_bf16_mode_estimator_noise_level(lib, tiny):
    cfg, sd = tiny
    flow = CausalMaskedDiffWithXvec(sd, cfg, lib=lib, n_timesteps=3, precision="bf16")
    g = torch.Generator().manual_seed(2)
    T = 41
    x = torch.randn(2, 80, T, generator=g); mu = torch.randn(2, 80, T, generator=g); cond = torch.randn(2, 80, T, generator=g)
    spk = torch.randn(2, 80, generator=g); t = torch.tensor([0.3, 0.7]); mask = torch.ones(2, 1, T)
    out = flow.decoder.estimator(x, mask, mu, t, spk, cond, streaming=False).cpu()
    with OF.bf16_act():
        mirror = OF.estimator(sd, cfg, x, mask, mu, t, spk, cond, False)
    ref32 = OF.estimator(sd, cfg, x, mask, mu, t, spk, cond, False)
    e_prod, e_mirror = (out - ref32).abs(), (mirror - ref32).abs()
    assert e_prod.mean() < 1.5 * e_mirror.mean() + 1e-4, (e_prod.mean().item(), e_mirror.mean().item())
    assert e_prod.max() < 2.5 * e_mirror.max() + 1e-3, (e_prod.max().item(), e_mirror.max().item())
    assert e_prod.mean() > 0.2 * e_mirror.mean()                                  # ... and the mode is really on
    u = _inputs(cfg)
    n = lambda k: torch.tensor([k], dtype=torch.int32)
    mel, _ = flow.inference(token=u["token"], token_len=n(13), prompt_token=u["prompt_token"], prompt_token_len=n(7), prompt_feat=u["prompt_feat"],
                            prompt_feat_len=n(14), embedding=u["embedding"], streaming=False, finalize=True)
    refm = OF.inference(sd, cfg, u["token"], u["prompt_token"], u["prompt_feat"], u["embedding"], streaming=False, finalize=True, n_timesteps=3)
    d = (mel.cpu() - refm).abs()
    assert d.max() < 5e-2 and d.mean() < 1e-2, (d.max().item(), d.mean().item())


def test_bf16_mode_vs_reference_golden(lib):
    """bf16 mode against the REAL reference's fp32 output (dim 512 golden): the stated tolerance of the mode."""
    g = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(G, "flow_small.npz")).items()}
    cfg = W.ref_small_flow()
    flow = CausalMaskedDiffWithXvec(W.make_flow(cfg), cfg, lib=lib, precision="bf16")
    t = lambda n: torch.tensor([n], dtype=torch.int32)
    mel, _ = flow.inference(token=g["token"], token_len=t(16), prompt_token=g["prompt_token"], prompt_token_len=t(9), prompt_feat=g["prompt_feat"],
                            prompt_feat_len=t(18), embedding=g["embedding"], streaming=False, finalize=True)
    d = (mel.cpu() - g["mel_full"]).abs()
    print("bf16 mode vs reference golden: max %.3e mean %.3e" % (d.max().item(), d.mean().item()))
    assert d.max().item() < 5e-2 and d.mean().item() < 1e-2, (d.max().item(), d.mean().item())


def test_bf16_mode_waveform_snr(lib, tiny):
    """waveform SNR >= 30 dB vs the fp32 path given an identical harmonic source.  Mel of one request in both precisions
    -> the same HiFT (fp32) decode with the source computed once from the fp32 mel."""
    from cosyvoice_amd.hift import HiFTGenerator
    cfg, sd = tiny
    u = _inputs(cfg, n_p=3, n_t=5)                  # 10 mel frames: the three HiFT passes dominate the emulator time
    t = lambda n: torch.tensor([n], dtype=torch.int32)
    mels = {}
    for prec in ("fp32", "bf16"):
        flow = CausalMaskedDiffWithXvec(sd, cfg, lib=lib, precision=prec, n_timesteps=4)
        mels[prec], _ = flow.inference(token=u["token"], token_len=t(5), prompt_token=u["prompt_token"], prompt_token_len=t(3), prompt_feat=u["prompt_feat"],
                                       prompt_feat_len=t(6), embedding=u["embedding"], streaming=False, finalize=True)
    hc = W.tiny()[2]
    hift = HiFTGenerator(W.make_hift(hc), hc, lib=lib)
    _, s = hift.inference(mels["fp32"])            # (speech, source): the harmonic source of the fp32 mel, reused for both decodes
    w32 = hift.decode(mels["fp32"], s).cpu()
    w16 = hift.decode(mels["bf16"], s).cpu()
    snr = 10 * torch.log10(w32.pow(2).sum() / (w32 - w16).pow(2).sum().clamp_min(1e-20))
    print("bf16 flow -> waveform SNR %.1f dB" % snr.item())
    assert snr.item() >= 30.0, snr.item()


def test_estimator_module_contract(lib):
    """B3: EstimatorModule is an nn.Module (the branch of the reference's forward_estimator, flow_matching.py:126-128), takes the reference's
    layouts, retains nothing, and reproduces what the REAL ConditionalCFM.solve_euler produced with the REAL estimator
    (tests/golden/estimator_module.npz, generated by running this module inside the reference's own solve_euler / forward_estimator)."""
    import numpy as np, os
    from cosyvoice_amd.flow import EstimatorModule
    from oracle import flow as OF2
    g = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(os.path.dirname(__file__), "golden", "estimator_module.npz")).items()}
    cfg = W.ref_small_flow()
    sd = W.make_flow(cfg)
    flow = CausalMaskedDiffWithXvec(sd, cfg, lib=lib)
    est = EstimatorModule(flow)
    assert isinstance(est, torch.nn.Module) and len(list(est.parameters())) == 0
    T = g["mu"].shape[2]
    # the reference's solve_euler restated around the module (flow_matching.py:71-124): 3 steps, cosine schedule, CFG 0.7, buffer reuse
    x = flow.decoder.rand_noise[:, :, :T].clone()
    t_span = 1 - torch.cos(torch.linspace(0, 1, 4) * 0.5 * torch.pi)
    t, dt = t_span[0], t_span[1] - t_span[0]
    x_in, mask_in, mu_in = torch.zeros(2, 80, T), torch.ones(2, 1, T), torch.zeros(2, 80, T)
    t_in, spks_in, cond_in = torch.zeros(2), torch.zeros(2, 80), torch.zeros(2, 80, T)
    for step in range(1, 4):
        x_in[:] = x; mu_in[0] = g["mu"][0]; t_in[:] = t; spks_in[0] = g["spks"][0]; cond_in[0] = g["cond"][0]
        d = est(x_in, mask_in, mu_in, t_in, spks_in, cond_in, streaming=False)
        assert d.dtype == x_in.dtype and d.device == x_in.device and d.shape == (2, 80, T)
        d0, d1 = d[0:1].cpu(), d[1:2].cpu()
        x = x + dt * ((1.0 + 0.7) * d0 - 0.7 * d1)
        t = t + dt
        if step < 3:
            dt = t_span[step + 1] - t
    torch.testing.assert_close(x, g["out"], rtol=1e-3, atol=1e-3)


@pytest.mark.gpu
def test_estimator_engine_raw_addresses_on_a_real_stream(hip_lib):
    """B3, the NON-Module branch of forward_estimator (flow/flow_matching.py:129-153) on the MI355X: the statements of that branch - acquire, `with stream`, six
    input shapes, seven raw `data_ptr()`s with the output aliased on x, `execute_async_v3(torch.cuda.current_stream().cuda_stream)`, synchronize, release - restated
    here because /root/reference does not travel to the GPU box (the REAL branch drives the same object under the emulator, tests/test_dropin_reference.py).  The raw
    hipStream_t is the context's own side stream, the result must be the nn.Module form's (EstimatorModule), bit for bit, and x must hold it (the aliased output)."""
    from cosyvoice_amd.flow import EstimatorEngine, EstimatorModule
    cfg = W.ref_small_flow()
    flow = CausalMaskedDiffWithXvec(W.make_flow(cfg), cfg, lib=hip_lib)
    eng, mod = EstimatorEngine(flow), EstimatorModule(flow)
    dev = flow.device
    g = torch.Generator().manual_seed(31)
    for T in (48, 131):
        x, mu, cond = (torch.randn(2, 80, T, generator=g).to(dev) for _ in range(3))
        spks, t, mask = torch.randn(2, 80, generator=g).to(dev), torch.tensor([0.25, 0.25]).to(dev), torch.ones(2, 1, T, device=dev)
        want = mod(x.clone(), mask, mu, t, spks, cond, streaming=False)
        x0 = x.clone()
        [ctx, stream], engine = eng.acquire_estimator()
        torch.cuda.current_stream().synchronize()
        with stream:
            side = torch.cuda.current_stream().cuda_stream
            assert side != 0 and side != torch.cuda.default_stream().cuda_stream          # a real, non-default hipStream_t crosses the C ABI
            for name, shape in (("x", (2, 80, T)), ("mask", (2, 1, T)), ("mu", (2, 80, T)), ("t", (2,)), ("spks", (2, 80)), ("cond", (2, 80, T))):
                ctx.set_input_shape(name, shape)
            ptrs = [x.contiguous().data_ptr(), mask.contiguous().data_ptr(), mu.contiguous().data_ptr(), t.contiguous().data_ptr(), spks.contiguous().data_ptr(),
                    cond.contiguous().data_ptr(), x.data_ptr()]
            for i, p in enumerate(ptrs):
                ctx.set_tensor_address(engine.get_tensor_name(i), p)
            assert ctx.execute_async_v3(side) is True
            torch.cuda.current_stream().synchronize()
        eng.release_estimator(ctx, stream)
        assert eng._pool.qsize() == 1
        assert torch.equal(x, want) and not torch.equal(x, x0)
        ref = OF.estimator(W.make_flow(cfg), cfg, x0.cpu(), mask.cpu(), mu.cpu(), t.cpu(), spks.cpu(), cond.cpu(), False)
        torch.testing.assert_close(x.cpu(), ref, rtol=2e-4, atol=2e-4)


_X = pytest.mark.experiments          # attn_flow_kernel (rounds 2-5) in its workgroup shapes: CV_BUILD_EXPERIMENTS builds only; the LN-GEMM tile shapes ride on the default attention
@pytest.mark.parametrize("tile,waves,kt,ks", [(0, 0, 0, 0), (1, 0, 0, 0), (2, 0, 0, 0), (3, 0, 0, 0), (4, 0, 0, 0)] + [pytest.param(*v, marks=_X) for v in
                                              ((0, 4, 2, 2), (1, 2, 1, 1), (2, 4, 1, 1), (3, 2, 2, 1), (4, 4, 2, 1), (0, 4, 1, 3), (0, 4, 1, 4))])
def test_fused_transformer_blocks_match_unfused(lib, tile, waves, kt, ks):
    """bf16 mode: the fused pipeline of the estimator's transformer blocks (flow_fused.h: LayerNorm in the GEMM prologue, bf16 Q / K / V^T /
    attention output / FF hidden between kernels, bf16-in flash attention) rounds the same operands at the same points as the unfused
    launches.  Two heads, a time axis that is not a
    multiple of 4 (the V^T epilogue), several key tiles, both mask modes, every LN-GEMM tile shape and both attention workgroup sizes."""
    import ctypes as C
    import dataclasses
    cfg = dataclasses.replace(W.tiny()[1], est_ch=128, est_heads=2, est_mid=1, chunk=13)
    sd = W.make_flow(cfg)
    flow = CausalMaskedDiffWithXvec(sd, cfg, lib=lib, precision="bf16")
    g = torch.Generator().manual_seed(3)
    for T in (45, 150):
        x = torch.randn(2, 80, T, generator=g); mu = torch.randn(2, 80, T, generator=g); cond = torch.randn(2, 80, T, generator=g)
        spk = torch.randn(2, 80, generator=g); t = torch.tensor([0.4, 0.4]); mask = torch.ones(2, 1, T)
        for streaming in (False, True):
            outs = []
            for fused in (0, 1):
                lib.cv_flow_set_option(flow._h, b"fused", C.c_int32(fused))
                lib.cv_flow_set_option(flow._h, b"flow_tile", C.c_int32(tile))
                lib.cv_flow_set_option(flow._h, b"attn32", C.c_int32(int(waves == 0)))      # waves == 0: the round-6 attention (attn_flow32_kernel, the default); else attn_flow_kernel<waves, kt, ks>
                if waves > 0:
                    lib.cv_flow_set_option(flow._h, b"attn_waves", C.c_int32(waves))
                    lib.cv_flow_set_option(flow._h, b"attn_kt", C.c_int32(kt))
                    lib.cv_flow_set_option(flow._h, b"attn_ks", C.c_int32(ks))
                outs.append(flow.decoder.estimator(x, mask, mu, t, spk, cond, streaming=streaming).cpu())
            ref = OF.estimator(sd, cfg, x, mask, mu, t, spk, cond, streaming)
            # A bf16 computation is not a smooth function of its inputs (DESIGN.md section 5 "bf16 mode" iii): where the LayerNorm statistics of the
            # two paths differ in the last ulp and flip one rounding, the outputs part by as much as bf16 itself costs; where nothing flips they
            # are bit-identical (same rounding points, same MFMA accumulation order).  So the criterion is the mode's own: the fused result is
            # as close to the fp32 oracle as the unfused one.
            e_un, e_fu = (outs[0] - ref).abs(), (outs[1] - ref).abs()
            assert torch.isfinite(outs[1]).all()
            assert e_fu.mean().item() < 1.3 * e_un.mean().item() + 1e-5 and e_fu.max().item() < 1.6 * e_un.max().item() + 1e-4, \
                (T, streaming, e_fu.mean().item(), e_un.mean().item(), e_fu.max().item(), e_un.max().item())


def test_two_n_tiles_per_workgroup_is_bit_identical(lib):
    """bf16 mode, round 3: flow_gemm_kernel<.., NTILE = 2> (option flow_ntile = 2: a workgroup of the LayerNorm-prologue GEMMs walks two 64-column tiles
    of the same rows - LayerNorm and the A tile once, the second tile's weights in flight under the first's MFMAs) computes every output element
    exactly as the single-tile launch does: Q | K rows, the V^T section (a pair that straddles n_row and one entirely inside it), FF1 + GELU; a
    ragged last row tile, both mask modes."""
    import ctypes as C
    import dataclasses
    cfg = dataclasses.replace(W.tiny()[1], est_ch=128, est_heads=2, est_mid=1, chunk=13)
    sd = W.make_flow(cfg)
    flow = CausalMaskedDiffWithXvec(sd, cfg, lib=lib, precision="bf16")
    g = torch.Generator().manual_seed(4)
    for T in (45, 150):
        x = torch.randn(2, 80, T, generator=g); mu = torch.randn(2, 80, T, generator=g); cond = torch.randn(2, 80, T, generator=g)
        spk = torch.randn(2, 80, generator=g); t = torch.tensor([0.3, 0.3]); mask = torch.ones(2, 1, T)
        for streaming in (False, True):
            outs = []
            for ntile in (1, 2):
                lib.cv_flow_set_option(flow._h, b"flow_ntile", C.c_int32(ntile))
                outs.append(flow.decoder.estimator(x, mask, mu, t, spk, cond, streaming=streaming).cpu().clone())
            assert torch.isfinite(outs[0]).all() and outs[0].abs().max() > 0
            assert torch.equal(outs[0], outs[1]), (T, streaming, (outs[0] - outs[1]).abs().max().item())
    lib.cv_flow_set_option(flow._h, b"flow_ntile", C.c_int32(0))


@pytest.mark.parametrize("est_blocks", [1, 3])
@pytest.mark.experiments
def test_fused_tail_matches_five_launch_blocks(lib, est_blocks):
    """bf16 mode, round 3: everything after a block's attention as ONE launch per 16-row band (flow_tail.h: out-projection + residual -> LayerNorm ->
    FF1 + GELU -> FF2 + residual -> the next block's LayerNorm -> Q | K | V^T) against the five-launch form of round 2 (option fused_tail = 0) and
    the fp32 oracle.  The rounding points are the same, so the criterion is the mode's own (see the test above): as close to the oracle as the
    five-launch path, and close to it.  est_blocks = 3 exercises the chained variant (tail + next QKV, twice) and the closing one; a time axis that is
    not a multiple of 16 (ragged last band) nor of 4 (V^T groups straddling), CFG batch rows, both mask modes, graph replay through inference()."""
    import ctypes as C
    import dataclasses
    cfg = dataclasses.replace(W.tiny()[1], est_blocks=est_blocks, est_mid=1, chunk=13, n_timesteps=2)
    sd = W.make_flow(cfg)
    flow = CausalMaskedDiffWithXvec(sd, cfg, lib=lib, precision="bf16")
    assert any(k.endswith(".tail") for k in flow._tensors), "weights.pack_flow did not produce the fragment-ordered tail streams"
    g = torch.Generator().manual_seed(5)
    for T in (45, 150):
        x = torch.randn(2, 80, T, generator=g); mu = torch.randn(2, 80, T, generator=g); cond = torch.randn(2, 80, T, generator=g)
        spk = torch.randn(2, 80, generator=g); t = torch.tensor([0.4, 0.4]); mask = torch.ones(2, 1, T)
        for streaming in (False, True):
            outs = []
            for tail in (0, 1):
                lib.cv_flow_set_option(flow._h, b"fused_tail", C.c_int32(tail))
                outs.append(flow.decoder.estimator(x, mask, mu, t, spk, cond, streaming=streaming).cpu())
            ref = OF.estimator(sd, cfg, x, mask, mu, t, spk, cond, streaming)
            e_5, e_t, d = (outs[0] - ref).abs(), (outs[1] - ref).abs(), (outs[1] - outs[0]).abs()
            assert torch.isfinite(outs[1]).all()
            assert e_t.mean().item() < 1.3 * e_5.mean().item() + 1e-5 and e_t.max().item() < 1.6 * e_5.max().item() + 1e-4, \
                (T, streaming, e_t.mean().item(), e_5.mean().item(), e_t.max().item(), e_5.max().item())
            assert d.mean().item() < 1.5 * e_5.mean().item() + 1e-5, (T, streaming, d.mean().item(), e_5.mean().item())
            if lib.emulated:        # same rounding points, same k order in every MFMA chain, same bias / residual order: under one libm the two forms agree bit for bit
                assert torch.equal(outs[0], outs[1])
    # the whole inference (encoder + Euler loop, captured graph on the second sighting) with the tail on stays inside the mode's stated tolerance
    lib.cv_flow_set_option(flow._h, b"fused_tail", C.c_int32(1))
    u = _inputs(cfg)
    n = lambda k: torch.tensor([k], dtype=torch.int32)
    kw = dict(token=u["token"], token_len=n(13), prompt_token=u["prompt_token"], prompt_token_len=n(7), prompt_feat=u["prompt_feat"], prompt_feat_len=n(14),
              embedding=u["embedding"], streaming=False, finalize=True)
    mels = [flow.inference(**kw)[0].cpu() for _ in range(3)]
    assert torch.equal(mels[0], mels[1]) and torch.equal(mels[1], mels[2])                       # eager == graph replay
    refm = OF.inference(sd, cfg, u["token"], u["prompt_token"], u["prompt_feat"], u["embedding"], streaming=False, finalize=True, n_timesteps=2)
    dm = (mels[0] - refm).abs()
    assert dm.max() < 5e-2 and dm.mean() < 1e-2, (dm.max().item(), dm.mean().item())


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_flow_inference_batch_equals_single(lib, precision):
    """cv_flow_inference_batch: utterances of equal shape solved in one pass (estimator batch rows = 2 x utterances) give, each, exactly the mel
    `inference()` gives for it alone - the reference's contract for batched flow (flow/flow.py:246)."""
    import dataclasses
    cfg = dataclasses.replace(W.tiny()[1], n_timesteps=2)
    sd = W.make_flow(cfg)
    flow = CausalMaskedDiffWithXvec(sd, cfg, lib=lib, precision=precision)
    g = torch.Generator().manual_seed(71)
    n = lambda k: torch.tensor([k], dtype=torch.int32)
    items = []
    for _ in range(3):
        items.append(dict(token=torch.randint(0, cfg.vocab, (1, 9), generator=g, dtype=torch.int32), prompt_token=torch.randint(0, cfg.vocab, (1, 5), generator=g, dtype=torch.int32),
                          prompt_feat=torch.randn(1, 10, cfg.mel, generator=g) * 2 - 5, embedding=torch.randn(1, cfg.spk_dim, generator=g)))
    alone = [flow.inference(token=it["token"], token_len=n(9), prompt_token=it["prompt_token"], prompt_token_len=n(5), prompt_feat=it["prompt_feat"],
                            prompt_feat_len=n(10), embedding=it["embedding"], streaming=False, finalize=True)[0].cpu() for it in items]
    assert not torch.equal(alone[0], alone[1])
    for rep in range(3):                                          # the third call replays the captured graph of the batched solve
        got = flow.inference_batch(items)
        for a, b in zip(alone, got):
            assert torch.equal(a, b.cpu())
    got2 = flow.inference_batch(items[1:])                        # another batch size on the same handle, then the single path again
    assert torch.equal(alone[2], got2[1].cpu())
    again = flow.inference(token=items[0]["token"], token_len=n(9), prompt_token=items[0]["prompt_token"], prompt_token_len=n(5), prompt_feat=items[0]["prompt_feat"],
                           prompt_feat_len=n(10), embedding=items[0]["embedding"], streaming=False, finalize=True)[0].cpu()
    assert torch.equal(again, alone[0])
    # a LONGER single request on the handle that served batches: the per-utterance buffers must follow its token count, not the batch's row count
    tok_long = torch.randint(0, cfg.vocab, (1, 9 * 3), generator=g, dtype=torch.int32)
    it = items[0]
    long1 = flow.inference(token=tok_long, token_len=n(27), prompt_token=it["prompt_token"], prompt_token_len=n(5), prompt_feat=it["prompt_feat"], prompt_feat_len=n(10),
                           embedding=it["embedding"], streaming=False, finalize=True)[0].cpu()
    fresh = CausalMaskedDiffWithXvec(sd, cfg, lib=lib, precision=precision)
    long2 = fresh.inference(token=tok_long, token_len=n(27), prompt_token=it["prompt_token"], prompt_token_len=n(5), prompt_feat=it["prompt_feat"], prompt_feat_len=n(10),
                            embedding=it["embedding"], streaming=False, finalize=True)[0].cpu()
    assert torch.equal(long1, long2)


@pytest.mark.parametrize("streaming,finalize", [(True, False), (False, False), (True, True)])
def test_flow_inference_batch_encoder_together(lib, streaming, finalize):
    """Round 4: the conformer encoder runs ONCE over the stacked rows of an equal-shape pass (option enc_batch, default on; flow.hip flow_encoder nu > 1):
    chunk masks, the pre-lookahead context rows that follow every utterance's encoded tokens (finalize = False), and the per-utterance
    relative-position attention all give each utterance the mel of its single call; enc_batch = 0 (the per-utterance loop) gives the same bits."""
    import dataclasses
    cfg = dataclasses.replace(W.tiny()[1], n_timesteps=2)
    sd = W.make_flow(cfg)
    flow = CausalMaskedDiffWithXvec(sd, cfg, lib=lib, precision="bf16")
    g = torch.Generator().manual_seed(79)
    n = lambda k: torch.tensor([k], dtype=torch.int32)
    items = [dict(token=torch.randint(0, cfg.vocab, (1, 11), generator=g, dtype=torch.int32), prompt_token=torch.randint(0, cfg.vocab, (1, 6), generator=g, dtype=torch.int32),
                  prompt_feat=torch.randn(1, 12, cfg.mel, generator=g) * 2 - 5, embedding=torch.randn(1, cfg.spk_dim, generator=g)) for _ in range(3)]
    alone = [flow.inference(token=it["token"], token_len=n(11), prompt_token=it["prompt_token"], prompt_token_len=n(6), prompt_feat=it["prompt_feat"],
                            prompt_feat_len=n(12), embedding=it["embedding"], streaming=streaming, finalize=finalize)[0].cpu() for it in items]
    assert not torch.equal(alone[0], alone[1])
    for on in (1, 0, 1):
        flow.lib.cv_flow_set_option(flow._h, b"enc_batch", on)
        got = flow.inference_batch(items, streaming=streaming, finalize=finalize)
        for a, b in zip(alone, got):
            assert a.shape == b.shape and torch.equal(a, b.cpu())


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("streaming,finalize", [(False, True), (True, False)])
def test_flow_inference_ragged_equals_single(lib, precision, streaming, finalize):
    """cv_flow_inference_ragged: utterances of DIFFERENT lengths (tokens, prompt tokens, prompt frames) padded into one pass give, each, exactly the mel
    `inference()` gives for it alone - the reference's contract for its masked, padded batch (flow/flow.py:236-281).  Convolutions are causal, norms
    per row, attention takes the key count of every batch row; then equal-shape and single calls on the same handle."""
    import dataclasses
    cfg = dataclasses.replace(W.tiny()[1], n_timesteps=2)
    sd = W.make_flow(cfg)
    flow = CausalMaskedDiffWithXvec(sd, cfg, lib=lib, precision=precision)
    g = torch.Generator().manual_seed(73)
    n = lambda k: torch.tensor([k], dtype=torch.int32)
    shapes = [(14, 5, 10), (9, 7, 12), (21, 4, 6)]                 # (new tokens, prompt tokens, prompt frames)
    items = [dict(token=torch.randint(0, cfg.vocab, (1, a), generator=g, dtype=torch.int32), prompt_token=torch.randint(0, cfg.vocab, (1, b), generator=g, dtype=torch.int32),
                  prompt_feat=torch.randn(1, c, cfg.mel, generator=g) * 2 - 5, embedding=torch.randn(1, cfg.spk_dim, generator=g)) for a, b, c in shapes]
    single = lambda it: flow.inference(token=it["token"], token_len=n(it["token"].shape[1]), prompt_token=it["prompt_token"], prompt_token_len=n(it["prompt_token"].shape[1]),
                                       prompt_feat=it["prompt_feat"], prompt_feat_len=n(it["prompt_feat"].shape[1]), embedding=it["embedding"],
                                       streaming=streaming, finalize=finalize)[0].cpu()
    alone = [single(it) for it in items]
    assert len({a.shape[2] for a in alone}) == 3
    for rep in range(3):                                          # the third call replays the captured graph of the padded solve
        got = flow.inference_batch(items, streaming=streaming, finalize=finalize)
        for a, b in zip(alone, got):
            assert a.shape == b.shape and torch.equal(a, b.cpu())
    got = flow.inference_batch(items[::-1], streaming=streaming, finalize=finalize)       # another order: another padding pattern
    for a, b in zip(alone[::-1], got):
        assert torch.equal(a, b.cpu())
    assert torch.equal(single(items[0]), alone[0])
    same = flow.inference_batch([items[1], items[1]], streaming=streaming, finalize=finalize)   # the equal-shape path after a padded one
    assert torch.equal(same[0].cpu(), alone[1]) and torch.equal(same[1].cpu(), alone[1])


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_two_stream_estimator_is_bit_identical(lib, precision):
    """Round 3: the estimator's batch rows as two launch chains on two HIP streams (option est_streams = 2, the default: fork / join through events,
    graph edges inside a captured solve) against one chain over all rows (est_streams = 1).  Same kernels on the same rows -> the same bits: through
    the estimator boundary with a DIFFERENT time value per batch row (the second half must pick its own time-embedding row), through inference()
    eager / captured / replayed, through an equal-shape batch and a padded batch (key counts per batch row follow the half)."""
    import ctypes as C
    import dataclasses
    cfg = dataclasses.replace(W.tiny()[1], n_timesteps=2)
    sd = W.make_flow(cfg)
    flows = []
    for streams in (1, 2):
        f = CausalMaskedDiffWithXvec(sd, cfg, lib=lib, precision=precision)
        lib.cv_flow_set_option(f._h, b"est_streams", C.c_int32(streams))
        flows.append(f)
    g = torch.Generator().manual_seed(91)
    T = 37
    x = torch.randn(2, 80, T, generator=g); mu = torch.randn(2, 80, T, generator=g); cond = torch.randn(2, 80, T, generator=g)
    spk = torch.randn(2, 80, generator=g); t = torch.tensor([0.3, 0.8]); mask = torch.ones(2, 1, T)
    for streaming in (False, True):
        a, b = (f.decoder.estimator(x, mask, mu, t, spk, cond, streaming=streaming).cpu() for f in flows)
        assert torch.equal(a, b)
        if precision == "fp32":
            torch.testing.assert_close(b, OF.estimator(sd, cfg, x, mask, mu, t, spk, cond, streaming), rtol=1e-3, atol=1e-3)
    n = lambda k: torch.tensor([k], dtype=torch.int32)
    shapes = [(14, 5, 10), (9, 7, 12), (21, 4, 6), (14, 5, 10)]
    items = [dict(token=torch.randint(0, cfg.vocab, (1, a), generator=g, dtype=torch.int32), prompt_token=torch.randint(0, cfg.vocab, (1, b), generator=g, dtype=torch.int32),
                  prompt_feat=torch.randn(1, c, cfg.mel, generator=g) * 2 - 5, embedding=torch.randn(1, cfg.spk_dim, generator=g)) for a, b, c in shapes]
    it = items[0]
    kw = dict(token=it["token"], token_len=n(14), prompt_token=it["prompt_token"], prompt_token_len=n(5), prompt_feat=it["prompt_feat"], prompt_feat_len=n(10),
              embedding=it["embedding"], streaming=False, finalize=True)
    single = [[f.inference(**kw)[0].cpu().clone() for _ in range(3)] for f in flows]          # eager, capture + launch, replay
    for m in single[0] + single[1]:
        assert torch.equal(m, single[0][0])
    for rep in range(3):
        ra, rb = (f.inference_batch(items[:3]) for f in flows)                  # different lengths: the padded pass
        for p, q in zip(ra, rb):
            assert torch.equal(p.cpu(), q.cpu())
    ea, eb = (f.inference_batch([items[0], items[3]]) for f in flows)
    for p, q in zip(ea, eb):
        assert torch.equal(p.cpu(), q.cpu())


def test_attention_workgroup_size_keeps_the_bits(lib):
    """bf16 mode: attn_flow32_kernel with 2 or 4 waves (64 / 128 queries) per workgroup (option attn32_waves; 0 = 4, the measured choice) computes every query with
    the same operations in the same order, and the residual GEMMs on 32 x 32 or 32 x 64 tiles (option res_tile) every element in the same k order: bit-identical
    estimator outputs, both mask modes, a time axis that ends inside a 32-query group."""
    import ctypes as C
    import dataclasses
    cfg = dataclasses.replace(W.tiny()[1], est_ch=128, est_heads=2, est_mid=1, chunk=13)
    sd = W.make_flow(cfg)
    flow = CausalMaskedDiffWithXvec(sd, cfg, lib=lib, precision="bf16")
    g = torch.Generator().manual_seed(5)
    for T in (70, 150):
        x = torch.randn(2, 80, T, generator=g); mu = torch.randn(2, 80, T, generator=g); cond = torch.randn(2, 80, T, generator=g)
        spk = torch.randn(2, 80, generator=g); t = torch.tensor([0.3, 0.3]); mask = torch.ones(2, 1, T)
        for streaming in (False, True):
            outs = {}
            for waves in (0, 2, 4):
                lib.cv_flow_set_option(flow._h, b"attn32_waves", C.c_int32(waves))
                outs[waves] = flow.decoder.estimator(x, mask, mu, t, spk, cond, streaming=streaming).cpu()
            assert torch.isfinite(outs[2]).all() and torch.equal(outs[2], outs[4]) and torch.equal(outs[0], outs[2])
            lib.cv_flow_set_option(flow._h, b"res_tile", C.c_int32(0))      # the residual GEMMs on 32 x 64 tiles (the default is 32 x 32): the same k order per element
            assert torch.equal(flow.decoder.estimator(x, mask, mu, t, spk, cond, streaming=streaming).cpu(), outs[0])
            lib.cv_flow_set_option(flow._h, b"res_tile", C.c_int32(1))
