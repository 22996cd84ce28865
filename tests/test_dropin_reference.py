"""The drop-in boundary exercised INSIDE THE REAL REFERENCE CLASS.

Build-container only: needs the reference tree (/root/reference; the GPU box does not have it, so every test here skips there) and runs
the product objects on the CPU emulator of the HIP execution model.  A real `cosyvoice.cli.model.CosyVoice2Model` is built around the real
tiny `CausalMaskedDiffWithXvec` / `HiFTGenerator` modules; then the attribute replacements a maintainer would make are applied one plug point
at a time - B3 `flow.decoder.estimator`, B4 `flow.encoder`, B5 `model.flow`, B6 `model.hift`, B2 `model.llm` - and the reference's OWN
`tts()` (its llm_job thread, chunk loop, hop doubling, mel / source / speech caches, `fade_in_out`) is run in streaming and one-shot mode.
Every swapped model must give the waveform of the unswapped one (5e-3, the tolerance of the other model-level tests: fp32 summation
order through flow + vocoder), chunk for chunk.
"""
import dataclasses
import os
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))

from cosyvoice_amd import synthetic as W  # noqa: E402

pytestmark = pytest.mark.skipif(not os.path.isdir(os.environ.get("COSYVOICE_REFERENCE", "/root/reference")),
                                reason="the reference tree is only present in the build container")

class _NoSleepTime:
    """Stands in for the `time` module INSIDE cosyvoice.cli.model only (its tts() polls with time.sleep(0.1)).  Assigning to `M.time.sleep` would patch the
    stdlib module for the whole process - and the pytest-xdist worker goes on to run other files (a sleep-based scheduler test failed that way)."""
    sleep = staticmethod(lambda s: None)

    def __getattr__(self, name):
        import time
        return getattr(time, name)


N_STEPS = 2          # Euler steps of the tiny fixtures (the reference hard-codes 10, flow/flow.py:278; patched like tests/golden/make_golden.py)


@pytest.fixture(scope="module")
def ref():
    """The real reference modules (through the import stubs of tests/golden/ref_import.py) + seeded inputs + the unswapped model's output."""
    import ref_import
    ref_import.install()
    import make_golden as MG                                  # build_ref_flow / build_ref_hift: real classes, strict=True state-dict load
    import cosyvoice.cli.model as M
    lc, _, hc = W.tiny()
    fc = dataclasses.replace(W.ref_small_flow(), chunk=5, n_timesteps=N_STEPS)
    u = W.synthetic_utterance(lc, fc, n_prompt_tok=8, n_prompt_text=4, n_text=2, seed=21)
    tokens = torch.randint(0, fc.vocab, (24,), generator=torch.Generator().manual_seed(31)).tolist()
    M.time = _NoSleepTime()                                   # the reference polls with sleep(0.1)
    ctx = dict(M=M, MG=MG, cfgs=(lc, fc, hc), u=u, tokens=tokens)
    ctx["base"] = {stream: _run(ctx, _real_model(ctx), stream) for stream in (False, True)}
    assert len(ctx["base"][False]) == 1 and len(ctx["base"][True]) >= 3          # one-shot: one waveform; streaming: hops 5 + pad, 10, rest
    assert sum(c.shape[1] for c in ctx["base"][True]) == ctx["base"][False][0].shape[1] == len(tokens) * 960
    return ctx


class _Scripted:
    def __init__(self, tokens):
        self.tokens = tokens

    def inference(self, **kw):
        yield from self.tokens


def _real_model(ctx, llm=None):
    """A fresh REAL CosyVoice2Model around fresh real modules (so that a swap on one never leaks into the next test)."""
    lc, fc, hc = ctx["cfgs"]
    flow, hift = ctx["MG"].build_ref_flow(fc), ctx["MG"].build_ref_hift(hc)
    m = ctx["M"].CosyVoice2Model(llm or _Scripted(ctx["tokens"]), flow, hift)
    m.token_hop_len, m.token_max_hop_len = 5, 20
    return m


def _run(ctx, m, stream):
    """The reference's own tts() with its 10 hard-coded Euler steps cut to the fixture's 2 and SineGen2's additive noise zeroed (the convention of
    every model-level test here: the noise is an explicit argument on the product side)."""
    u = ctx["u"]
    dec = type(m.flow.decoder) if hasattr(m.flow, "decoder") and isinstance(m.flow.decoder, torch.nn.Module) else None
    orig_fwd = dec.forward if dec is not None else None
    if dec is not None:
        def fwd(self, mu, mask, spks, cond, n_timesteps=10, **kw):
            return orig_fwd(self, mu=mu, mask=mask, spks=spks, cond=cond, n_timesteps=N_STEPS, **kw)
        dec.forward = fwd
    orig_randn_like = torch.randn_like
    torch.randn_like = lambda t, **kw: torch.zeros_like(t)
    try:
        with torch.inference_mode():
            return [o["tts_speech"] for o in m.tts(text=u["text"], flow_embedding=u["flow_embedding"], llm_embedding=u["llm_embedding"], prompt_text=u["prompt_text"],
                                                   llm_prompt_speech_token=u["llm_prompt_speech_token"], flow_prompt_speech_token=u["flow_prompt_speech_token"],
                                                   prompt_speech_feat=u["prompt_speech_feat"], stream=stream)]
    finally:
        torch.randn_like = orig_randn_like
        if dec is not None:
            dec.forward = orig_fwd


def _amd_flow(ctx, lib):
    from cosyvoice_amd.flow import CausalMaskedDiffWithXvec
    fc = ctx["cfgs"][1]
    return CausalMaskedDiffWithXvec(W.make_flow(fc), fc, lib=lib, n_timesteps=N_STEPS)


def _amd_hift(ctx, lib):
    from cosyvoice_amd.hift import HiFTGenerator
    hc = ctx["cfgs"][2]
    h = HiFTGenerator(W.make_hift(hc), hc, lib=lib)
    inf = h.inference
    h.inference = lambda speech_feat, cache_source=None: inf(speech_feat, cache_source, noise=torch.zeros(speech_feat.shape[2] * 480, hc.harmonics + 1))
    return h


def _same(ctx, got, stream):
    want = ctx["base"][stream]
    assert [g.shape for g in got] == [w.shape for w in want]
    for g, w in zip(got, want):
        torch.testing.assert_close(g.cpu(), w, rtol=0, atol=5e-3)
    assert float(torch.cat(want, 1).abs().max()) > 0.05            # a comparison of silences would prove nothing


def test_b3_estimator_swap_inside_the_real_model(emu_lib, ref):
    """`model.flow.decoder.estimator = EstimatorModule(amd_flow)`; the reference's solve_euler / forward_estimator
    (flow_matching.py:71-153) drive the product estimator, in the chunk-masked streaming calls as well as the one-shot call."""
    from cosyvoice_amd.flow import EstimatorModule
    for stream in (False, True):
        m = _real_model(ref)
        m.flow.decoder.estimator = EstimatorModule(_amd_flow(ref, emu_lib))
        assert isinstance(m.flow.decoder.estimator, torch.nn.Module)
        _same(ref, _run(ref, m, stream), stream)


def test_b3_estimator_with_a_padded_mask_inside_the_real_cfm(emu_lib, ref):
    """B3 with the reference's padded `mask`: the REAL CausalConditionalCFM (forward -> solve_euler -> forward_estimator,
    flow_matching.py:71-153,203-227) solves a sequence whose mask ends 9 frames before the tensors do - once around the real estimator, once around
    EstimatorModule.  The valid frames agree; the reference multiplies by the mask inside every block, the product ends the key loops at the mask's length."""
    from cosyvoice_amd.flow import EstimatorModule
    fc = ref["cfgs"][1]
    g = torch.Generator().manual_seed(77)
    T, n = 64, 55
    mu, cond, spks = torch.randn(1, 80, T, generator=g), torch.randn(1, 80, T, generator=g), torch.randn(1, 80, generator=g)
    mask = torch.zeros(1, 1, T); mask[:, :, :n] = 1
    outs = []
    for swap in (False, True):
        flow = ref["MG"].build_ref_flow(fc)
        if swap:
            flow.decoder.estimator = EstimatorModule(_amd_flow(ref, emu_lib))
        with torch.inference_mode():
            y, _ = flow.decoder(mu=mu * mask, mask=mask, spks=spks, cond=cond * mask, n_timesteps=N_STEPS)
        outs.append(y.cpu())
    torch.testing.assert_close(outs[1][:, :, :n], outs[0][:, :, :n], rtol=2e-4, atol=2e-4)
    assert float(outs[0][:, :, :n].abs().max()) > 0.1


def test_b3_engine_form_inside_the_real_cfm(emu_lib, ref, monkeypatch):
    """B3, the NON-Module branch of forward_estimator (flow/flow_matching.py:129-153, the shape of the reference's TensorRT path): `del decoder.estimator;
    decoder.estimator = EstimatorEngine(amd_flow)`.  The REAL CausalConditionalCFM.forward -> solve_euler -> forward_estimator then acquires the context, sets six
    input shapes and seven raw tensor addresses (the output aliased on x), executes on the current stream and releases - and gets what the nn.Module form
    (EstimatorModule) gives, bit for bit, and the real estimator's result within the fp32 bound.  (No GPU here: `torch.cuda.current_stream()` of that branch is
    stood in for; the addresses are host addresses the emulator build reads.)"""
    from cosyvoice_amd.flow import EstimatorEngine, EstimatorModule
    fc = ref["cfgs"][1]
    g = torch.Generator().manual_seed(78)
    T = 48
    mu, cond, spks = torch.randn(1, 80, T, generator=g), torch.randn(1, 80, T, generator=g), torch.randn(1, 80, generator=g)
    mask = torch.ones(1, 1, T)

    class _Stream:
        cuda_stream = 0

        def synchronize(self):
            pass
    monkeypatch.setattr(torch.cuda, "current_stream", lambda *a, **k: _Stream())
    outs = {}
    for form in ("real", "module", "engine"):
        flow = ref["MG"].build_ref_flow(fc)
        amd = _amd_flow(ref, emu_lib)
        if form == "module":
            flow.decoder.estimator = EstimatorModule(amd)
        elif form == "engine":
            del flow.decoder.estimator                          # (a registered submodule name: cli/model.py:load_trt deletes it first, too)
            flow.decoder.estimator = EstimatorEngine(amd)
            assert not isinstance(flow.decoder.estimator, torch.nn.Module)
        with torch.inference_mode():
            y, _ = flow.decoder(mu=mu, mask=mask, spks=spks, cond=cond, n_timesteps=N_STEPS)
        outs[form] = y.cpu()
        if form == "engine":
            eng = flow.decoder.estimator
            assert eng._pool.qsize() == 1                       # released after every call
            ctx = eng._pool.queue[0][0]
            assert ctx.shapes["x"] == (2, 80, T) and ctx.addrs["estimator_out"] == ctx.addrs["x"] and set(ctx.addrs) == set(EstimatorEngine.TENSOR_NAMES)
            with pytest.raises(ValueError):                     # a context that was not given the estimator's tensors says so instead of reading address 0
                EstimatorEngine._Context(eng).execute_async_v3(0)
    assert torch.equal(outs["engine"], outs["module"])
    torch.testing.assert_close(outs["engine"], outs["real"], rtol=2e-4, atol=2e-4)
    assert float(outs["real"].abs().max()) > 0.1


def test_b4_encoder_swap_inside_the_real_model(emu_lib, ref):
    """B4: `model.flow.encoder = amd_flow.encoder`, the contract of the reference's own TorchScript encoder swap (cli/model.py:277-279)."""
    for stream in (False, True):
        m = _real_model(ref)
        m.flow.encoder = _amd_flow(ref, emu_lib).encoder
        _same(ref, _run(ref, m, stream), stream)


def test_b5_b6_flow_and_hift_swaps_inside_the_real_model(emu_lib, ref):
    """B5 `model.flow = amd_flow` (flow.inference incl. the on-device Euler loop), B6 `model.hift = amd_hift`, each alone and both together, under
    the reference's token2wav (its caches slice the product's tensors, its fade_in_out mixes them)."""
    for swap in ("flow", "hift", "both"):
        for stream in (False, True):
            m = _real_model(ref)
            if swap in ("flow", "both"):
                m.flow = _amd_flow(ref, emu_lib)
            if swap in ("hift", "both"):
                m.hift = _amd_hift(ref, emu_lib)
            _same(ref, _run(ref, m, stream), stream)
            assert not m.tts_speech_token_dict and not m.hift_cache_dict


def test_b2_llm_swap_inside_the_real_model(emu_lib, ref):
    """B2: `model.llm = cosyvoice_amd.llm.Qwen2LM(...)`; the reference's llm_job iterates `self.llm.inference(text=..., text_len=..., prompt_text=...,
    ..., embedding=..., uuid=...)` (cli/model.py:101-129) on its own thread and appends the yielded ints.  The tokens must be the greedy tokens of
    the oracle (itself pinned to the real Qwen2LM by tests/golden/llm_tiny.npz), the audio what the real model makes of those tokens."""
    from cosyvoice_amd.llm import Qwen2LM
    from oracle import llm as OL
    lc = ref["cfgs"][0]
    sd = W.make_llm(lc)
    u = ref["u"]
    want_tokens = OL.inference(sd, lc, u["text"], u["prompt_text"], u["llm_prompt_speech_token"])
    assert len(want_tokens) >= 8
    seen = []
    amd = Qwen2LM(sd, lc, lib=emu_lib, max_len=160, sampling="greedy")
    inner = amd.inference

    def spy(**kw):
        for t in inner(**kw):
            seen.append(t)
            yield t
    amd.inference = spy
    got = _run(ref, _real_model(ref, llm=amd), False)
    assert seen == want_tokens and all(isinstance(t, int) for t in seen)
    want = _run(ref, _real_model(ref, llm=_Scripted(want_tokens)), False)
    assert [g.shape for g in got] == [w.shape for w in want]
    torch.testing.assert_close(got[0], want[0], rtol=0, atol=0)     # same real flow + vocoder on the same tokens
