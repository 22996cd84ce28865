"""bench.py's `cosyvoice300m` extra (SURVEY.md section 8 row f4 under the driver's clock) dry-run under the emulator at configs.tiny_cv1_k() and 25 tokens:
the function's host logic (forced length, stage split, token check against the torch-eager plumbing) runs before it ever sees the MI355X."""
import importlib.util
import os
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cosyvoice300m_extra_dry_run(emu_lib, monkeypatch):
    spec = importlib.util.spec_from_file_location("bench_mod_cv1", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    from cosyvoice_amd import cosyvoice1_hip as CK, synthetic as W
    monkeypatch.setenv("CV_BENCH_CV1_TOKENS", "16")
    monkeypatch.setattr(W, "cv1", W.tiny_cv1_k)
    monkeypatch.setattr(CK, "get_lib", lambda: emu_lib)
    import cosyvoice_amd.hift as H
    monkeypatch.setattr(H, "get_lib", lambda: emu_lib)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    res = bench.cv1_workload(types.SimpleNamespace(steps=2))
    assert res["token_check"] == {"checked": 16, "equal_torch_eager_cpu": True, "first_difference": None, "equal_real_reference_class": None}   # (the real-class fixture holds the 500-id request)
    assert res["audio_s_per_s"] > 0 and set(res["stages"]) == {"llm_ms", "flow_ms", "hift_ms", "llm_us_per_token"}
    f16 = res["fp16_mode"]                                          # round 6: the model's fp16 mode next to it (W16A32 LM + bf16-mode estimator)
    assert f16["token_check"]["equal"] and f16["token_check"]["checked"] == 16 and f16["audio_s_per_s"] > 0
    assert 0 < f16["flow_mel_vs_fp32_mode"]["rel_l2"] < 5e-2
