"""BASELINE.json configs[0]: CosyVoice-300M-SFT `inference_sft`, one short utterance, PyTorch CPU eager - plumbing only,
no GPU.  Real CosyVoice-300M dimensions (configs.cv1()), seeded random weights (synthetic.make_cv1_*), text of 20 ids, a speaker embedding and no
prompt (what frontend_sft produces, cli/frontend.py:186-190); decode length capped at 5 tokens per text token.  Prints one JSON line with the
wall time per stage.  Usage: python tools/run_config1.py [--threads N]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=min(16, os.cpu_count() or 1))
    ap.add_argument("--ratio", type=float, default=5.0)
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    from cosyvoice_amd import cosyvoice1 as C1, synthetic as W
    cfg, hcfg = W.cv1()
    t0 = time.perf_counter()
    llm = C1.TransformerLM(W.make_cv1_llm(cfg), text_heads=cfg.text_heads, llm_heads=cfg.llm_heads)
    flow = C1.MaskedDiffWithXvec(W.make_cv1_flow(cfg), enc_heads=cfg.flow_heads, est_heads=cfg.est_heads, input_frame_rate=cfg.input_frame_rate)
    hift = C1.HiFTGenerator(W.make_hift(hcfg), sampling_rate=hcfg.sr, upsample_rates=hcfg.ups, upsample_kernel_sizes=hcfg.up_k, source_resblock_kernel_sizes=hcfg.src_k)
    model = C1.CosyVoiceModel(llm, flow, hift)
    t_build = time.perf_counter() - t0
    inf = llm.inference
    stamps = {}

    def timed_inference(**kw):
        t1 = time.perf_counter()
        n = 0
        for tok in inf(**dict(kw, max_token_text_ratio=a.ratio, min_token_text_ratio=a.ratio)):
            n += 1
            yield tok
        stamps["llm_s"], stamps["tokens"] = time.perf_counter() - t1, n
    llm.inference = timed_inference
    g = torch.Generator().manual_seed(1986)
    text = torch.randint(0, cfg.text_vocab, (1, 20), generator=g, dtype=torch.int32)
    emb = torch.randn(1, cfg.spk_dim, generator=g)
    torch.manual_seed(1986)
    t1 = time.perf_counter()
    out = next(iter(model.tts(text=text, flow_embedding=emb, llm_embedding=emb, stream=False)))["tts_speech"]
    wall = time.perf_counter() - t1
    audio_s = out.shape[1] / 22050.0
    assert torch.isfinite(out).all() and out.shape[1] == int(stamps["tokens"] / 50 * 22050 / 256) * 256
    print(json.dumps({"config": "BASELINE.json configs[0]: CosyVoice-300M inference_sft, 1 utterance, torch fp32 CPU eager (plumbing)", "threads": a.threads,
                      "host_cores": os.cpu_count(), "tokens": stamps["tokens"], "audio_s": round(audio_s, 3), "wall_s": round(wall, 2),
                      "llm_s": round(stamps["llm_s"], 2), "flow_hift_s": round(wall - stamps["llm_s"], 2), "rtf": round(wall / audio_s, 2),
                      "build_s": round(t_build, 1), "data": "synthetic (seeded random weights of the real architecture)"}))


if __name__ == "__main__":
    main()
