#!/usr/bin/env python
"""Headline benchmark: audio-seconds synthesised per wall-second (1/RTF), CosyVoice2-0.5B zero-shot, batch 1, 10 CFM Euler steps
(BASELINE.json configs[1]) on the synthetic U10 utterance of SURVEY.md §8d, plus first-chunk p50 latency in streaming mode.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One process per GPU.  Utterances are independent, so ranks are replicas of the whole pipeline (no data-path collective, xGMI
idle — SURVEY.md §8e); RCCL is used only for the barrier + max-over-ranks of the timed region.  A "step" is one complete
utterance: lm_input build -> LLM prefill(131) -> 250 greedy decode steps -> flow (encoder + 10 CFG Euler steps, T=674) -> HiFT
(500 frames) -> 240 000 samples copied to the host.  Inputs are resident on the GPU when the timed region starts.
Weights are seeded random tensors of the real architecture (no checkpoints on the box): `data: synthetic`.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

_T0 = time.time()


def log(msg):
    print("[bench %7.1fs] %s" % (time.time() - _T0, msg), file=sys.stderr, flush=True)


N_GEN, N_TEXT, N_PROMPT_TEXT, N_PROMPT_TOK = 250, 30, 12, 87
AUDIO_S = N_GEN / 25.0
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec


def build_model(flow_precision="bf16"):
    from cosyvoice_amd.configs import cv2
    from cosyvoice_amd.model import CosyVoice2Model
    from cosyvoice_amd import synthetic as W
    cfgs = cv2()
    model = CosyVoice2Model.from_state_dicts(W.make_llm(cfgs[0]), W.make_flow(cfgs[1]), W.make_hift(cfgs[2]), cfgs,
                                             max_len=1024, sampling="greedy", decode_chunk=64, fp16=(flow_precision == "bf16"))
    u = W.synthetic_utterance(cfgs[0], cfgs[1], n_prompt_tok=N_PROMPT_TOK, n_prompt_text=N_PROMPT_TEXT, n_text=N_TEXT)
    dev = model.device
    u = {k: v.to(dev) for k, v in u.items()}
    return model, u, cfgs


def one_utterance(model, u):
    """The hot path for one utterance; returns the host waveform [1, 240000]."""
    t = lambda n: torch.tensor([n], dtype=torch.int32)
    ratio = N_GEN / N_TEXT
    with model.llm_context:
        tokens = list(model.llm.inference(text=u["text"], text_len=t(N_TEXT), prompt_text=u["prompt_text"], prompt_text_len=t(N_PROMPT_TEXT),
                                          prompt_speech_token=u["llm_prompt_speech_token"], prompt_speech_token_len=t(N_PROMPT_TOK),
                                          embedding=u["llm_embedding"], max_token_text_ratio=ratio, min_token_text_ratio=ratio))
    assert len(tokens) == N_GEN, len(tokens)
    uid = "bench"
    model.hift_cache_dict[uid] = None
    wav = model.token2wav(token=torch.tensor(tokens).unsqueeze(0), prompt_token=u["flow_prompt_speech_token"], prompt_feat=u["prompt_speech_feat"],
                          embedding=u["flow_embedding"], token_offset=0, uuid=uid, finalize=True)
    out = wav.cpu()
    model.hift_cache_dict.pop(uid, None)
    assert out.shape[1] == N_GEN * 2 * 480
    return out


def batched_decode(model, u, nb, reps):
    """Serving-style extra (BASELINE.json configs[2]/[3]): NB copies of the U10 request through tts_batch.  The LM step streams every
    weight matrix once for all NB sequences; flow and HiFT still run per utterance.  Also reports the LM part alone."""
    keys = ("text", "flow_embedding", "llm_embedding", "prompt_text", "llm_prompt_speech_token", "flow_prompt_speech_token", "prompt_speech_feat")
    reqs = [{k: u[k] for k in keys} for _ in range(nb)]
    ratio = N_GEN / N_TEXT
    inf_b, inf_q = model.llm.inference_batch, model.llm.inference_queue
    model.llm.inference_batch = lambda r: inf_b(r, max_token_text_ratio=ratio, min_token_text_ratio=ratio)       # force 250 tokens each
    model.llm.inference_queue = lambda r, slots=8: inf_q(r, slots=slots, max_token_text_ratio=ratio, min_token_text_ratio=ratio)
    try:
        model.tts_batch(reqs)                                     # warm-up (graph capture for this batch size)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            outs = model.tts_batch(reqs)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        assert all(o["tts_speech"].shape[1] == N_GEN * 2 * 480 for o in outs)
        # pipeline: 2 x NB requests through NB slots, LM (continuous batching, LLM stream / thread) overlapped with flow + HiFT of finished ones
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        n_done = sum(1 for _ in model.tts_queue(reqs + reqs, slots=nb))
        torch.cuda.synchronize()
        pipe_s = time.perf_counter() - t2
        lm_reqs = [dict(text=u["text"], prompt_text=u["prompt_text"], prompt_speech_token=u["llm_prompt_speech_token"]) for _ in range(nb)]
        with model.llm_context:
            t1 = time.perf_counter()
            toks = model.llm.inference_batch(lm_reqs)
            torch.cuda.synchronize()
            lm_s = time.perf_counter() - t1
    finally:
        model.llm.inference_batch, model.llm.inference_queue = inf_b, inf_q
    return {"batch": nb, "audio_s_per_s": round(nb * reps * AUDIO_S / el, 3), "ms_per_batch": round(1e3 * el / reps, 2),
            "pipeline_audio_s_per_s": round(n_done * AUDIO_S / pipe_s, 3),
            "lm_tokens_per_s": round(sum(len(t) for t in toks) / lm_s, 1), "lm_us_per_step": round(1e6 * lm_s / max(len(toks[0]), 1), 1)}


def concurrent_streams(model0, u, n_streams, steps):
    """Serving-style extra: the batch-1 decode is a latency chain that leaves most of the 256 CUs idle, so S independent model
    instances (own weights, KV cache, graphs and HIP streams; one host thread each - ctypes drops the GIL inside the library)
    overlap on one GPU.  Not the headline `value` (that stays batch=1, one utterance in flight)."""
    import threading
    models = [model0] + [build_model("bf16" if model0.fp16 else "fp32")[0] for _ in range(n_streams - 1)]
    for m in models[1:]:
        one_utterance(m, u)
    torch.cuda.synchronize()
    errs = []

    def work(m):
        try:
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                for _ in range(steps):
                    one_utterance(m, u)
            s.synchronize()
        except Exception as e:                      # pragma: no cover - reported below
            errs.append(repr(e))

    th = [threading.Thread(target=work, args=(m,)) for m in models]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if errs:
        raise RuntimeError("; ".join(errs))
    return {"streams": n_streams, "utterances": n_streams * steps, "audio_s_per_s": round(n_streams * steps * AUDIO_S / el, 3),
            "ms_per_utterance_per_stream": round(1e3 * el / steps, 2)}


def first_chunk_latency(model, u, reps):
    """Streaming tts(): time from the call to the first yielded chunk (client_grpc.py:73-87 definition)."""
    lat = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        gen = model.tts(text=u["text"], flow_embedding=u["flow_embedding"], llm_embedding=u["llm_embedding"], prompt_text=u["prompt_text"],
                        llm_prompt_speech_token=u["llm_prompt_speech_token"], flow_prompt_speech_token=u["flow_prompt_speech_token"],
                        prompt_speech_feat=u["prompt_speech_feat"], stream=True)
        first = next(gen)
        lat.append((time.perf_counter() - t0) * 1e3)
        assert first["tts_speech"].shape[1] > 0
        for _ in gen:
            pass
    lat.sort()
    return lat[len(lat) // 2]


def roofline_llm(model, u, cfgs):
    """Per-kernel HIP-event timing of eager decode steps (cv_llm_profile_step) -> roofline of the dominant kernel class."""
    from cosyvoice_amd.llm import SamplingC
    lc = cfgs[0]
    llm = model.llm
    with model.llm_context:
        llm.prefill(llm.build_lm_input(u["text"], u["prompt_text"], u["llm_prompt_speech_token"]))
        sp = SamplingC(0, llm.eos_token, 3, N_GEN, N_GEN, 0.8, 25, 10, 0.1, 0, 0)
        counts, ms = (C.c_int32 * 8)(), (C.c_float * 8)()
        tot_c, tot_ms = [0] * 8, [0.0] * 8
        from cosyvoice_amd._lib import stream_ptr
        for i in range(40):
            llm.lib.cv_llm_profile_step(llm._h, C.byref(sp), counts, ms, stream_ptr(llm.lib))
            if i >= 8:
                for k in range(8):
                    tot_c[k] += counts[k]; tot_ms[k] += ms[k]
    H, I, V, A, Q = lc.hidden, lc.inter, lc.speech_token_size + 3, lc.heads * 64, (lc.heads + 2 * lc.kv_heads) * 64
    # algorithmic bytes per launch: bf16 weight rows streamed once (2 B/param) — activations/bias are negligible
    wbytes = {0: 2 * Q * H, 2: 2 * H * A, 3: 2 * 2 * I * H, 4: 2 * H * I, 5: 2 * V * H}
    names = {0: "gemv_kernel<7,1,1> qkv", 2: "gemv_kernel<2,1,4,8> o_proj (+ attention-partial merge)", 3: "gemv_kernel<7,2,1> gate_up", 4: "gemv_kernel<10,1,4> down",
             5: "gemv_kernel<7,2,1> head", 1: "attn_decode_kernel", 6: "sample_kernel"}
    per = {names[k]: dict(launches=tot_c[k], event_pair_avg_us=round(1e3 * tot_ms[k] / max(tot_c[k], 1), 2)) for k in names if tot_c[k]}
    # Per-launch duration inside the graph: each kernel class replayed as a dependent chain of its real launches (one per layer, own
    # weights) between ONE event pair on the decode stream (cv_llm_profile_chain).  An event pair around a single 3-7 us launch adds
    # ~3 us of its own (kept above as event_pair_avg_us); rocprofv3 averages sit between the two (profiles/).
    chain = {}
    with model.llm_context:
        for k in (0, 1, 2, 3, 4, 5):
            ms1, n1 = C.c_float(), C.c_int32()
            llm.lib.cv_llm_profile_chain(llm._h, k, 20, C.byref(ms1), C.byref(n1), stream_ptr(llm.lib))
            chain[k] = 1e3 * ms1.value / max(n1.value, 1)
            per[names[k]]["chain_avg_us"] = round(chain[k], 2)
            if k in wbytes:
                per[names[k]]["weight_bytes"] = wbytes[k]
                per[names[k]]["GBps"] = round(wbytes[k] / (chain[k] * 1e-6) / 1e9, 1)
    # dominant kernel of the whole path: gemv_kernel<7,2,1> on the gate_up matrices (24 launches per token, ~30 % of the decode step,
    # the largest single share of an utterance)
    bytes_per_launch = wbytes[3]
    avg_s = chain[3] * 1e-6
    achieved = bytes_per_launch / avg_s / 1e9 if avg_s > 0 else 0.0
    # HBM traffic per launch from the PMC pass committed under profiles/ (rocprofv3 --pmc FETCH_SIZE in its own run, x1024 B, x2 for
    # the gfx950 wide-read under-count — MI355X_MICROARCH.md §HBM); PMC cannot be collected from inside this process, hence the file.
    traffic, traffic_src = None, None
    pmc = os.path.join(ROOT, "profiles", "r1_pmc_gemv_fetch.json")
    if os.path.exists(pmc):
        d = json.load(open(pmc))
        sel = [v for k, v in d.items() if "<7, 2, 1>" in k]
        if sel:
            traffic = int(sum(v["n"] * v["hbm_read_bytes_corrected"] for v in sel) / sum(v["n"] for v in sel))
            traffic_src = "profiles/r1_pmc_gemv_fetch.json (FETCH_SIZE*1024*2, mean over the gemv_kernel<7,2,1> launches of tools/profile_small.py llm)"
    step_us = sum(chain[k] * lc.layers for k in (0, 1, 2, 3, 4)) + chain[5]
    return dict(bound="hbm", kernel="gemv_kernel<7,2,1> (LLM decode, gate_up weight stream: 2 x 4864 x 896 bf16 per launch)", achieved=round(achieved, 1), peak=HBM_PEAK_GBS,
                unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 4), traffic=traffic, traffic_source=traffic_src, bytes_per_launch=int(bytes_per_launch),
                avg_launch_us=round(avg_s * 1e6, 2), timing="hipGraph chain of the kernel's 24 per-layer launches x 20 replays between one HIP-event pair on the decode stream",
                decode_step_us_from_chains=round(step_us, 1), per_kernel=per)


def cpu_baseline(cfgs):
    """The oracle (CPU restatement of the reference) timed on the host cores, on a bounded sample of U10, extrapolated per stage."""
    from oracle import flow as OF, hift as OH, llm as OL      # the ONLY place bench.py touches oracle/: the reported CPU baseline
    from cosyvoice_amd import synthetic as W
    lc, fc, hc = cfgs
    cores = min(os.cpu_count() or 1, 64)       # more OpenMP threads than that only adds barrier cost on these small ops
    torch.set_num_threads(cores)
    u = W.synthetic_utterance(lc, fc, n_prompt_tok=N_PROMPT_TOK, n_prompt_text=N_PROMPT_TEXT, n_text=N_TEXT)
    with torch.inference_mode():
        sd = W.make_llm(lc)
        m = OL.Qwen2Oracle(sd, lc)
        x = OL.build_lm_input(sd, lc, u["text"], u["prompt_text"], u["llm_prompt_speech_token"])
        t0 = time.perf_counter(); m.forward(x); t_prefill = time.perf_counter() - t0
        tok = sd["speech_embedding.weight"][5].reshape(1, -1)
        m.forward(tok)
        t0 = time.perf_counter()
        for _ in range(6):
            y = m.forward(tok)
            torch.nn.functional.linear(y[-1], sd["llm_decoder.weight"], sd["llm_decoder.bias"]).log_softmax(-1)
        t_tok = (time.perf_counter() - t0) / 6
        del sd, m
        fsd = W.make_flow(fc)
        T = 2 * (N_PROMPT_TOK + N_GEN)
        g = torch.Generator().manual_seed(0)
        emb = torch.randn(1, N_PROMPT_TOK + N_GEN, fc.dim, generator=g)
        t0 = time.perf_counter(); OF.encoder(fsd, fc, emb, None, False); t_enc = time.perf_counter() - t0
        xx = torch.randn(2, 80, T, generator=g)
        t0 = time.perf_counter()
        OF.estimator(fsd, fc, xx, torch.ones(2, 1, T), xx, torch.tensor([0.3, 0.3]), torch.randn(2, 80, generator=g), xx, False)
        t_est = time.perf_counter() - t0
        del fsd
        hsd = W.make_hift(hc)
        mel = torch.randn(1, 80, 100, generator=g) * 2 - 5
        OH.inference(hsd, hc, mel[:, :, :20])
        t0 = time.perf_counter(); OH.inference(hsd, hc, mel); t_hift = (time.perf_counter() - t0) * 5.0
    total = t_prefill + N_GEN * t_tok + t_enc + fc.n_timesteps * t_est + t_hift
    return dict(value=round(AUDIO_S / total, 4), unit="audio_s/s", cores=cores, kind="port",
                sample="oracle/ (torch fp32, %d threads): LLM prefill(131) + 6 decode steps, flow encoder(337 tok) + 1 of 10 estimator steps at T=674, "
                       "HiFT 100 of 500 frames; per-stage times extrapolated to the full U10 utterance" % cores,
                stage_seconds=dict(llm_prefill=round(t_prefill, 3), llm_per_token=round(t_tok, 4), flow_encoder=round(t_enc, 3),
                                   flow_estimator_step=round(t_est, 3), hift_500_frames=round(t_hift, 3)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--first-chunk-reps", type=int, default=3)
    ap.add_argument("--flow-precision", choices=("bf16", "fp32"), default="bf16",
                    help="operand precision of the flow's Linear/Conv1d products (BASELINE.json configs[1] is a bf16 configuration); "
                         "fp32 = exact-fp32 MFMA everywhere")
    ap.add_argument("--batch", type=int, default=0, help="extra (not `value`): NB requests through CosyVoice2Model.tts_batch - lock-step batched "
                    "LM decode (weights streamed once per step for all of them), flow + HiFT per utterance; reported as `batched_decode`")
    ap.add_argument("--streams", type=int, default=1, help="extra (not `value`): S independent model instances on this GPU, one host thread + HIP "
                    "stream each, all synthesising concurrently; reported as `concurrent_streams`")
    args = ap.parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node N for --gpus N"
    torch.cuda.set_device(local_rank)
    # the product path's host work is a few tiny CPU tensor ops per utterance: keep torch's intra-op pool small so that N ranks on
    # one node do not oversubscribe the host (cpu_baseline() sets its own thread count for the oracle run)
    torch.set_num_threads(max(1, min(8, (os.cpu_count() or 8) // max(world, 1))))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", init_method="env://", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    model, u, cfgs = build_model(args.flow_precision)
    log("model built")
    for _ in range(args.warmup):
        one_utterance(model, u)
    log("warmup done")

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_utterance(model, u)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        value = world * args.steps * AUDIO_S / elapsed
        log("timed region: %.3f s for %d steps -> %.2f audio_s/s" % (elapsed, args.steps, value))
        out = {
            "metric": "audio-sec/s (RTF^-1), CosyVoice2-0.5B zero-shot", "value": round(value, 3), "unit": "audio_s/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ("bf16 weights everywhere; LLM: fp32 activations/accumulate (W16A32, token ids bit-exact); flow Linear/Conv1d: bf16 x bf16 "
                      "MFMA with fp32 accumulate, attention/norms/Euler fp32; HiFT fp32") if args.flow_precision == "bf16" else
                     "f32 (bf16 weights, fp32 activations and accumulate; HiFT fp32)", "data": "synthetic",
            "config": {"workload": "CosyVoice2-0.5B zero-shot, batch=1, 10 CFM Euler steps, synthetic U10: prompt 87 speech tokens / 174 mel frames, "
                                   "12+30 text tokens, 250 generated tokens = 10.0 s @ 24 kHz (BASELINE.json configs[1])",
                       "utterances_per_gpu_per_step": 1, "sampler": "greedy, length forced to 250", "flow_precision": args.flow_precision, "parallelism": "replicas x%d, no collective" % world},
            "per_gpu_audio_s_per_s": round(value / world, 3),
        }
        if world == 1 and args.batch > 0:
            out["batched_decode"] = batched_decode(model, u, args.batch, max(1, args.steps // 2))
            log("batched decode done")
        if world == 1 and args.streams > 1:
            out["concurrent_streams"] = concurrent_streams(model, u, args.streams, args.steps)
            log("concurrent streams done")
        if world == 1:
            out["first_chunk_ms_p50"] = round(first_chunk_latency(model, u, args.first_chunk_reps), 2)
            log("first chunk p50 %.1f ms" % out["first_chunk_ms_p50"])
        out["roofline"] = roofline_llm(model, u, cfgs)          # rank 0's GPU (every replica runs the same kernels)
        log("roofline done")
        if world == 1 and not args.no_cpu_baseline:             # the CPU baseline is reported at N = 1 only
            out["cpu_baseline"] = cpu_baseline(cfgs)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
