"""Dev probe (round 3): one shared first-chunk token2wav_batch (4 requests) and one HiFT call of a first chunk (84 frames) on an idle GPU, before and after a
16-sequence tts_batch in the same process.   gpurun -- python tools/probe_stream_state3.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B

model, u, cfgs = B.build_model("bf16")
model.flow_batch = 4
model.set_lanes(4)
B.one_utterance(model, u)
tok = torch.randint(0, 6561, (1, 41), dtype=torch.int32)
def t2w(tag):
    ts = []
    for rep in range(6):
        jobs = []
        for i in range(4):
            key = "p%d_%d" % (rep, i); model.hift_cache_dict[key] = None
            jobs.append(dict(token=tok, prompt_token=u["flow_prompt_speech_token"], prompt_feat=u["prompt_speech_feat"], embedding=u["flow_embedding"], token_offset=0, uuid=key))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        model.token2wav_batch(jobs, stream=True, finalize=False)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        for j in jobs: model.hift_cache_dict.pop(j["uuid"], None)
    mel = torch.randn(1, 80, 84, device=model.device)
    hs = []
    for rep in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        model.hift.inference(speech_feat=mel, cache_source=torch.zeros(1, 1, 0))
        torch.cuda.synchronize(); hs.append((time.perf_counter() - t0) * 1e3)
    print("%s token2wav_batch(4 first chunks) ms: %s | hift.inference(84 frames) ms: %s" % (tag, ["%.1f" % x for x in ts], ["%.2f" % x for x in hs]), flush=True)
t2w("fresh ")
r = B.batched_decode(model, u, 16, 1)
print("batched_decode 16:", r["audio_s_per_s"], flush=True)
t2w("after ")
r = B.streaming_clients(model, u, 8, 104)
print("streaming after:", {k: r[k] for k in ("first_chunk_ms_p50", "first_chunk_ms_p90", "first_chunk_split_ms_p50", "audio_s_per_s", "shared_flow_passes")}, flush=True)
t2w("after2")
