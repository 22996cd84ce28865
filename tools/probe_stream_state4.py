"""Dev probe (round 3): which part of a 16-sequence offline batch degrades 8-client streaming afterwards - tts_batch or tts_queue?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B

model, u, cfgs = B.build_model("bf16")
model.flow_batch = 4
model.set_lanes(4)
B.one_utterance(model, u)
keys = ("text", "flow_embedding", "llm_embedding", "prompt_text", "llm_prompt_speech_token", "flow_prompt_speech_token", "prompt_speech_feat")
reqs = [dict({k: u[k] for k in keys}, min_token_text_ratio=250 / 30, max_token_text_ratio=250 / 30) for _ in range(16)]
def show(tag):
    r = B.streaming_clients(model, u, 8, 104)
    print(tag, {k: r[k] for k in ("first_chunk_ms_p50", "first_chunk_ms_p90", "first_chunk_split_ms_p50", "audio_s_per_s")}, r["shared_flow_passes"]["requests_in_them"], flush=True)
which = sys.argv[1]
show("fresh            ")
if which == "batch":
    model.tts_batch(reqs); torch.cuda.synchronize()
    show("after tts_batch16")
else:
    n = sum(1 for _ in model.tts_queue(reqs, slots=16)); torch.cuda.synchronize()
    show("after tts_queue16")
show("again            ")
