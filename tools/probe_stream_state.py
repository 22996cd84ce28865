"""Dev probe (round 3): 8 streaming clients before / after a 16-sequence batch run in the same process - which state does the batch run leave behind?
    gpurun -- python tools/probe_stream_state.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B

model, u, cfgs = B.build_model("bf16")
model.flow_batch = 4
model.set_lanes(int(os.environ.get("LANES", "4")))
B.one_utterance(model, u)
def show(tag):
    r = B.streaming_clients(model, u, 8, 104)
    print(tag, {k: r[k] for k in ("first_chunk_ms_p50", "first_chunk_ms_p90", "first_chunk_split_ms_p50", "audio_s_per_s", "shared_flow_passes")}, flush=True)
show("fresh          ")
which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("all", "lm"):
    lm_reqs = [dict(text=u["text"], prompt_text=u["prompt_text"], prompt_speech_token=u["llm_prompt_speech_token"]) for _ in range(16)]
    with model.llm_context:
        model.llm.inference_batch(lm_reqs, max_token_text_ratio=250 / 30, min_token_text_ratio=250 / 30)
    show("after LM batch16")
    show("again          ")
if which in ("all", "full"):
    r = B.batched_decode(model, u, 16, 1)
    print("batched_decode 16:", r["audio_s_per_s"], r["lm_us_per_step"], flush=True)
    show("after tts_batch16")
    show("again          ")
