"""Dev probe (round 3): time of one shared first-chunk flow pass (4 requests, 87 + 41 tokens, streaming) on every token2wav lane, idle GPU, before and after a
16-sequence tts_batch in the same process.   gpurun -- python tools/probe_stream_state2.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B

model, u, cfgs = B.build_model("bf16")
model.flow_batch = 4
model.set_lanes(4)
B.one_utterance(model, u)
tok = torch.randint(0, 6561, (1, 41), dtype=torch.int32)
item = dict(token=tok, prompt_token=u["flow_prompt_speech_token"], prompt_feat=u["prompt_speech_feat"], embedding=u["flow_embedding"])
lanes = list(model._lane_q.queue)
def passes(tag):
    for li, lane in enumerate(lanes):
        st = lane.stream
        ts = []
        for rep in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            with torch.cuda.stream(st):
                lane.flow.inference_batch([item] * 4, streaming=True, finalize=False)
            t1 = time.perf_counter(); torch.cuda.synchronize(); ts.append(((t1 - t0) * 1e3, (time.perf_counter() - t0) * 1e3))
        print("%s lane %d: host / total ms per 4-request first-chunk pass: %s" % (tag, li, ["%.1f/%.1f" % x for x in ts]), flush=True)
passes("fresh ")
r = B.batched_decode(model, u, 16, 1)
print("batched_decode 16:", r["audio_s_per_s"], flush=True)
passes("after ")
