"""Dev tool: the headline hot path (bench.one_utterance) 3 x warm + 2 x, nothing else - for rocprofv3 --kernel-trace and tools/trace_gaps.py."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B

model, u, cfgs = B.build_model("bf16")
for _ in range(3):
    B.one_utterance(model, u)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(2):
    B.one_utterance(model, u)
torch.cuda.synchronize()
print("2 utterances: %.2f ms each" % ((time.perf_counter() - t0) * 500), flush=True)
