"""Dev tool (round 6): cProfile of bench.py's timed step (one_utterance: llm.inference -> token2wav -> .cpu()) on the MI355X - which Python frames hold the wall time that is
not kernel time.   python tools/probe_host_overhead.py [reps]"""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
model, u, cfgs = bench.build_model("bf16") if hasattr(bench, "build_model") else (None, None, None)
for _ in range(4):
    bench.one_utterance(model, u)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(reps):
    bench.one_utterance(model, u)
torch.cuda.synchronize()
print("one_utterance: %.2f ms" % ((time.perf_counter() - t0) / reps * 1e3), flush=True)
pr = cProfile.Profile()
pr.enable()
for _ in range(reps):
    bench.one_utterance(model, u)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue())
