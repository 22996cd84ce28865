"""Dev tool (round 6): HiFT at full size (500 frames = 10 s), in-kernel noise: wall time per call; under `rocprofv3 --kernel-trace --stats` the kernel totals / calls
give the GPU-busy time per call - the difference is what the 70-odd launches of a call spend between kernels (tools/gpu_run.sh <tag> profpy:probe_hift_busy.py)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cosyvoice_amd import synthetic as W
from cosyvoice_amd.hift import HiFTGenerator
lc, fc, hc = W.cv2()
hift = HiFTGenerator(W.make_hift(hc), hc)
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 500
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 40
mel = (torch.randn(1, 80, frames, generator=torch.Generator().manual_seed(1)) * 2 - 5).cuda()
for _ in range(3):
    hift.inference(mel)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(calls):
    hift.inference(mel)
torch.cuda.synchronize()
print("hift.inference: %.3f ms per call (%d frames, %d + 3 calls in this process)" % ((time.perf_counter() - t0) / calls * 1e3, frames, calls))
