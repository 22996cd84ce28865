"""Dev tool (round 6, CV_BUILD_EXPERIMENTS build): HiFT at full size with the two-sided-split convolutions on the double-buffered loop (CV_GEMM_WX3_DB = 1: half-depth k tiles,
two LDS buffers, one barrier per k-step; 2: the same at full depth for 32-row tiles) against the default loop: time per call and bit-equality."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cosyvoice_amd import synthetic as W
from cosyvoice_amd.hift import HiFTGenerator
lc, fc, hc = W.cv2()
hift = HiFTGenerator(W.make_hift(hc), hc)
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 500
mel = (torch.randn(1, 80, frames, generator=torch.Generator().manual_seed(1)) * 2 - 5).cuda()
outs = {}
for db, mb in (("0", "720"), ("1", "720"), ("1", "480"), ("1", "1000"), ("2", "720"), ("0", "720"), ("1", "720"), ("1", "480")):
    os.environ["CV_GEMM_WX3_DB"] = db; os.environ["CV_GEMM_MIN_BLOCKS_F32"] = mb
    for _ in range(3):
        w, _ = hift.inference(mel, seed=7)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(40):
        w, _ = hift.inference(mel, seed=7)
    torch.cuda.synchronize()
    outs[db, mb] = w.cpu()
    print("DB %s min_blocks %s: %.3f ms per call (%d frames)  bit-identical to the default loop: %s" % (db, mb, (time.perf_counter() - t0) / 40 * 1e3, frames, bool(torch.equal(outs[db, mb], outs["0", "720"]))), flush=True)
