"""Dev tool (round 6): HiFT at full size (500 frames) with the ResBlock operands as pre-split bf16 planes (option "presplit", default) or as fp32 values split by every
consumer tile (0): time per call and bit-equality of the two waveforms.   python tools/probe_hift_presplit.py [frames]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cosyvoice_amd import synthetic as W
from cosyvoice_amd.hift import HiFTGenerator
lc, fc, hc = W.cv2()
hift = HiFTGenerator(W.make_hift(hc), hc)
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 500
mel = (torch.randn(1, 80, frames, generator=torch.Generator().manual_seed(1)) * 2 - 5).cuda()
outs = {}
for pre in (1, 0, 1, 0):
    hift.lib.cv_hift_set_option(hift._h, b"presplit", C.c_int32(pre))
    for _ in range(3):
        w, _ = hift.inference(mel, seed=7)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(40):
        w, _ = hift.inference(mel, seed=7)
    torch.cuda.synchronize()
    outs[pre] = w.cpu()
    print("presplit %d: %.3f ms per call (%d frames)" % (pre, (time.perf_counter() - t0) / 40 * 1e3, frames), flush=True)
print("bit-identical:", bool(torch.equal(outs[0], outs[1])), " finite:", bool(torch.isfinite(outs[1]).all()), " |wav| max %.3f" % outs[1].abs().max())
