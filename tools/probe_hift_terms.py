"""Dev tool (round 6): HiFT at full size (500 frames = 10 s) with the decoder's convolutions on six (fp32-exact class, default) or three plane products of the
two-sided bf16 split (option "terms"): time per call, and the 3-term waveform against the 6-term one and against the fp32 CPU oracle with the same SineGen noise."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cosyvoice_amd import synthetic as W
from cosyvoice_amd.hift import HiFTGenerator
lc, fc, hc = W.cv2()
sd = W.make_hift(hc)
hift = HiFTGenerator(sd, hc)
g = torch.Generator().manual_seed(1)
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 500
mel = (torch.randn(1, 80, frames, generator=g) * 2 - 5).cuda()
noise = torch.randn(frames * 480, 9, generator=g)
outs = {}
for terms in (6, 3, 6, 3):
    hift.lib.cv_hift_set_option(hift._h, b"terms", C.c_int32(terms))
    sp, src = hift.inference(mel, noise=noise)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        sp, src = hift.inference(mel, noise=noise)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    outs[terms] = sp.cpu()
    print("terms %d: %.3f ms per call (%d frames)" % (terms, ms, frames))
a, b = outs[6].double(), outs[3].double()
err = (a - b)
print("3 terms vs 6 terms: max |d| %.3e  rel-L2 %.3e  SNR %.1f dB  (|wav| max %.3f rms %.4f)" % (err.abs().max(), err.norm() / a.norm(), 20 * torch.log10(a.norm() / err.norm()), a.abs().max(), a.pow(2).mean().sqrt()))
if "oracle" in sys.argv:
    from oracle import hift as OH                              # dev probe: the checker, not the product
    t0 = time.perf_counter()
    want = OH.inference(sd, hc, mel.cpu(), noise=noise)[0].double()
    print("oracle (fp32 CPU) in %.1f s" % (time.perf_counter() - t0))
    for terms in (6, 3):
        e = outs[terms].double() - want
        print("terms %d vs oracle: max |d| %.3e  rel-L2 %.3e  SNR %.1f dB" % (terms, e.abs().max(), e.norm() / want.norm(), 20 * torch.log10(want.norm() / e.norm())))
