"""Dev tool (round 6): does starting the bands of a launch out of phase help?  One flow pass over `nu` utterances of U10 shape
(cv_flow_inference_batch) with option "band_stagger" = n: band b sleeps (b mod 4) * n * ~0.9 us before its first load (FlowBandArgs::stagger).
    python tools/probe_band_stagger.py [nu ...]        ->  profiles/r6_band_stagger.txt"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cosyvoice_amd import synthetic as W
from cosyvoice_amd.flow import CausalMaskedDiffWithXvec

lc, fc, hc = W.cv2()
u = W.synthetic_utterance(lc, fc)
flow = CausalMaskedDiffWithXvec(W.make_flow(fc), fc, precision="bf16")
g = torch.Generator().manual_seed(0)
tok = torch.randint(0, fc.vocab, (1, 250), generator=g, dtype=torch.int32)
item = dict(token=tok, prompt_token=u["flow_prompt_speech_token"], prompt_feat=u["prompt_speech_feat"], embedding=u["flow_embedding"])


def run(nu, n, reps=4):
    flow.lib.cv_flow_set_option(flow._h, b"band_stagger", C.c_int32(n))
    out = None
    for _ in range(2):
        out = flow.inference_batch([item] * nu)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        flow.inference_batch([item] * nu)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, out


for nu in [int(a) for a in sys.argv[1:]] or [8, 4]:
    base = None
    for n in (0, 1, 2, 3, 4, 6, 8, 0):
        ms, out = run(nu, n)
        mel = out[0][0] if isinstance(out[0], (tuple, list)) else out[0]
        if base is None:
            base = mel.clone()
        same = bool(torch.equal(mel, base))
        print("utterances %d  band_stagger %d (<= %.1f us late): %.2f ms per pass  bit-identical to stagger 0: %s" % (nu, n, 3 * n * 0.9, ms, same), flush=True)
