"""Dev tool (round 5): the band launch of a large flow pass with and without the next block's QKV GEMM folded in, by rows per band.
`nu` utterances of U10 through ONE flow pass (cv_flow_inference_batch), 2 warm + 3 timed passes per variant; every variant's mel must equal the first one's bit for bit.
    python tools/probe_band.py "8 6 4 3 2" """
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cosyvoice_amd import synthetic as W
from cosyvoice_amd.flow import CausalMaskedDiffWithXvec

nus = [int(x) for x in (" ".join(sys.argv[1:]) or "8 6 4 3 2 1").split()]
lc, fc, hc = W.cv2()
u = W.synthetic_utterance(lc, fc)
flow = CausalMaskedDiffWithXvec(W.make_flow(fc), fc, precision="bf16")
g = torch.Generator().manual_seed(0)
tok = torch.randint(0, fc.vocab, (1, 250), generator=g, dtype=torch.int32)
item = dict(token=tok, prompt_token=u["flow_prompt_speech_token"], prompt_feat=u["prompt_speech_feat"], embedding=u["flow_embedding"])


def opt(**kw):
    for k, v in kw.items():
        flow.lib.cv_flow_set_option(flow._h, k.encode(), C.c_int32(int(v)))


VARIANTS = [("small-tile path    ", dict(big_rows=1 << 30, fused_band=0, band_qkv=0, band_bm=0)),
            ("five launches      ", dict(big_rows=1, fused_band=0, band_qkv=0, band_bm=0)),
            ("band, rule         ", dict(fused_band=1, band_qkv=0, band_bm=0)),
            ("band + qkv, rule   ", dict(fused_band=1, band_qkv=1, band_bm=0)),
            ("band + qkv, 64 rows", dict(fused_band=1, band_qkv=1, band_bm=64)),
            ("band + qkv, 48 rows", dict(fused_band=1, band_qkv=1, band_bm=48)),
            ("band + qkv, 32 rows", dict(fused_band=1, band_qkv=1, band_bm=32)),
            ("band, 48 rows      ", dict(fused_band=1, band_qkv=0, band_bm=48)),
            ("+ qkv, 48, no pipe ", dict(fused_band=1, band_qkv=1, band_bm=48, band_pipe=0)),       # band_pipe: the FF chunks of a 48-row (2: and 32-row) band as a software pipeline
            ("+ qkv, 32, pipe    ", dict(fused_band=1, band_qkv=1, band_bm=32, band_pipe=2)),
            ("+ qkv, rule, pipe 2", dict(fused_band=1, band_qkv=1, band_bm=0, band_pipe=2)),
            ("+ qkv, rule, pipe 0", dict(fused_band=1, band_qkv=1, band_bm=0, band_pipe=0)),
            ("2 chains, rule     ", dict(fused_band=1, band_qkv=1, band_bm=0, est_streams=2)),       # the batch rows as two launch chains on two streams (est_streams, round 3): eager passes only
            ("2 chains, 64 rows  ", dict(fused_band=1, band_qkv=1, band_bm=64, est_streams=2)),
            ("2 chains, 48 rows  ", dict(fused_band=1, band_qkv=1, band_bm=48, est_streams=2)),
            ("2 chains, 32 rows  ", dict(fused_band=1, band_qkv=1, band_bm=32, est_streams=2))]
for nu in nus:
    ref = None
    for name, kw in VARIANTS:
        opt(**{"est_streams": 1, "band_pipe": 1, **kw})
        for _ in range(2):
            out = flow.inference_batch([item] * nu)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            out = flow.inference_batch([item] * nu)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 3 * 1e3
        mel = out[0].cpu()
        if ref is None:
            ref = mel
        print("%d utterance(s), M = %5d  %s  %7.2f ms per pass = %6.2f ms per utterance   mel == first variant: %s" % (nu, 2 * nu * 674, name, ms, ms / nu, bool(torch.equal(mel, ref))), flush=True)
opt(big_rows=2000, fused_band=1, band_qkv=1, band_bm=0, est_streams=1, band_pipe=1)
