"""Dev probe, round 4: the shared flow pass (cv_flow_inference_batch over nu copies of U10) with the small-tile kernels against the large-M kernel set of flow_big.h
(options big_rows / attn2_rows / big_tile0 / big_tile1), ms per utterance per configuration.  Under `rocprofv3 --kernel-trace --stats` the kernel names carry the tile
template arguments, so one run gives the per-kernel averages of every variant (`profile` = 1 warm + 2 timed passes per configuration).
    python tools/probe_flow_big.py [profile] [check]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cosyvoice_amd import synthetic as W
from cosyvoice_amd.flow import CausalMaskedDiffWithXvec

profile, check = "profile" in sys.argv, "check" in sys.argv
lc, fc, hc = W.cv2()
u = W.synthetic_utterance(lc, fc)
flow = CausalMaskedDiffWithXvec(W.make_flow(fc), fc, precision="bf16")
g = torch.Generator().manual_seed(0)
tok = torch.randint(0, fc.vocab, (1, 250), generator=g, dtype=torch.int32)
item = dict(token=tok, prompt_token=u["flow_prompt_speech_token"], prompt_feat=u["prompt_speech_feat"], embedding=u["flow_embedding"])


def opt(**kw):
    for k, v in kw.items():
        flow.lib.cv_flow_set_option(flow._h, k.encode(), C.c_int32(v))


def run(nu, label, **kw):
    opt(**kw)
    warm, reps = (1, 2) if profile else (3, 4)
    for _ in range(warm):
        out = flow.inference_batch([item] * nu)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        out = flow.inference_batch([item] * nu)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    print("nu=%d %-44s %8.2f ms = %6.2f ms per utterance" % (nu, label, ms, ms / nu), flush=True)
    return out[0].clone()


ref = {}
for nu in ((8,) if profile else (1, 2, 4, 8)):
    ref[nu] = run(nu, "small tiles (round-3 kernels, new LDS pitch)", big_rows=0, attn2_rows=0)
    a = run(nu, "big GEMMs 128x128 / 128x64", big_rows=1, attn2_rows=0, big_tile0=1, big_tile1=2)
    b = run(nu, "big GEMMs + 128-query attention", big_rows=1, attn2_rows=1, big_tile0=1, big_tile1=2)
    c = run(nu, "small GEMMs + 128-query attention", big_rows=0, attn2_rows=1)
    if check:
        print("   bit-identical to the small-tile pass:", torch.equal(ref[nu], a), torch.equal(ref[nu], b), torch.equal(ref[nu], c), " same across nu:", torch.equal(ref[nu], ref[min(ref)]), flush=True)
for nu in ((8,) if profile else (4, 8)):
    for t0, t1 in ((1, 1), (2, 2), (3, 3), (1, 3), (2, 3)):
        run(nu, "big GEMMs tile0=%d tile1=%d + attn2" % (t0, t1), big_rows=1, attn2_rows=1, big_tile0=t0, big_tile1=t1)
opt(big_rows=5000, attn2_rows=0, big_tile0=0, big_tile1=0)
