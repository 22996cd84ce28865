"""Dev probe, round 4 (second): the persistent large-M GEMMs (flow_big.h) - workgroups per CU x tile shape at 4 and 8 utterances per pass, against the one-tile-per-workgroup
form (big_persist = -1) and the small-tile kernels.  `profile`: nu = 8 only, 1 warm + 2 timed passes (for rocprofv3 --kernel-trace / --pmc).
    python tools/probe_flow_big2.py [profile] [nu=8] [cfg=<persist>,<tile0>,<tile1>]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cosyvoice_amd import synthetic as W
from cosyvoice_amd.flow import CausalMaskedDiffWithXvec

profile = "profile" in sys.argv
only = [a[4:] for a in sys.argv if a.startswith("cfg=")]
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
    warm, reps = (1, 2) if profile else (2, 4)
    for _ in range(warm):
        out = flow.inference_batch([item] * nu)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        out = flow.inference_batch([item] * nu)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    print("nu=%d %-52s %8.2f ms = %6.2f ms per utterance" % (nu, label, ms, ms / nu), flush=True)
    return out[0].clone()


if only:
    ps, t0, t1 = (int(x) for x in only[0].split(","))
    run(8, "persist=%d tile0=%d tile1=%d" % (ps, t0, t1), big_rows=1, attn2_rows=0, big_persist=ps, big_tile0=t0, big_tile1=t1)
    sys.exit(0)
for nu in ((8,) if profile else (4, 8)):
    ref = run(nu, "small tiles", big_rows=0, attn2_rows=0)
    same = []
    for ps in (-1, 0, 1, 2, 3):
        for t0, t1 in ((3, 3), (2, 2), (1, 2), (1, 1)):
            if ps == 3 and t0 == 1:
                continue                                        # 128 x 128: 64 KB of LDS per workgroup, two per CU at most
            o = run(nu, "big, persist=%2d (wg per CU, -1: tile per wg) tile0=%d tile1=%d" % (ps, t0, t1), big_rows=1, attn2_rows=0, big_persist=ps, big_tile0=t0, big_tile1=t1)
            same.append(torch.equal(o, ref))
    print("   all bit-identical to the small-tile pass:", all(same), flush=True)
opt(big_rows=5000, attn2_rows=0, big_tile0=0, big_tile1=0, big_persist=0)
