"""Dev probe: flow.inference for U10 alone vs cv_flow_inference_batch over 2 / 4 / 8 copies (ms per utterance)."""
import os, sys, time
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
for nu in (1, 2, 4, 8):
    for _ in range(3):
        flow.inference_batch([item] * nu)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(4):
        flow.inference_batch([item] * nu)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 4 * 1e3
    print("flow pass over %d utterance(s): %.2f ms = %.2f ms per utterance" % (nu, ms, ms / nu), flush=True)

# round 3: the one-launch block tail (flow_tail.h) against the five-launch block when 4 / 8 utterances share a pass (M = 5392 / 10784 rows)
if len(sys.argv) > 1 and sys.argv[1] == "tail":
    import ctypes as C
    for nu in (1, 4, 8):
        for tail in (0, 1):
            flow.lib.cv_flow_set_option(flow._h, b"fused_tail", C.c_int32(tail))
            for _ in range(2):
                flow.inference_batch([item] * nu)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(4):
                flow.inference_batch([item] * nu)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 4 * 1e3
            print("nu=%d fused_tail=%d  %.2f ms = %.2f ms per utterance" % (nu, tail, ms, ms / nu), flush=True)
    flow.lib.cv_flow_set_option(flow._h, b"fused_tail", C.c_int32(0))

# tile / attention variants at 4 utterances per pass (M = 4 x 1348 rows: enough workgroups that bigger tiles may pay)
if len(sys.argv) > 1 and sys.argv[1] == "sweep":
    import ctypes as C
    def opt(name, v):
        flow.lib.cv_flow_set_option(flow._h, name.encode(), C.c_int32(v))
    for nu in (4, 8):
        for name, sets in (("default", {}), ("tile 64x64", {"flow_tile": 1}), ("tile 64x128", {"flow_tile": 2}), ("attn ks=1 kt=2", {"attn_ks": 1, "attn_kt": 2}),
                           ("attn ks=1 kt=1", {"attn_ks": 1, "attn_kt": 1})):
            for k, v in sets.items():
                opt(k, v)
            for _ in range(2):
                flow.inference_batch([item] * nu)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(4):
                flow.inference_batch([item] * nu)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 4 * 1e3
            print("nu=%d %-16s %.2f ms = %.2f ms per utterance" % (nu, name, ms, ms / nu), flush=True)
            opt("flow_tile", 0); opt("attn_ks", 2); opt("attn_kt", 1)

# padded pass over utterances of different lengths (cv_flow_inference_ragged) vs the same utterances one by one
if len(sys.argv) > 1 and sys.argv[1] == "ragged":
    for lens in ((250, 235, 220, 205), (250, 250, 200, 200), (500, 450, 420, 400), (125, 120, 110, 100)):
        its = [dict(item, token=torch.randint(0, fc.vocab, (1, k), generator=g, dtype=torch.int32)) for k in lens]
        def one_by_one():
            for it in its:
                flow.inference_batch([it])
        def padded():
            flow.inference_batch(its)
        res = []
        for fn in (one_by_one, padded):
            for _ in range(3):
                fn()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(4):
                fn()
            torch.cuda.synchronize()
            res.append((time.perf_counter() - t0) / 4 * 1e3)
        a = [flow.inference_batch([it])[0].clone() for it in its]
        b = flow.inference_batch(its)
        same = all(torch.equal(x, y) for x, y in zip(a, b))
        print("tokens %s: one by one %.2f ms, one padded pass %.2f ms (x%.2f), bit-identical %s" % (lens, res[0], res[1], res[0] / res[1], same), flush=True)
