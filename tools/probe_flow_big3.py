"""Dev probe, round 4 (third): where the large-M kernel set (64 x 64 tiles) starts to pay - the shared flow pass over 1, 2, 3, 4 copies of U10 on the small-tile kernels
(big_rows = 0) and on the large-M set (big_rows = 1), both bit-identical.     python tools/probe_flow_big3.py"""
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


def run(nu, label, **kw):
    for k, v in kw.items():
        flow.lib.cv_flow_set_option(flow._h, k.encode(), C.c_int32(v))
    for _ in range(3):
        out = flow.inference_batch([item] * nu)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        out = flow.inference_batch([item] * nu)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    print("nu=%d %-44s %8.2f ms = %6.2f ms per utterance" % (nu, label, ms, ms / nu), flush=True)
    return out[0].clone()


for nu in (1, 2, 3, 4):
    a = run(nu, "small tiles", big_rows=0, graph_max_rows=3000)
    b = run(nu, "large-M set, 64 x 64 (graph below 3000 rows)", big_rows=1, big_tile0=3, big_tile1=3, big_persist=-1)
    c = run(nu, "large-M set, 64 x 64, always a graph", big_rows=1, big_tile0=3, big_tile1=3, big_persist=-1, graph_max_rows=1000000)
    print("   bit-identical:", torch.equal(a, b), torch.equal(a, c), flush=True)
