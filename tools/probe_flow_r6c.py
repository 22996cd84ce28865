import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cosyvoice_amd import synthetic as W
from cosyvoice_amd.flow import CausalMaskedDiffWithXvec
lc, fc, hc = W.cv2()
u = W.synthetic_utterance(lc, fc)
t = lambda n: torch.tensor([n], dtype=torch.int32)
flow = CausalMaskedDiffWithXvec(W.make_flow(fc), fc, precision="bf16")
tok = torch.randint(0, fc.vocab, (1, 250), generator=torch.Generator().manual_seed(0), dtype=torch.int32)
opt = lambda k, v: flow.lib.cv_flow_set_option(flow._h, k, C.c_int32(v))
def run():
    return flow.inference(token=tok, token_len=t(250), prompt_token=u["flow_prompt_speech_token"], prompt_token_len=t(87), prompt_feat=u["prompt_speech_feat"],
                          prompt_feat_len=t(174), embedding=u["flow_embedding"], streaming=False, finalize=True)[0]
ref = None
for rep in range(2):
    for name, kv in (("default", {}), ("ntile2", {b"flow_ntile": 2}), ("ntile1", {b"flow_ntile": 1}), ("tile1 64x64", {b"flow_tile": 1}), ("big_rows 1000", {b"big_rows": 1000}), ("graph off", {b"use_graph": 0})):
        opt(b"flow_ntile", 0); opt(b"flow_tile", 0); opt(b"big_rows", 2000); opt(b"use_graph", 1)
        for k, v in kv.items(): opt(k, v)
        for _ in range(3): run()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): mel = run()
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 10 * 1e3
        ref = mel if ref is None else ref
        print("%-16s %6.2f ms, bits equal %s" % (name, ms, bool(torch.equal(mel, ref))), flush=True)
