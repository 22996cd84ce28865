"""Round 6 (late): one flow.inference at the U10 size (batch 1: T = 674, 1348 estimator rows, the captured solve) with attn_flow32_kernel on 64- or 128-query workgroups
(option attn32_waves) and the residual GEMMs on 32 x 64 or 32 x 32 tiles (option res_tile); every combination returns the same bits.

    gpurun -- python tools/probe_flow_r6b.py
"""
import ctypes as C
import os
import sys
import time

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
    for waves, res in ((4, 0), (2, 0), (4, 1), (2, 1), (0, 0)):
        opt(b"attn32_waves", waves); opt(b"res_tile", res)
        for _ in range(3):
            run()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            mel = run()
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 10 * 1e3
        ref = mel if ref is None else ref
        print("attn32_waves %d res_tile %d: %6.2f ms per flow.inference, bits equal: %s" % (waves, res, ms, bool(torch.equal(mel, ref))), flush=True)
