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
