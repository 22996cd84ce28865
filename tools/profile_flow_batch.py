"""Dev tool: `nu` utterances of U10 shape through ONE flow pass (cv_flow_inference_batch), 2 warm + 3 passes, nothing else -
for `rocprofv3 --kernel-trace --stats` / `--pmc` (round 4: per-kernel evidence of the batched pass).   python tools/profile_flow_batch.py <nu> [opt=value ...]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cosyvoice_amd import synthetic as W
from cosyvoice_amd.flow import CausalMaskedDiffWithXvec

nu = int(sys.argv[1]) if len(sys.argv) > 1 else 1
lc, fc, hc = W.cv2()
u = W.synthetic_utterance(lc, fc)
flow = CausalMaskedDiffWithXvec(W.make_flow(fc), fc, precision="bf16")
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    flow.lib.cv_flow_set_option(flow._h, k.encode(), C.c_int32(int(v)))
g = torch.Generator().manual_seed(0)
tok = torch.randint(0, fc.vocab, (1, 250), generator=g, dtype=torch.int32)
item = dict(token=tok, prompt_token=u["flow_prompt_speech_token"], prompt_feat=u["prompt_speech_feat"], embedding=u["flow_embedding"])
for _ in range(2):
    flow.inference_batch([item] * nu)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3):
    flow.inference_batch([item] * nu)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 3 * 1e3
print("flow pass over %d utterance(s): %.2f ms = %.2f ms per utterance" % (nu, ms, ms / nu), flush=True)
