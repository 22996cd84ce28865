"""Bounded workload for rocprofv3 (kernel trace / PMC): 1 flow inference (T=674, 10 steps) + 1 HiFT (500 frames) + 24 LLM decode steps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cosyvoice_amd import synthetic as W
from cosyvoice_amd.llm import Qwen2LM
from cosyvoice_amd.flow import CausalMaskedDiffWithXvec
from cosyvoice_amd.hift import HiFTGenerator
lc, fc, hc = W.cv2()
u = W.synthetic_utterance(lc, fc)
t = lambda n: torch.tensor([n], dtype=torch.int32)
which = sys.argv[1:] or ["llm", "flow", "hift"]
if "llm" in which:
    lm = Qwen2LM(W.make_llm(lc), lc, max_len=1024, sampling="greedy", decode_chunk=24, use_graph=False)
    lm.prefill(lm.build_lm_input(u["text"], u["prompt_text"], u["llm_prompt_speech_token"]))
    sp = lm.make_sampling(250, 250)
    # advance to a mid-utterance context (L ~ 256) without tracing every token: the graph path is not used here on purpose
    toks, fin = lm.decode(24, sp)
    torch.cuda.synchronize(); del lm
if "flow" in which:
    flow = CausalMaskedDiffWithXvec(W.make_flow(fc), fc, precision=os.environ.get("FLOW_PRECISION", "bf16"))
    tok = torch.randint(0, fc.vocab, (1, 250), generator=torch.Generator().manual_seed(0), dtype=torch.int32)
    mel, _ = flow.inference(token=tok, token_len=t(250), prompt_token=u["flow_prompt_speech_token"], prompt_token_len=t(87), prompt_feat=u["prompt_speech_feat"],
                            prompt_feat_len=t(174), embedding=u["flow_embedding"], streaming=False, finalize=True)
    torch.cuda.synchronize(); del flow
if "hift" in which:
    hift = HiFTGenerator(W.make_hift(hc), hc)
    mel = (torch.randn(1, 80, 500, generator=torch.Generator().manual_seed(1)) * 2 - 5).cuda()
    sp_, src = hift.inference(mel)
    torch.cuda.synchronize()
print("profile workload done")
