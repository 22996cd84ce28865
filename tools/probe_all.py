"""Dev probe: full-size CosyVoice2 stage timings on the MI355X (random weights, U10 utterance)."""
import os, sys, time
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
def sync(): torch.cuda.synchronize()
if "llm" in which:
    lm = Qwen2LM(W.make_llm(lc), lc, max_len=1024, sampling="greedy", decode_chunk=32)
    for rep in range(3):
        sync(); t0 = time.time()
        x = lm.build_lm_input(u["text"], u["prompt_text"], u["llm_prompt_speech_token"]); lm.prefill(x); sync(); t1 = time.time()
        sp = lm.make_sampling(250, 250); n = 0
        while n < 250:
            toks, fin = lm.decode(min(32, 250 - n + 1), sp); n += len(toks)
            if fin: break
        sync(); t2 = time.time()
        print("llm rep%d prefill=%.2f ms decode(%d)=%.2f ms (%.1f us/tok)" % (rep, (t1 - t0) * 1e3, n, (t2 - t1) * 1e3, (t2 - t1) * 1e6 / n), flush=True)
    del lm
if "flow" in which:
    flow = CausalMaskedDiffWithXvec(W.make_flow(fc), fc)
    g = torch.Generator().manual_seed(0)
    tok = torch.randint(0, fc.vocab, (1, 250), generator=g, dtype=torch.int32)
    for rep in range(3):
        sync(); t0 = time.time()
        mel, _ = flow.inference(token=tok, token_len=t(250), prompt_token=u["flow_prompt_speech_token"], prompt_token_len=t(87), prompt_feat=u["prompt_speech_feat"],
                                prompt_feat_len=t(174), embedding=u["flow_embedding"], streaming=False, finalize=True)
        sync(); t1 = time.time()
        print("flow rep%d T=674 10 steps: %.2f ms  mel %s finite=%s" % (rep, (t1 - t0) * 1e3, tuple(mel.shape), bool(torch.isfinite(mel).all())), flush=True)
    del flow
if "hift" in which:
    hift = HiFTGenerator(W.make_hift(hc), hc)
    g = torch.Generator().manual_seed(1)
    mel = (torch.randn(1, 80, 500, generator=g) * 2 - 5).cuda()
    for rep in range(3):
        sync(); t0 = time.time()
        sp, src = hift.inference(mel)
        sync(); t1 = time.time()
        print("hift rep%d 500 frames: %.2f ms finite=%s" % (rep, (t1 - t0) * 1e3, bool(torch.isfinite(sp).all())), flush=True)
