"""Dev probe: full-size CosyVoice2 LLM decode timing on the MI355X (random weights)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cosyvoice_amd import synthetic as W
from cosyvoice_amd.llm import Qwen2LM

cfg = W.cv2()[0]
t0 = time.time(); sd = W.make_llm(cfg); print("weights %.1fs" % (time.time() - t0), flush=True)
u = W.synthetic_utterance(cfg, W.cv2()[1])
for use_graph, splits in ((True, 4), (True, 8), (True, 16), (False, 8)):
    lm = Qwen2LM(sd, cfg, max_len=1024, sampling="greedy", decode_chunk=int(os.environ.get("CHUNK", "32")), use_graph=use_graph, attn_splits=splits)
    t = lambda n: torch.tensor([n], dtype=torch.int32)
    kw = dict(text=u["text"], text_len=t(30), prompt_text=u["prompt_text"], prompt_text_len=t(12), prompt_speech_token=u["llm_prompt_speech_token"],
              prompt_speech_token_len=t(87), embedding=None, max_token_text_ratio=250 / 30, min_token_text_ratio=250 / 30)
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.time()
        lm_input = lm.build_lm_input(u["text"], u["prompt_text"], u["llm_prompt_speech_token"])
        lm.prefill(lm_input); torch.cuda.synchronize(); t1 = time.time()
        sp = lm.make_sampling(250, 250)
        n = 0
        while n < 250:
            toks, fin = lm.decode(min(lm.decode_chunk, 250 - n + 1), sp); n += len(toks)
            if fin: break
        torch.cuda.synchronize(); t2 = time.time()
        print("graph=%s splits=%d rep%d prefill(131)=%.2f ms decode(%d tok)=%.2f ms -> %.1f us/token" % (use_graph, splits, rep, (t1 - t0) * 1e3, n, (t2 - t1) * 1e3, (t2 - t1) * 1e6 / max(n, 1)), flush=True)
    del lm
