"""Dev tool (round 6): the batch-1 LM stage of U10 (prefill + 250 greedy tokens) against the decode attention's key-range slices per head (cv_llm option attn_splits 4 | 8 | 16)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
model, u, cfgs = bench.build_model("bf16")
llm = model.llm
t = lambda n: torch.tensor([n], dtype=torch.int32)
N = int(sys.argv[1]) if len(sys.argv) > 1 else bench.N_GEN
ratio = N / bench.N_TEXT
def run():
    with model.llm_context:
        return list(llm.inference(text=u["text"], text_len=t(bench.N_TEXT), prompt_text=u["prompt_text"], prompt_text_len=t(bench.N_PROMPT_TEXT), prompt_speech_token=u["llm_prompt_speech_token"],
                                  prompt_speech_token_len=t(bench.N_PROMPT_TOK), embedding=u["llm_embedding"], max_token_text_ratio=ratio, min_token_text_ratio=ratio))
base = None
for nsp in [int(a) for a in sys.argv[2:]] or (8, 4, 16, 8, 4, 16):
    llm.lib.cv_llm_set_option(llm._h, b"attn_splits", C.c_int32(nsp))
    for _ in range(3):
        toks = run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        toks = run()
    torch.cuda.synchronize()
    base = base or toks
    print("attn_splits %2d: %.2f ms per LM stage (%d tokens)  tokens equal to the first run: %s" % (nsp, (time.perf_counter() - t0) / 10 * 1e3, len(toks), toks == base), flush=True)
