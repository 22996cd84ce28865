"""Dev tool (round 6; attn_splits other than 4 | 8 | 16 need the matching gemv_kernel<2,1,4,NSP> instantiation added to csrc/llm.hip gemv()): the decode attention (category 1) and the o_proj GEMV with its partial merge (category 2) as hipGraph chains of their 24 per-layer launches
(cv_llm_profile_chain) against attn_splits, at the context a U10 request ends with (381 positions)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from cosyvoice_amd._lib import stream_ptr
model, u, cfgs = bench.build_model("bf16")
llm = model.llm
t = lambda n: torch.tensor([n], dtype=torch.int32)
ratio = bench.N_GEN / bench.N_TEXT
for nsp in [int(a) for a in sys.argv[1:]] or (8, 4, 6, 10, 12, 16, 8):
    llm.lib.cv_llm_set_option(llm._h, b"attn_splits", C.c_int32(nsp))
    with model.llm_context:
        toks = list(llm.inference(text=u["text"], text_len=t(bench.N_TEXT), prompt_text=u["prompt_text"], prompt_text_len=t(bench.N_PROMPT_TEXT), prompt_speech_token=u["llm_prompt_speech_token"],
                                  prompt_speech_token_len=t(bench.N_PROMPT_TOK), embedding=u["llm_embedding"], max_token_text_ratio=ratio, min_token_text_ratio=ratio))
        out = []
        for k in (0, 1, 2, 3, 4):
            ms, n = C.c_float(), C.c_int32()
            llm.lib.cv_llm_profile_chain(llm._h, k, 20, C.byref(ms), C.byref(n), stream_ptr(llm.lib))
            out.append(1e3 * ms.value / max(n.value, 1))
    print("attn_splits %2d: chain us per launch  qkv %.2f  attention %.2f  o_proj+merge %.2f  gate_up %.2f  down %.2f   layer %.2f" % (nsp, *out, sum(out)), flush=True)
