"""Dev probe (round 3): HOST time of the big graph launches - how long the calling thread spends inside flow.inference (one ~3600-node hipGraphLaunch, taken
under the process-wide runtime lock) before the GPU has finished, and inside one 8-step LM decode burst - i.e. how long one thread can keep the other from
launching.   gpurun -- python tools/probe_host_launch.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B

model, u, cfgs = B.build_model("bf16")
for _ in range(3):
    B.one_utterance(model, u)
t = lambda n: torch.tensor([n], dtype=torch.int32)
tok = torch.randint(0, 6561, (1, 250), dtype=torch.int32)
for n_tok, streaming, fin in ((250, False, True), (41, True, False)):
    tk = tok[:, :n_tok]
    for rep in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        mel, _ = model.flow.inference(token=tk, token_len=t(n_tok), prompt_token=u["flow_prompt_speech_token"], prompt_token_len=t(87), prompt_feat=u["prompt_speech_feat"],
                                      prompt_feat_len=t(174), embedding=u["flow_embedding"], streaming=streaming, finalize=fin)
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print("flow.inference(%d tokens, streaming=%s): host returns after %.2f ms, GPU done after %.2f ms" % (n_tok, streaming, (t1 - t0) * 1e3, (t2 - t0) * 1e3), flush=True)
lm = model.llm
with model.llm_context:
    lm.prefill(lm.build_lm_input(u["text"], u["prompt_text"], u["llm_prompt_speech_token"]))
    sp = lm.make_sampling(250, 250)
    for rep in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        lm.decode(8, sp)
        t1 = time.perf_counter()
        print("llm.decode(8 steps): %.2f ms (returns after its own synchronisation)" % ((t1 - t0) * 1e3), flush=True)
