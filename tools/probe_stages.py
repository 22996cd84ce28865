"""Dev probe: where one U10 utterance of bench.py spends its wall time (same model object and call path as the timed region)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B

model, u, cfgs = B.build_model("bf16")
for _ in range(3):
    B.one_utterance(model, u)
t = lambda n: torch.tensor([n], dtype=torch.int32)
ratio = B.N_GEN / B.N_TEXT
acc = {}
flow_inf, hift_inf = model.flow.inference, model.hift.inference


def timed(name, fn):
    def w(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize(); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
        return r
    return w


model.flow.inference, model.hift.inference = timed("flow.inference", flow_inf), timed("hift.inference", hift_inf)
reps = 5
tot = 0.0
for _ in range(reps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with model.llm_context:
        tokens = list(model.llm.inference(text=u["text"], text_len=t(B.N_TEXT), prompt_text=u["prompt_text"], prompt_text_len=t(B.N_PROMPT_TEXT),
                                          prompt_speech_token=u["llm_prompt_speech_token"], prompt_speech_token_len=t(B.N_PROMPT_TOK),
                                          embedding=u["llm_embedding"], max_token_text_ratio=ratio, min_token_text_ratio=ratio))
    torch.cuda.synchronize(); t1 = time.perf_counter()
    model.hift_cache_dict["p"] = None
    wav = model.token2wav(token=torch.tensor(tokens).unsqueeze(0), prompt_token=u["flow_prompt_speech_token"], prompt_feat=u["prompt_speech_feat"],
                          embedding=u["flow_embedding"], token_offset=0, uuid="p", finalize=True)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    out = wav.cpu(); t3 = time.perf_counter()
    acc["llm (prefill + 250 tokens)"] = acc.get("llm (prefill + 250 tokens)", 0.0) + t1 - t0
    acc["token2wav total"] = acc.get("token2wav total", 0.0) + t2 - t1
    acc[".cpu()"] = acc.get(".cpu()", 0.0) + t3 - t2
    tot += t3 - t0
for k, v in acc.items():
    print("%-28s %.2f ms" % (k, 1e3 * v / reps))
print("%-28s %.2f ms (with the extra synchronisations of this probe)" % ("utterance", 1e3 * tot / reps))
