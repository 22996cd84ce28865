"""Dev probe, round 4: what running the lock-step LM and the vocoder side by side costs each of them.  16 U10 sequences through the batched decode (LM thread / stream)
and 16 finished utterances through token2wav_batch (two lanes, 8 per shared pass) - each alone, then together; then the same with the flow's large-M GEMMs as
persistent workgroups that leave room on every CU (big_persist = n workgroups per CU).      gpurun -- python tools/probe_overlap.py"""
import ctypes as C, json, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B

model, u, cfgs = B.build_model("bf16")
model.set_lanes(2)
model.flow_batch = 8
NB = 16
ratio = B.N_GEN / B.N_TEXT
gold = json.load(open(os.path.join(B.ROOT, "tests", "golden", "u10_oracle_tokens.json")))["tokens"]
lm_reqs = [dict(text=u["text"], prompt_text=u["prompt_text"], prompt_speech_token=u["llm_prompt_speech_token"]) for _ in range(NB)]
tok = torch.tensor(gold).unsqueeze(0)


def lm_once():
    with model.llm_context:
        toks = model.llm.inference_batch(lm_reqs, max_token_text_ratio=ratio, min_token_text_ratio=ratio)
        torch.cuda.current_stream().synchronize()
    assert [int(x) for x in toks[0]] == gold


def voc_group(tag):
    jobs = [dict(token=tok, prompt_token=u["flow_prompt_speech_token"], prompt_feat=u["prompt_speech_feat"], embedding=u["flow_embedding"], token_offset=0, uuid="%s%d" % (tag, i))
            for i in range(8)]
    for j in jobs:
        model.hift_cache_dict[j["uuid"]] = None
    outs = model.token2wav_batch(jobs, stream=False, finalize=True)
    for j in jobs:
        model.hift_cache_dict.pop(j["uuid"], None)
    return [o.cpu() for o in outs]


def voc_once():
    th = [threading.Thread(target=voc_group, args=(t,)) for t in ("a", "b")]
    for t in th: t.start()
    for t in th: t.join()


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / reps


def both():
    a = threading.Thread(target=lm_once); b = threading.Thread(target=voc_once)
    a.start(); b.start(); a.join(); b.join()


def set_flow(**kw):
    lanes = []
    while not model._lane_q.empty():
        lanes.append(model._lane_q.get_nowait())
    for ln in lanes:
        for k, v in kw.items():
            ln.flow.lib.cv_flow_set_option(ln.flow._h, k.encode(), C.c_int32(v))
        model._lane_q.put(ln)


for label, kw in (("default (one tile per workgroup)", dict(big_persist=-1)), ("big GEMMs persistent, 3 workgroups per CU", dict(big_persist=3)),
                  ("big GEMMs persistent, 2 workgroups per CU", dict(big_persist=2)), ("small-tile kernels", dict(big_rows=0))):
    set_flow(big_rows=4000, big_persist=-1)
    set_flow(**kw)
    t_lm, t_voc, t_both = timed(lm_once), timed(voc_once), timed(both)
    print("%-46s LM alone %7.1f ms | vocoder alone (16 utterances, 2 lanes) %7.1f ms | together %7.1f ms = %.2f x max, %.2f x sum" %
          (label, t_lm, t_voc, t_both, t_both / max(t_lm, t_voc), t_both / (t_lm + t_voc)), flush=True)
