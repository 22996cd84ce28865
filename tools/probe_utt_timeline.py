"""Dev probe: where the headline step's wall time goes between its stages (bench.one_utterance taken apart; host clocks, with and without a device synchronisation
after every stage).  The bench's stage split sums to ~173 ms of a 181 ms step: this prints the rest.
    gpurun -- python tools/probe_utt_timeline.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B

model, u, cfgs = B.build_model("bf16")
t = lambda n: torch.tensor([n], dtype=torch.int32)
ratio = B.N_GEN / B.N_TEXT
for _ in range(3):
    B.one_utterance(model, u)


def run(sync):
    S = torch.cuda.synchronize if sync else (lambda: None)
    marks = []
    mark = lambda name: marks.append((name, time.perf_counter()))
    torch.cuda.synchronize(); mark("start")
    with model.llm_context:
        gen = model.llm.inference(text=u["text"], text_len=t(B.N_TEXT), prompt_text=u["prompt_text"], prompt_text_len=t(B.N_PROMPT_TEXT), prompt_speech_token=u["llm_prompt_speech_token"],
                                  prompt_speech_token_len=t(B.N_PROMPT_TOK), embedding=u["llm_embedding"], max_token_text_ratio=ratio, min_token_text_ratio=ratio)
        tokens = [next(gen)]; mark("first token handed back")
        tokens += list(gen)
    S(); mark("last token")
    tok = torch.tensor(tokens).unsqueeze(0); mark("token tensor")
    uid = "probe"; model.hift_cache_dict[uid] = None
    with model._lane() as lane:
        mark("lane taken")
        mel, _ = lane.flow.inference(token=tok.to(torch.int32), token_len=t(tok.shape[1]), prompt_token=u["flow_prompt_speech_token"], prompt_token_len=t(u["flow_prompt_speech_token"].shape[1]),
                                     prompt_feat=u["prompt_speech_feat"], prompt_feat_len=t(u["prompt_speech_feat"].shape[1]), embedding=u["flow_embedding"], streaming=False, finalize=True)
        mark("flow.inference returned"); S(); mark("flow done on the device" if sync else "-")
        wav = model._t2w_tail(lane, mel, tok, 0, uid, True, 1.0)
        mark("HiFT enqueued"); S(); mark("HiFT done on the device" if sync else "-")
    out = wav.cpu(); mark("waveform on the host")
    model.hift_cache_dict.pop(uid, None)
    return marks


for sync in (True, False, False):
    m = run(sync)
    print("---- %s" % ("synchronised after every stage" if sync else "as the bench runs it"))
    for (a, ta), (b, tb) in zip(m, m[1:]):
        if b != "-":
            print("  %-28s +%8.3f ms   (%.3f)" % (b, 1e3 * (tb - ta), 1e3 * (tb - m[0][1])))


# ---- second part: the vocoder half alone (flow graph + HiFT), with device events, under three host behaviours
tok = torch.tensor(B.one_utterance.__globals__["N_GEN"] * [0]).unsqueeze(0)
keep = {}
B.one_utterance(model, u, keep)
tok = torch.tensor(keep["tokens"]).unsqueeze(0)


def half(mode):
    uid = "probe2"; model.hift_cache_dict[uid] = None
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with model._lane() as lane:
        st = torch.cuda.current_stream()
        e[0].record(st)
        mel, _ = lane.flow.inference(token=tok.to(torch.int32), token_len=t(tok.shape[1]), prompt_token=u["flow_prompt_speech_token"], prompt_token_len=t(u["flow_prompt_speech_token"].shape[1]),
                                     prompt_feat=u["prompt_speech_feat"], prompt_feat_len=t(u["prompt_speech_feat"].shape[1]), embedding=u["flow_embedding"], streaming=False, finalize=True)
        e[1].record(st)
        if mode == "sync":
            st.synchronize()
        elif mode == "event":
            e[1].synchronize()
        wav = model._t2w_tail(lane, mel, tok, 0, uid, True, 1.0)
        e[2].record(st)
    out = wav.cpu()
    wall = 1e3 * (time.perf_counter() - t0)
    model.hift_cache_dict.pop(uid, None)
    return wall, e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2])


for mode in ("none", "sync", "event", "none", "sync"):
    r = [half(mode) for _ in range(6)][1:]
    print("host between flow and HiFT: %-5s  wall %.2f ms | device: flow %.2f ms, HiFT %.2f ms" % (mode, sum(x[0] for x in r) / len(r), sum(x[1] for x in r) / len(r), sum(x[2] for x in r) / len(r)))
