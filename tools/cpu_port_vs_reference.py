"""Dev tool, BUILD CONTAINER ONLY (imports /root/reference through tests/golden/make_golden*.py): bench.py's `cpu_baseline` times the oracle - a CPU PORT of the
reference (`kind: port`).  This script times the REAL reference modules beside the port on the same stage samples `bench.py --cpu-stage` uses, in one fresh process
per stage, same thread count, so that the port's CPU numbers have a stated ratio to the reference's own code.

    python tools/cpu_port_vs_reference.py            -> profiles/r6_cpu_port_vs_reference.txt

  llm    cosyvoice.llm.llm.Qwen2LM's Qwen2Encoder.forward_one_step (transformers Qwen2ForCausalLM, KV cache) + llm_decoder + log_softmax per token at contexts
         ~131 / ~256 / ~381, vs oracle.llm.Qwen2Oracle.forward + the same head
  flow   the real UpsampleConformerEncoder on 337 tokens and the real CausalConditionalDecoder estimator (over the restated Matcha blocks, matcha_stub.py) on one CFG
         pair at T = 674, vs oracle.flow.encoder / oracle.flow.estimator
  hift   the real HiFTGenerator.inference on 100 frames vs oracle.hift.inference
"""
import json
import os
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
N_GEN, N_TEXT, N_PROMPT_TEXT, N_PROMPT_TOK = 250, 30, 12, 87
THREADS = int(os.environ.get("CV_CPU_THREADS", min(8, os.cpu_count() or 8)))


def best(fn, reps=3):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return min(ts)


def stage(name):
    _argv, sys.argv = sys.argv, sys.argv[:1]
    import make_golden as MG          # installs the reference import stubs
    import make_golden_fullsize as MF
    sys.argv = _argv
    from cosyvoice_amd import synthetic as W
    from oracle import flow as OF, hift as OH, llm as OL
    torch.set_num_threads(THREADS)
    lc, fc, hc = W.cv2()
    u = W.synthetic_utterance(lc, fc, n_prompt_tok=N_PROMPT_TOK, n_prompt_text=N_PROMPT_TEXT, n_text=N_TEXT)
    res = {}
    with torch.inference_mode():
        if name == "llm":
            sd = W.make_llm(lc)
            lm, _ = MF._qwen_lm(lc, "Qwen2LM")
            x = OL.build_lm_input(sd, lc, u["text"], u["prompt_text"], u["llm_prompt_speech_token"])
            tok = sd["speech_embedding.weight"][5].reshape(1, -1)
            fill = sd["speech_embedding.weight"][torch.arange(7, 7 + 121)]
            # port
            m = OL.Qwen2Oracle(sd, lc)
            m.forward(x); m.forward(tok)
            per = []
            for hop in range(3):
                if hop:
                    m.forward(fill)
                per.append(best(lambda: torch.nn.functional.linear(m.forward(tok)[-1], sd["llm_decoder.weight"], sd["llm_decoder.bias"]).log_softmax(-1), 4))
            res["port_llm_per_token_ms"] = 1e3 * sum(per) / len(per)
            # real class: the statements of Qwen2LM.inference_wrapper's loop body (llm/llm.py:535-549) around the real forward_one_step
            xs = x.unsqueeze(0)
            y, cache = lm.llm.forward_one_step(xs, masks=torch.tril(torch.ones((1, xs.shape[1], xs.shape[1]))).to(torch.bool), cache=None)
            one = tok.unsqueeze(0)
            y, cache = lm.llm.forward_one_step(one, masks=torch.ones((1, 1, 1), dtype=torch.bool), cache=cache)
            per = []
            for hop in range(3):
                if hop:
                    L = fill.shape[0]
                    y, cache = lm.llm.forward_one_step(fill.unsqueeze(0), masks=torch.tril(torch.ones((1, L, L))).to(torch.bool), cache=cache)

                def step():
                    nonlocal cache
                    y, cache = lm.llm.forward_one_step(one, masks=torch.ones((1, 1, 1), dtype=torch.bool), cache=cache)
                    lm.llm_decoder(y[:, -1]).log_softmax(dim=-1)
                per.append(best(step, 4))
            res["reference_llm_per_token_ms"] = 1e3 * sum(per) / len(per)
        elif name == "flow":
            fsd = W.make_flow(fc)
            flow = MG.build_ref_flow(fc)
            T = 2 * (N_PROMPT_TOK + N_GEN)
            g = torch.Generator().manual_seed(0)
            emb = torch.randn(1, N_PROMPT_TOK + N_GEN, fc.dim, generator=g)
            xx = torch.randn(2, 80, T, generator=g); spk = torch.randn(2, 80, generator=g)
            tt, mask = torch.tensor([0.3, 0.3]), torch.ones(2, 1, T)
            n = torch.tensor([N_PROMPT_TOK + N_GEN], dtype=torch.int32)
            res["port_encoder_ms"] = 1e3 * best(lambda: OF.encoder(fsd, fc, emb, None, False))
            res["reference_encoder_ms"] = 1e3 * best(lambda: flow.encoder(emb, n, streaming=False))
            res["port_estimator_step_ms"] = 1e3 * best(lambda: OF.estimator(fsd, fc, xx, mask, xx, tt, spk, xx, False), 2)
            res["reference_estimator_step_ms"] = 1e3 * best(lambda: flow.decoder.estimator(xx, mask, xx, tt, spk, xx, streaming=False), 2)
        elif name == "hift":
            hsd = W.make_hift(hc)
            hift = MG.build_ref_hift(hc)
            mel = torch.randn(1, 80, 100, generator=torch.Generator().manual_seed(0)) * 2 - 5
            OH.inference(hsd, hc, mel[:, :, :20]); hift.inference(speech_feat=mel[:, :, :20])
            res["port_hift_100_frames_ms"] = 1e3 * best(lambda: OH.inference(hsd, hc, mel))
            res["reference_hift_100_frames_ms"] = 1e3 * best(lambda: hift.inference(speech_feat=mel))
    print(json.dumps(res), flush=True)


def main():
    if len(sys.argv) > 1:
        return stage(sys.argv[1])
    rows = {}
    for name in ("llm", "flow", "hift"):
        p = subprocess.run([sys.executable, os.path.abspath(__file__), name], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=3600)
        if p.returncode != 0:
            raise SystemExit("stage %s failed:\n%s" % (name, p.stderr[-3000:]))
        rows.update(json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1]))
    lines = ["# The CPU oracle (`cpu_baseline.kind: port`, oracle/*.py) beside the REAL reference modules on bench.py's CPU stage samples.",
             "# build container, %d torch threads of %d cores, torch %s, fp32, one fresh process per stage, best of a few repetitions (tools/cpu_port_vs_reference.py)" % (THREADS, os.cpu_count() or 0, torch.__version__),
             "# stage sample                                              port (oracle)   reference class   port / reference"]
    for label, key in (("LLM decode step, contexts ~131 / 256 / 381 (ms per token)", "llm_per_token_ms"), ("flow encoder, 337 tokens (ms)", "encoder_ms"),
                       ("flow estimator, one CFG pair at T = 674 (ms)", "estimator_step_ms"), ("HiFT, 100 frames (ms)", "hift_100_frames_ms")):
        a, b = rows["port_" + key], rows["reference_" + key]
        lines.append("%-60s %10.1f %16.1f %14.2f" % (label, a, b, a / b))
    lines.append("# reading: the port restates the reference's arithmetic as flat state-dict functions over the same torch operators, so the two run within tens of per cent of")
    lines.append("# each other; the reference's Qwen2 step goes through transformers' Qwen2ForCausalLM (module dispatch, DynamicCache, lm_head over the 151 936-row text vocabulary")
    lines.append("# computed and discarded per token - SURVEY.md Appendix C.3), which the port skips.")
    out = os.path.join(ROOT, "profiles", "r6_cpu_port_vs_reference.txt")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
