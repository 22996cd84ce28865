"""Dev tool (GPU box): wall time of one flow.inference at the U10 size (337 tokens -> T = 674 frames, 10 Euler steps, bf16 mode) for the
knobs of the fused estimator pipeline (flow_fused.h): fused on / off, LN-GEMM tile, attention workgroup size."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cosyvoice_amd import synthetic as W
from cosyvoice_amd.flow import CausalMaskedDiffWithXvec

lc, fc, hc = W.cv2()
u = W.synthetic_utterance(lc, fc)
t = lambda n: torch.tensor([n], dtype=torch.int32)
flow = CausalMaskedDiffWithXvec(W.make_flow(fc), fc, precision="bf16")
tok = torch.randint(0, fc.vocab, (1, 250), generator=torch.Generator().manual_seed(0), dtype=torch.int32)


def run():
    mel, _ = flow.inference(token=tok, token_len=t(250), prompt_token=u["flow_prompt_speech_token"], prompt_token_len=t(87), prompt_feat=u["prompt_speech_feat"],
                            prompt_feat_len=t(174), embedding=u["flow_embedding"], streaming=False, finalize=True)
    return mel


def bench(label, reps=8):
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        mel = run()
    torch.cuda.synchronize()
    print("%-40s %7.2f ms per flow.inference" % (label, (time.perf_counter() - t0) / reps * 1e3), flush=True)
    return mel


which = sys.argv[1:] or ["all"]
opt = lambda k, v: flow.lib.cv_flow_set_option(flow._h, k, C.c_int32(v))
if "all" in which:
    opt(b"fused", 0); ref = bench("unfused (round-1 path)")
    opt(b"fused", 1)
    for tile, waves, kt, ks in ((0, 4, 1, 1), (0, 4, 2, 2), (0, 4, 2, 1)):
        opt(b"flow_tile", tile); opt(b"attn_waves", waves); opt(b"attn_kt", kt); opt(b"attn_ks", ks)
        mel = bench("fused tile=%d attn_waves=%d attn_kt=%d attn_ks=%d" % (tile, waves, kt, ks))
        print("    max |fused - unfused| = %.3e (mel std %.2f)" % ((mel - ref).abs().max().item(), ref.std().item()), flush=True)
elif "streams" in which:             # round 3: the estimator's batch rows as one launch chain or as two chains on two streams (est_streams)
    opt(b"fused", 1); opt(b"fused_tail", 0)
    ref = None
    for streams in (1, 2, 1, 2):
        opt(b"est_streams", streams)
        mel = bench("estimator batch rows on %d stream(s)" % streams)
        if ref is None:
            ref = mel
        else:
            print("    bit-identical to the first run: %s" % torch.equal(mel, ref), flush=True)
    # first-chunk-sized streaming request (87 prompt + 41 tokens)
    tok1 = tok[:, :41]
    def run1():
        mel, _ = flow.inference(token=tok1, token_len=t(41), prompt_token=u["flow_prompt_speech_token"], prompt_token_len=t(87), prompt_feat=u["prompt_speech_feat"],
                                prompt_feat_len=t(174), embedding=u["flow_embedding"], streaming=True, finalize=False)
        return mel
    for streams in (1, 2):
        opt(b"est_streams", streams)
        for _ in range(3):
            run1()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(8):
            run1()
        torch.cuda.synchronize()
        print("first streaming chunk (128 tokens)  %d stream(s)  %7.2f ms per flow.inference" % (streams, (time.perf_counter() - t0) / 8 * 1e3), flush=True)
    opt(b"est_streams", 2)
elif "attn" in which:                # round 3: key splits inside the 64-query attention workgroup (8 / 12 / 16 waves)
    opt(b"fused", 1); opt(b"fused_tail", 0)
    ref = None
    for ks, kt in ((2, 1), (3, 1), (4, 1), (1, 2)):
        opt(b"attn_ks", ks); opt(b"attn_kt", kt)
        mel = bench("attention key splits %d (kt %d)" % (ks, kt))
        if ref is None:
            ref = mel
        else:
            print("    max |this - ks 2| = %.3e (mel std %.2f)" % ((mel - ref).abs().max().item(), ref.std().item()), flush=True)
    opt(b"attn_ks", 2); opt(b"attn_kt", 1)
elif "ntile" in which:               # round 3: one or two N tiles per workgroup of the LayerNorm-prologue GEMMs (flow_gemm_kernel<.., NTILE>)
    opt(b"fused", 1); opt(b"fused_tail", 0)
    opt(b"flow_ntile", 1); ref = bench("one N tile per workgroup (round 2)")
    for nt, label in ((2, "two N tiles per workgroup, QKV and FF1"), (0, "two where that makes one round (QKV only at T = 674)"), (1, "one again")):
        opt(b"flow_ntile", nt)
        mel = bench(label)
        print("    bit-identical %s" % bool(torch.equal(mel, ref)), flush=True)
    opt(b"flow_ntile", 0)
elif "tail" in which:                # round 3: the one-launch block tail (flow_tail.h) against the five-launch block, ring depth 8 / 16
    opt(b"fused", 1); opt(b"fused_tail", 0); ref = bench("five launches per block (round 2)")
    for ring in (8, 16):
        opt(b"fused_tail", 1); opt(b"tail_ring", ring)
        mel = bench("attention + ONE tail launch per block, ring %d" % ring)
        print("    max |tail - five-launch| = %.3e, bit-identical %s (mel std %.2f)" % ((mel - ref).abs().max().item(), bool(torch.equal(mel, ref)), ref.std().item()), flush=True)
else:
    opt(b"fused", int(os.environ.get("FUSED", "1"))); opt(b"flow_tile", int(os.environ.get("TILE", "0"))); opt(b"attn_waves", int(os.environ.get("WAVES", "4"))); opt(b"attn_kt", int(os.environ.get("KT", "1"))); opt(b"attn_ks", int(os.environ.get("KS", "2")))
    opt(b"fused_tail", int(os.environ.get("TAIL", "1")))
    opt(b"use_graph", 0)
    bench("profile run", reps=1)
