"""Dev probe, round 4: where a workgroup of the flow attention kernel (attn_flow_kernel<4,2,2>) spends its time - clock64() stamps of thread 0 at the phase
boundaries (flow option attn_dbg), mean over the workgroups of the LAST attention launch of a shared pass over nu utterances.   python tools/probe_flow_phases.py
Also the large-M GEMMs (flow option gemm_dbg = 1 QKV | 2 FF1 | 3 out-projection | 4 FF2) at 8 utterances per pass."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cosyvoice_amd import synthetic as W
from cosyvoice_amd.flow import CausalMaskedDiffWithXvec

lc, fc, hc = W.cv2()
u = W.synthetic_utterance(lc, fc)
flow = CausalMaskedDiffWithXvec(W.make_flow(fc), fc, precision="bf16")
g = torch.Generator().manual_seed(0)
tok = torch.randint(0, fc.vocab, (1, 250), generator=g, dtype=torch.int32)
item = dict(token=tok, prompt_token=u["flow_prompt_speech_token"], prompt_feat=u["prompt_speech_feat"], embedding=u["flow_embedding"])
opt = lambda k, v: flow.lib.cv_flow_set_option(flow._h, k.encode(), C.c_int32(v))


def stat(name):
    v = C.c_int64(0)
    flow.lib.cv_flow_get_stat(flow._h, name.encode(), C.byref(v))
    return v.value


opt("use_graph", 0)
for nu in (1, 8):
    for qg2 in (0, 1):
        opt("attn_dbg", 0); opt("attn2_rows", 1 if qg2 else 0)
        flow.inference_batch([item] * nu)
        opt("attn_dbg", 1)
        flow.inference_batch([item] * nu)
        torch.cuda.synchronize()
        ph = [stat("attn_phase_%d" % k) for k in range(4)]
        print("nu=%d QG=%d: %d workgroups | mean shader clocks: first tile parked %d, first tile multiplied %d, rest of the key loop %d, merge + store %d | sum %d | launch span %d, mean start offset %d"
              % (nu, 2 if qg2 else 1, stat("attn_phase_7"), ph[0], ph[1], ph[2], ph[3], sum(ph), stat("attn_phase_8"), stat("attn_phase_9")), flush=True)
opt("attn_dbg", 0); opt("attn2_rows", 0)
names = {1: "QKV (K 256, N 1536, bf16 + V^T out)", 2: "FF1 (K 256, N 1024, GELU, bf16 out)", 3: "out-projection (K 512, N 256, fp32 + residual)", 4: "FF2 (K 1024, N 256, fp32 + residual)"}
for epi in (1, 0):
    for tiles in ((3, 3), (2, 2), (1, 1)):
        opt("big_rows", 1); opt("big_tile0", tiles[0]); opt("big_tile1", tiles[1]); opt("big_lds_epi", epi)
        for which in (1, 2, 3, 4):
            opt("gemm_dbg", 0)
            flow.inference_batch([item] * 8)
            opt("gemm_dbg", which)
            flow.inference_batch([item] * 8)
            torch.cuda.synchronize()
            ph = [stat("gemm_phase_%d" % k) for k in range(3)]
            print("nu=8 tile %s lds_epilogue=%d %-48s %5d workgroups | mean clocks: first stage parked %6d, stages %6d, output %6d | sum %6d | launch span %7d, mean start offset %6d, mean end offset %6d"
                  % ({1: "128x128", 2: "128x64", 3: "64x64"}[tiles[which > 2]], epi, names[which], stat("gemm_phase_7"), ph[0], ph[1], ph[2], sum(ph), stat("gemm_phase_8"), stat("gemm_phase_9"),
                     stat("gemm_phase_10")), flush=True)
opt("gemm_dbg", 0); opt("big_rows", 5000); opt("big_tile0", 0); opt("big_tile1", 0); opt("big_lds_epi", 1); opt("use_graph", 1)
