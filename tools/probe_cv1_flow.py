"""CosyVoice-300M flow at its real dimensions on the MI355X: the U-Net estimator as one launch per operator (python), inside one library handle in fp32, and in bf16 mode
(cosyvoice_amd/cosyvoice1_hip.py EstimatorHandle, csrc/flow.hip cfg.estimator == 2).  Per variant: ms per flow.inference of a 500-token request (T = 861, 10 Euler steps;
passes 3+ replay the captured solve) and the mel against the operator sequence.

    gpurun -- python tools/probe_cv1_flow.py [T_tokens] [words selecting variants]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cosyvoice_amd import cosyvoice1_hip as CK, synthetic as W   # noqa: E402

cfg, _ = W.cv1()
sd = W.make_cv1_flow(cfg)
n_tok = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 500
only = [a for a in sys.argv[1:] if not a.isdigit()]          # e.g. `bf16` / `fp32` / `operators`: only the variants whose label contains every word
tl = lambda n: torch.tensor([n], dtype=torch.int32)
g = torch.Generator().manual_seed(300)
token = torch.randint(0, cfg.speech_token_size, (1, n_tok), generator=g, dtype=torch.int32)
ptok = torch.randint(0, cfg.speech_token_size, (1, 87), generator=g, dtype=torch.int32)
pfeat = torch.randn(1, 150, 80, generator=g) * 2 - 5
emb = torch.randn(1, cfg.spk_dim, generator=g)
kw = dict(token=token, token_len=tl(n_tok), prompt_token=ptok, prompt_token_len=tl(87), prompt_feat=pfeat, prompt_feat_len=tl(150), embedding=emb, flow_cache=torch.zeros(1, 80, 0, 2))
sync = torch.cuda.synchronize
ref = None
for est, prec, opts in (("operators", "fp32", {}), ("handle", "fp32", {}), ("handle", "bf16", {}), ("handle", "bf16", {"big_rows": 0}), ("handle", "bf16", {"big_rows": 1000}),
                        ("handle", "bf16", {"use_graph": 1}), ("handle", "fp32", {"use_graph": 1})):
    if only and not all(w in "%s %s %s" % (est, prec, opts) for w in only):
        continue
    flow = CK.MaskedDiffWithXvec(sd, enc_heads=cfg.flow_heads, est_heads=cfg.est_heads, input_frame_rate=cfg.input_frame_rate, estimator=est, precision=prec)
    for k, v in opts.items():
        flow.estimator.set_option(k, v)
    ms = []
    for rep in range(5):
        torch.manual_seed(7)
        sync(); t0 = time.perf_counter()
        mel, _ = flow.inference(**kw)
        sync(); ms.append(1e3 * (time.perf_counter() - t0))
    mel = mel.cpu()
    if ref is None:
        ref = mel
    rel = float((mel - ref).norm() / ref.norm())
    print("%-9s %-4s %-18s flow.inference ms: %s   mel [%d frames] rel-L2 vs operators %.3e, max abs %.3e" %
          (est, prec, opts or "", " ".join("%.1f" % m for m in ms), mel.shape[2], rel, float((mel - ref).abs().max())), flush=True)
    del flow
