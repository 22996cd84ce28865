"""The whole benchmark request at the FULL model dimensions: the oracle pipeline against the REAL cosyvoice.cli.model.CosyVoice2Model.tts (a file of its own: pytest-xdist
hands out whole files, and this one test is 2.5 minutes of single-threaded oracle arithmetic).  See tests/test_fullsize_pinned.py for the per-stage pins."""
import os

import numpy as np
import torch

from cosyvoice_amd import synthetic as W

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
N_GEN, N_TEXT, N_PROMPT_TEXT, N_PROMPT_TOK = 250, 30, 12, 87


def load(name):
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(G, name + ".npz")).items()}


def test_whole_request_fullsize():
    """U10 end to end: oracle.model.Pipeline (what tests/test_zz_fullsize.py holds `CosyVoice2Model.tts` on the MI355X to) against the REAL
    cosyvoice.cli.model.CosyVoice2Model.tts around the real full-size flow + HiFT with its default streaming settings - offline (240 000 samples) and streamed (first
    chunk 32 640 samples = 25 + 13 + 3 tokens, then hops of 50 and 100 tokens and the rest): the same chunk lengths, the same waveform (every 8th sample stored;
    tolerance as at test dimensions: the harmonic phase integration amplifies fp32 round-off)."""
    from oracle import model as OM
    g = load("fullsize_model")
    cfgs = W.cv2()
    lc, fc, hc = cfgs
    u = W.synthetic_utterance(lc, fc, n_prompt_tok=N_PROMPT_TOK, n_prompt_text=N_PROMPT_TEXT, n_text=N_TEXT)
    tokens = load("fullsize_llm")["tokens"].tolist()
    pipe = OM.Pipeline((None, W.make_flow(fc), W.make_hift(hc)), cfgs)
    with torch.inference_mode():
        # the streamed half is five flow passes + five vocoder calls at full size (2 minutes on one thread): every run checks its chunk SCHEDULE against the real
        # class's, the offline waveform AND (round 6: no longer opt-in; CV_TEST_SKIP_FULL=1 leaves it out) the streamed waveform
        n_p = u["flow_prompt_speech_token"].shape[1]
        sched, off, hop, la = [], 0, pipe.token_hop_len, fc.pre_lookahead
        pad = int(np.ceil(n_p / hop) * hop - n_p)
        while len(tokens) - off >= (hop + pad if off == 0 else hop) + la:
            this = hop + pad if off == 0 else hop
            sched.append(480 * 2 * this - (480 * 8 if off == 0 else 0))          # the first chunk withholds the 8-frame vocoder cache
            off += this
            hop = min(pipe.token_max_hop_len, hop * pipe.stream_scale_factor)
        sched.append(480 * 2 * len(tokens) - sum(sched))
        assert sched == g["stream_n"].tolist()
        for key, stream in (("offline", False), ("stream", True)):
            if stream and os.environ.get("CV_TEST_SKIP_FULL"):
                continue
            outs = pipe.tts(tokens, u, stream=stream)
            assert [o.shape[1] for o in outs] == g[key + "_n"].tolist()
            wav = torch.cat(outs, 1)
            assert wav.shape[1] == 2 * 480 * N_GEN
            torch.testing.assert_close(wav[:, ::8], g[key], rtol=0, atol=5e-3)
    assert g["stream_n"].tolist() == [32640, 48000, 96000, 63360] and float(g["offline"].abs().max()) > 0.05


def test_cv3_whole_request_fullsize():
    """bench.py's cosyvoice3 request end to end at Fun-CosyVoice3-0.5B dimensions: oracle.model.Pipeline3 against the REAL cli.model.CosyVoice3Model.tts (silent-token filter,
    accumulating mel cache, speech offsets) around the real DiT flow with its 10 Euler steps + CausalHiFTGenerator, offline (240 000 samples, every 8th stored).  Ten
    full-size DiT passes take a few minutes of CPU: part of the default suite since round 6 (profiles/r6_cv_test_full.log; CV_TEST_SKIP_FULL=1 leaves it out)."""
    import pytest
    if os.environ.get("CV_TEST_SKIP_FULL"):
        pytest.skip("CV_TEST_SKIP_FULL=1: the full-size CosyVoice3 request through the oracle is left out")
    from cosyvoice_amd import configs as CF
    from oracle import model as OM
    g = load("fullsize_model_cv3")
    lc, fc, hc = CF.cv3_llm(), CF.cv3_flow(), CF.cv3_hift()
    u = W.synthetic_utterance(lc, fc, n_prompt_tok=N_PROMPT_TOK, n_prompt_text=24, n_text=N_TEXT, seed=2025)
    tokens = load("fullsize_llm_cv3")["tokens"].tolist()
    pipe = OM.Pipeline3((None, W.make_flow_dit(fc), W.make_hift(hc)), (lc, fc, hc))
    with torch.inference_mode():
        outs = pipe.tts(tokens, u, stream=False)
    assert [o.shape[1] for o in outs] == g["offline_n"].tolist() == [2 * 480 * N_GEN]
    torch.testing.assert_close(torch.cat(outs, 1)[:, ::8], g["offline"], rtol=0, atol=5e-3)
    assert float(g["offline"].abs().max()) > 0.05
