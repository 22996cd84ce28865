"""Oracle: CosyVoice2Model pipeline glue (TEST INFRASTRUCTURE — see oracle/__init__.py).

Restates cosyvoice/cli/model.py:292-326 (token2wav), :328-394 (tts, without the threads: the chunk schedule only depends
on how many tokens exist) and cosyvoice/utils/common.py:170-178 (fade_in_out).  SineGen2 noise is an explicit argument
(zeros by default) so that the stochastic part of HiFT does not enter the comparison.
"""
import numpy as np
import torch

from . import flow as OF
from . import hift as OH
from . import llm as OL


def fade_in_out(fade_in_mel, fade_out_mel, window):
    """utils/common.py:170-178."""
    window = torch.from_numpy(np.asarray(window)).to(fade_in_mel.dtype) if not torch.is_tensor(window) else window
    n = int(window.shape[0] / 2)
    out = fade_in_mel.clone()
    out[..., :n] = fade_in_mel[..., :n] * window[:n] + fade_out_mel[..., -n:] * window[n:]
    return out


class Pipeline:
    def __init__(self, sds, cfgs, token_hop_len=25, n_timesteps=None):
        self.llm_sd, self.flow_sd, self.hift_sd = sds
        self.lc, self.fc, self.hc = cfgs
        self.token_hop_len = token_hop_len
        self.token_max_hop_len = 4 * token_hop_len
        self.stream_scale_factor = 2
        self.mel_cache_len = 8
        self.source_cache_len = self.mel_cache_len * 480
        self.speech_window = np.hamming(2 * self.source_cache_len).astype(np.float32)
        self.n_timesteps = n_timesteps

    def hift(self, mel, cache_source):
        L = mel.shape[2] * 480
        return OH.inference(self.hift_sd, self.hc, mel, cache_source, None, torch.zeros(1, L, self.hc.harmonics + 1))

    def token2wav(self, token, prompt_token, prompt_feat, embedding, token_offset, cache, stream=False, finalize=False, speed=1.0):
        mel = OF.inference(self.flow_sd, self.fc, token, prompt_token, prompt_feat, embedding, streaming=stream, finalize=finalize,
                           n_timesteps=self.n_timesteps)
        mel = mel[:, :, token_offset * 2:]
        if cache is not None:
            mel = torch.cat([cache["mel"], mel], dim=2)
            cs = cache["source"]
        else:
            cs = None
        if finalize and speed != 1.0:                                  # cli/model.py:320-322
            assert cache is None
            mel = torch.nn.functional.interpolate(mel, size=int(mel.shape[2] / speed), mode="linear")
        speech, source = self.hift(mel, cs)
        if cache is not None:
            speech = fade_in_out(speech, cache["speech"], self.speech_window)
        if not finalize:
            new_cache = {"mel": mel[:, :, -self.mel_cache_len:], "source": source[:, :, -self.source_cache_len:], "speech": speech[:, -self.source_cache_len:]}
            return speech[:, : -self.source_cache_len], new_cache
        return speech, cache

    def tts(self, tokens, u, stream=False, speed=1.0):
        """tokens: python list produced by the LLM; returns the list of yielded waveforms."""
        outs = []
        if not stream:
            sp, _ = self.token2wav(torch.tensor(tokens).unsqueeze(0), u["flow_prompt_speech_token"], u["prompt_speech_feat"], u["flow_embedding"], 0, None, False, True,
                                   speed=speed)
            return [sp]
        token_offset, hop, cache, la = 0, self.token_hop_len, None, self.fc.pre_lookahead
        n_p = u["flow_prompt_speech_token"].shape[1]
        pad = int(np.ceil(n_p / hop) * hop - n_p)
        while True:
            this_hop = hop + pad if token_offset == 0 else hop
            if len(tokens) - token_offset >= this_hop + la:
                sp, cache = self.token2wav(torch.tensor(tokens[: token_offset + this_hop + la]).unsqueeze(0), u["flow_prompt_speech_token"],
                                           u["prompt_speech_feat"], u["flow_embedding"], token_offset, cache, True, False)
                token_offset += this_hop
                hop = min(self.token_max_hop_len, hop * self.stream_scale_factor)
                outs.append(sp)
            else:
                break
        sp, _ = self.token2wav(torch.tensor(tokens).unsqueeze(0), u["flow_prompt_speech_token"], u["prompt_speech_feat"], u["flow_embedding"], token_offset, cache, False, True)
        outs.append(sp)
        return outs


class Pipeline3:
    """CosyVoice3Model glue (cli/model.py:397-450): token2wav with the accumulating mel cache and speech offsets over the DiT flow and the causal
    HiFT, the streaming loop of CosyVoice2Model.tts (inherited there), and the silent-token filter of llm_job (:101-129, :423)."""
    SILENT = [1, 2, 28, 29, 55, 248, 494, 2241, 2242, 2322, 2323]

    def __init__(self, sds, cfgs, token_hop_len=25):
        from . import dit as OD
        self.OD = OD
        self.llm_sd, self.flow_sd, self.hift_sd = sds
        self.lc, self.fc, self.hc = cfgs
        self.token_hop_len, self.token_max_hop_len, self.stream_scale_factor = token_hop_len, 4 * token_hop_len, 2

    @classmethod
    def filter_silent(cls, tokens, max_run=5):
        out, run = [], 0
        for t in tokens:
            if t in cls.SILENT:
                run += 1
                if run > max_run:
                    continue
            else:
                run = 0
            out.append(t)
        return out

    def token2wav(self, token, u, token_offset, cache, stream, finalize):
        mel = self.OD.inference(self.flow_sd, self.fc, token, u["flow_prompt_speech_token"], u["prompt_speech_feat"], u["flow_embedding"], streaming=stream,
                                finalize=finalize, n_timesteps=self.fc.n_timesteps)
        mel = mel[:, :, token_offset * 2:]
        if cache is not None:
            mel = torch.cat([cache["mel"], mel], dim=2)
            cache["mel"] = mel
        else:
            cache = {"mel": mel, "speech_offset": 0}
        speech, _ = OH.causal_inference(self.hift_sd, self.hc, mel, finalize, None, None, f0_dtype=torch.float32)
        speech = speech[:, cache["speech_offset"]:]
        cache["speech_offset"] += speech.shape[1]
        return speech, cache

    def tts(self, tokens, u, stream=False):
        tokens = self.filter_silent(tokens)
        if not stream:
            return [self.token2wav(torch.tensor(tokens).unsqueeze(0), u, 0, None, False, True)[0]]
        outs, token_offset, hop, cache, la = [], 0, self.token_hop_len, None, self.fc.pre_lookahead
        n_p = u["flow_prompt_speech_token"].shape[1]
        pad = int(np.ceil(n_p / hop) * hop - n_p)
        while True:
            this_hop = hop + pad if token_offset == 0 else hop
            if len(tokens) - token_offset < this_hop + la:
                break
            sp, cache = self.token2wav(torch.tensor(tokens[: token_offset + this_hop + la]).unsqueeze(0), u, token_offset, cache, True, False)
            token_offset += this_hop
            hop = min(self.token_max_hop_len, hop * self.stream_scale_factor)
            outs.append(sp)
        sp, _ = self.token2wav(torch.tensor(tokens).unsqueeze(0), u, token_offset, cache, False, True)
        outs.append(sp)
        return outs
