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
