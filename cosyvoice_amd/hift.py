"""Host mirror of cosyvoice.hifigan.generator.HiFTGenerator for inference (boundary B6).

`inference(speech_feat[1,80,m], cache_source[1,1,c]) -> (speech[1,480m], source[1,1,480m])` like the reference
(hifigan/generator.py:557-569); f0 prediction, the harmonic source, the conv stack and the iSTFT all run behind cv_hift_*.
"""
import ctypes as C

import torch

from . import weights as Wt
from ._lib import get_lib, stream_ptr
from .llm import register_tensors


class HiftConfigC(C.Structure):
    _fields_ = [("mel", C.c_int32), ("base", C.c_int32), ("harmonics", C.c_int32), ("sr", C.c_int32),
                ("n_ups", C.c_int32), ("ups", C.c_int32 * 4), ("up_k", C.c_int32 * 4),
                ("n_res", C.c_int32), ("res_k", C.c_int32 * 4), ("src_k", C.c_int32 * 4),
                ("n_dil", C.c_int32), ("dil", C.c_int32 * 4),
                ("n_fft", C.c_int32), ("hop", C.c_int32), ("f0_ch", C.c_int32),
                ("nsf_alpha", C.c_float), ("nsf_sigma", C.c_float), ("voiced_thr", C.c_float), ("lrelu", C.c_float), ("audio_limit", C.c_float),
                ("causal", C.c_int32), ("look_right", C.c_int32)]


def _arr4(v):
    return (C.c_int32 * 4)(*(list(v) + [0] * (4 - len(v))))


class _F0Predictor:
    def __init__(self, hift):
        self.hift = hift

    def __call__(self, speech_feat):
        h = self.hift
        m = speech_feat.shape[2]
        x = h.lib.hook(speech_feat.to(h.device, torch.float32).contiguous())
        out = h.lib.hook(torch.empty(1, m, dtype=torch.float32, device=h.device))
        h.lib.cv_hift_f0(h._h, C.c_void_p(x.data_ptr()), C.c_int32(m), C.c_void_p(out.data_ptr()), stream_ptr(h.lib))
        return out


class HiFTGenerator:
    def __init__(self, state_dict, cfg, lib=None, seed=1986, _tensors=None, f0_float64=False, terms=6):
        """f0_float64: the f0 predictor with every sum in double (cv_hift_set_option "f0_float64") - the mode the reference runs the causal generator's
        predictor in (generator.py:716-717); default off: fp32 on the exact-fp32 matrix pipe (bounds against float64: tests/test_zz_fullsize.py).
        terms: plane products per k of the decoder's convolutions (cv_hift_set_option "terms"): 6 = the fp32-exact class (default), 3 = 16 significand bits per
        factor with fp32 accumulation - the reduced-precision mode CosyVoice3Model(fp16=True) selects where the reference runs its vocoder under
        autocast (cli/model.py:426-447); the f0 predictor and the source are not affected."""
        self.lib = lib or get_lib()
        self.cfg = cfg
        self.device = torch.device(self.lib.device)
        self.sampling_rate = cfg.sr
        self.seed = seed
        self._calls = 0
        self._next_seed = None                 # set by CosyVoice2Model.token2wav: RNG key of the next inference() call (see inference)
        self.upsample_scale = cfg.hop
        for u in cfg.ups:
            self.upsample_scale *= u
        self._tensors = _tensors if _tensors is not None else {k: self.lib.hook(v) for k, v in Wt.pack_hift(state_dict, cfg, self.device).items()}
        c = HiftConfigC(cfg.mel, cfg.base, cfg.harmonics, cfg.sr, len(cfg.ups), _arr4(cfg.ups), _arr4(cfg.up_k), len(cfg.res_k), _arr4(cfg.res_k),
                        _arr4(cfg.src_k), len(cfg.res_d), _arr4(cfg.res_d), cfg.n_fft, cfg.hop, cfg.f0_ch, cfg.nsf_alpha, cfg.nsf_sigma,
                        cfg.voiced_thr, cfg.lrelu, cfg.audio_limit, int(cfg.causal), cfg.look_right)
        self._h = C.c_void_p()
        self.lib.cv_hift_create(C.byref(self._h), C.byref(c))
        register_tensors(self.lib, "cv_hift_set_tensor", self._h, self._tensors)
        self.lib.cv_hift_finalize(self._h)
        self.f0_float64 = bool(f0_float64)
        if self.f0_float64:
            self.lib.cv_hift_set_option(self._h, b"f0_float64", C.c_int32(1))
        self.terms = int(terms)
        if self.terms != 6:
            self.lib.cv_hift_set_option(self._h, b"terms", C.c_int32(self.terms))
        self.f0_predictor = _F0Predictor(self)

    def clone(self):
        """Same device weights, own library handle (workspaces): one per token2wav lane of CosyVoice2Model."""
        return type(self)(None, self.cfg, lib=self.lib, seed=self.seed, _tensors=self._tensors, f0_float64=self.f0_float64, terms=self.terms)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self.lib.raw("cv_hift_destroy", None)(self._h)
                self._h = None
        except Exception:
            pass

    @torch.inference_mode()
    def decode(self, x, s):
        """HiFTGenerator.decode(x[1,80,m], s[1,1,480m]) -> [1,480m]   (generator.py:507-539)."""
        m = x.shape[2]
        xx = self.lib.hook(x.to(self.device, torch.float32).contiguous())
        ss = self.lib.hook(s.to(self.device, torch.float32).contiguous())
        assert ss.numel() == m * self.upsample_scale
        out = self.lib.hook(torch.empty(1, m * self.upsample_scale, dtype=torch.float32, device=self.device))
        self.lib.cv_hift_decode(self._h, C.c_void_p(xx.data_ptr()), C.c_int32(m), C.c_void_p(ss.data_ptr()), C.c_void_p(out.data_ptr()), stream_ptr(self.lib))
        return out

    @torch.inference_mode()
    def inference(self, speech_feat, cache_source=None, noise=None, seed=None):
        """-> (generated_speech[1,480m], source[1,1,480m]).  `noise` ([480m,9] N(0,1)) is a parity hook; by default the SineGen2
        noise comes from an in-kernel counter RNG (the reference consumes the global device RNG, generator.py:312).  `seed`: the counter
        RNG's key for this call (CosyVoice2Model.token2wav derives it from the request's tokens, so a request's audio does not depend on
        the lane / rank / order it was served in); default: model seed + call count."""
        m = speech_feat.shape[2]
        L = m * self.upsample_scale
        x = self.lib.hook(speech_feat.to(self.device, torch.float32).contiguous())
        speech = self.lib.hook(torch.empty(1, L, dtype=torch.float32, device=self.device))
        source = self.lib.hook(torch.empty(1, 1, L, dtype=torch.float32, device=self.device))
        cs, cl = None, 0
        if cache_source is not None and cache_source.shape[2] != 0:
            cs = self.lib.hook(cache_source.to(self.device, torch.float32).contiguous())
            cl = cs.shape[2]
        nz = None if noise is None else self.lib.hook(noise.to(self.device, torch.float32).reshape(L, -1).contiguous())
        if seed is None:
            seed, self._next_seed = self._next_seed, None
        self._calls += 1
        self.lib.cv_hift_inference(self._h, C.c_void_p(x.data_ptr()), C.c_int32(m), C.c_void_p(cs.data_ptr()) if cs is not None else None, C.c_int32(cl),
                                   C.c_void_p(nz.data_ptr()) if nz is not None else None, C.c_uint64((self.seed + self._calls) if seed is None else int(seed)),
                                   C.c_void_p(speech.data_ptr()), C.c_void_p(source.data_ptr()), stream_ptr(self.lib))
        return speech, source


    @torch.inference_mode()
    def inference_batch(self, speech_feats, seeds, noise=None):
        """n one-shot utterances of EQUAL length through ONE launch sequence (cv_hift_inference_batch; the reference's high-throughput runtime batches
        token2wav, runtime/triton_trtllm/token2wav.py:20-22): speech_feats [n, 80, m], seeds: n counter-RNG keys (what `inference(seed=...)` takes per
        utterance), noise: optional [n, 480 m, 9] parity hook -> (speech [n, 480 m], source [n, 1, 480 m]); row i is bit-identical to
        `inference(speech_feats[i:i+1], seed=seeds[i])`."""
        n, _, m = speech_feats.shape
        assert 1 <= n <= 16 and len(seeds) == n
        L = m * self.upsample_scale
        x = self.lib.hook(speech_feats.to(self.device, torch.float32).contiguous())
        speech = self.lib.hook(torch.empty(n, L, dtype=torch.float32, device=self.device))
        source = self.lib.hook(torch.empty(n, 1, L, dtype=torch.float32, device=self.device))
        nz = None if noise is None else self.lib.hook(noise.to(self.device, torch.float32).reshape(n, L, -1).contiguous())
        keys = (C.c_uint64 * n)(*[int(k) for k in seeds])
        self._calls += n
        self.lib.cv_hift_inference_batch(self._h, C.c_int32(n), C.c_void_p(x.data_ptr()), C.c_int32(m), C.c_void_p(nz.data_ptr()) if nz is not None else None, keys,
                                         C.c_void_p(speech.data_ptr()), C.c_void_p(source.data_ptr()), stream_ptr(self.lib))
        return speech, source


class CausalHiFTGenerator(HiFTGenerator):
    """cosyvoice.hifigan.generator.CausalHiFTGenerator for inference (generator.py:572-726; Fun-CosyVoice3):
    `inference(speech_feat[1,80,m], finalize=True) -> (speech, source)`.  A non-final chunk (finalize=False) treats its last frames as look-ahead
    only: 3 for the f0 predictor, 4 more for conv_pre, and withholds one more frame of samples, so m frames give 480 (m - 8) samples - every
    sample a chunk emits equals the one-shot result (the reference's own invariance check, generator.py:729-746).

    The f0 predictor runs with every sum in double by default, like the reference (which converts the predictor to float64 on every call, :716-717, because its
    fp32 cuDNN results depend on the chunk): `f0_float64=True`, the golden f0 of the real class is met to fp32 rounding.  `f0_float64=False` selects fp32 sums
    (each f0 value the same sum whatever the chunk, so chunked = one-shot bit for bit there too; against the float64 reference the harmonic phase then agrees
    to 2e-3 like HiFT v2; measured cost of the default on the MI355X: INTEGRATION.md section 4).  Stated difference: the SineGen2 noise comes from a counter RNG
    (uniform, like the reference's fixed `torch.rand` buffer - 260 MB there) unless `noise` is given."""

    def __init__(self, state_dict, cfg, **kw):
        assert cfg.causal, "CausalHiFTGenerator needs a causal HiftConfig (configs.cv3_hift())"
        kw.setdefault("f0_float64", True)       # the reference's arithmetic for this predictor (generator.py:716-717) is the default since round 4; False = the fp32 sums
        super().__init__(state_dict, cfg, **kw)
        self.conv_pre_look_right = cfg.look_right

    @torch.inference_mode()
    def f0(self, speech_feat, finalize=True):
        m = speech_feat.shape[2]
        x = self.lib.hook(speech_feat.to(self.device, torch.float32).contiguous())
        out = self.lib.hook(torch.empty(1, m, dtype=torch.float32, device=self.device))
        n = C.c_int32(0)
        self.lib.cv_hift_causal_f0(self._h, C.c_void_p(x.data_ptr()), C.c_int32(m), C.c_int32(int(finalize)), C.c_void_p(out.data_ptr()), C.byref(n), stream_ptr(self.lib))
        return out[:, : n.value]

    @torch.inference_mode()
    def decode(self, x, s, finalize=True):
        m = x.shape[2]
        xx = self.lib.hook(x.to(self.device, torch.float32).contiguous())
        ss = self.lib.hook(s.to(self.device, torch.float32).contiguous())
        assert ss.numel() == m * self.upsample_scale
        out = self.lib.hook(torch.empty(1, m * self.upsample_scale, dtype=torch.float32, device=self.device))
        n = C.c_int64(0)
        self.lib.cv_hift_causal_decode(self._h, C.c_void_p(xx.data_ptr()), C.c_int32(m), C.c_void_p(ss.data_ptr()), C.c_int32(int(finalize)),
                                       C.c_void_p(out.data_ptr()), C.byref(n), stream_ptr(self.lib))
        return out[:, : n.value]

    @torch.inference_mode()
    def inference(self, speech_feat, finalize=True, noise=None):
        m = speech_feat.shape[2]
        L = m * self.upsample_scale
        x = self.lib.hook(speech_feat.to(self.device, torch.float32).contiguous())
        speech = self.lib.hook(torch.empty(1, L, dtype=torch.float32, device=self.device))
        source = self.lib.hook(torch.empty(1, 1, L, dtype=torch.float32, device=self.device))
        nz = None
        if noise is not None:
            nz = self.lib.hook(noise.to(self.device, torch.float32).reshape(-1, self.cfg.harmonics + 1).contiguous())
            assert nz.shape[0] >= L - (0 if finalize else 3 * self.upsample_scale)
        self._calls += 1
        n_sp, n_src = C.c_int64(0), C.c_int64(0)
        self.lib.cv_hift_causal_inference(self._h, C.c_void_p(x.data_ptr()), C.c_int32(m), C.c_int32(int(finalize)),
                                          C.c_void_p(nz.data_ptr()) if nz is not None else None, C.c_uint64(self.seed),      # one fixed "buffer" per model, like the reference
                                          C.c_void_p(speech.data_ptr()), C.byref(n_sp), C.c_void_p(source.data_ptr()), C.byref(n_src), stream_ptr(self.lib))
        return speech[:, : n_sp.value], source[:, :, : n_src.value]
