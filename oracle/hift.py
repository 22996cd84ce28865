"""Oracle: HiFT neural-source-filter + iSTFT vocoder (TEST INFRASTRUCTURE — see oracle/__init__.py).

Restates, in plain torch fp32 over a reference-named state dict (weight-norm folded exactly as torch's parametrization
does, w = g * v / ||v||):
  cosyvoice/hifigan/f0_predictor.py:23-59        ConvRNNF0Predictor
  cosyvoice/hifigan/generator.py:192-317         SineGen2 (non-causal: linear-interp phase trick)
  cosyvoice/hifigan/generator.py:320-375         SourceModuleHnNSF
  cosyvoice/hifigan/generator.py:46-122          ResBlock (+ cosyvoice/transformer/activation.py:73-84 Snake)
  cosyvoice/hifigan/generator.py:491-539         _stft / _istft / decode
  cosyvoice/hifigan/generator.py:557-569         inference
The stochastic inputs of SineGen2/SourceModule (torch.rand / randn_like on the global RNG) are explicit arguments here.
"""
import numpy as np
import torch
import torch.nn.functional as F


def fold_weight_norm(sd, p):
    """weight = original0 * original1 / ||original1||_(all dims but 0)   (torch.nn.utils.parametrizations.weight_norm)."""
    g, v = sd[p + "parametrizations.weight.original0"], sd[p + "parametrizations.weight.original1"]
    return g * v / v.reshape(v.shape[0], -1).norm(dim=1).reshape(-1, 1, 1)


def hann16():
    n = torch.arange(16, dtype=torch.float64)
    return (0.5 - 0.5 * torch.cos(2 * np.pi * n / 16)).float()          # scipy get_window("hann", 16, fftbins=True)


def snake(x, alpha):
    a = alpha[None, :, None]
    return x + (1.0 / (a + 1e-9)) * torch.sin(x * a) ** 2


def f0_predictor(sd, mel):
    """mel [1,80,m] -> f0 [1,m]."""
    x = mel
    for j in range(5):
        p = "f0_predictor.condnet.%d." % (2 * j)
        x = F.elu(F.conv1d(x, fold_weight_norm(sd, p), sd[p + "bias"], padding=1))
    return torch.abs(F.linear(x.transpose(1, 2), sd["f0_predictor.classifier.weight"], sd["f0_predictor.classifier.bias"]).squeeze(-1))


def sine_gen2(cfg, f0, rand_ini, noise):
    """SineGen2.forward: f0 [1,L,1] (already upsampled), rand_ini [1,9] (col 0 = 0), noise [1,L,9] ~ N(0,1)."""
    H = cfg.harmonics + 1
    scale = int(np.prod(cfg.ups) * cfg.hop)
    fn = f0 * torch.arange(1, H + 1, dtype=torch.float32).reshape(1, 1, H)
    rad = (fn / cfg.sr) % 1
    rad[:, 0, :] = rad[:, 0, :] + rand_ini
    rad = F.interpolate(rad.transpose(1, 2), scale_factor=1 / scale, mode="linear").transpose(1, 2)
    phase = torch.cumsum(rad, dim=1) * 2 * np.pi
    phase = F.interpolate(phase.transpose(1, 2) * scale, scale_factor=scale, mode="linear").transpose(1, 2)
    sines = torch.sin(phase) * cfg.nsf_alpha
    uv = (f0 > cfg.voiced_thr).float()
    noise_amp = uv * cfg.nsf_sigma + (1 - uv) * cfg.nsf_alpha / 3
    return sines * uv + noise_amp * noise, uv


def source_module(sd, cfg, f0_up, rand_ini, noise):
    """SourceModuleHnNSF.forward -> sine_merge [1,L,1]   (noise branch output is unused by HiFT.inference)."""
    sine_wavs, _ = sine_gen2(cfg, f0_up, rand_ini, noise)
    return torch.tanh(F.linear(sine_wavs, sd["m_source.l_linear.weight"], sd["m_source.l_linear.bias"]))


def resblock(sd, cfg, p, x, k):
    for j, d in enumerate(cfg.res_d):
        xt = snake(x, sd[p + "activations1.%d.alpha" % j])
        xt = F.conv1d(xt, fold_weight_norm(sd, p + "convs1.%d." % j), sd[p + "convs1.%d.bias" % j], dilation=d, padding=(k * d - d) // 2)
        xt = snake(xt, sd[p + "activations2.%d.alpha" % j])
        xt = F.conv1d(xt, fold_weight_norm(sd, p + "convs2.%d." % j), sd[p + "convs2.%d.bias" % j], padding=(k - 1) // 2)
        x = xt + x
    return x


def decode(sd, cfg, x, s, return_pre_istft=False):
    """HiFTGenerator.decode: x mel [1,80,m], s source [1,1,480m] -> waveform [1,480m]."""
    win = hann16()
    spec = torch.view_as_real(torch.stft(s.squeeze(1), cfg.n_fft, cfg.hop, cfg.n_fft, window=win, return_complex=True))
    s_stft = torch.cat([spec[..., 0], spec[..., 1]], dim=1)
    x = F.conv1d(x, fold_weight_norm(sd, "conv_pre."), sd["conv_pre.bias"], padding=3)
    rates = np.cumprod([1] + cfg.ups[::-1][:-1])[::-1]
    nk = len(cfg.res_k)
    for i, (u, k) in enumerate(zip(cfg.ups, cfg.up_k)):
        x = F.leaky_relu(x, cfg.lrelu)
        x = F.conv_transpose1d(x, fold_weight_norm(sd, "ups.%d." % i), sd["ups.%d.bias" % i], stride=u, padding=(k - u) // 2)
        if i == len(cfg.ups) - 1:
            x = F.pad(x, (1, 0), mode="reflect")
        r = int(rates[i])
        si = F.conv1d(s_stft, sd["source_downs.%d.weight" % i], sd["source_downs.%d.bias" % i], stride=r, padding=(r // 2 if r > 1 else 0))
        si = resblock(sd, cfg, "source_resblocks.%d." % i, si, cfg.src_k[i])
        x = x + si
        xs = None
        for j in range(nk):
            y = resblock(sd, cfg, "resblocks.%d." % (i * nk + j), x, cfg.res_k[j])
            xs = y if xs is None else xs + y
        x = xs / nk
    x = F.leaky_relu(x)
    x = F.conv1d(x, fold_weight_norm(sd, "conv_post."), sd["conv_post.bias"], padding=3)
    nb = cfg.n_fft // 2 + 1
    mag = torch.exp(x[:, :nb, :])
    phase = torch.sin(x[:, nb:, :])
    if return_pre_istft:
        return x
    mag = torch.clip(mag, max=1e2)
    y = torch.istft(torch.complex(mag * torch.cos(phase), mag * torch.sin(phase)), cfg.n_fft, cfg.hop, cfg.n_fft, window=win)
    return torch.clamp(y, -cfg.audio_limit, cfg.audio_limit)


def inference(sd, cfg, speech_feat, cache_source=None, rand_ini=None, noise=None):
    """HiFTGenerator.inference (generator.py:557-569): returns (speech [1,480m], source [1,1,480m])."""
    m = speech_feat.shape[2]
    scale = int(np.prod(cfg.ups) * cfg.hop)
    L = m * scale
    f0 = f0_predictor(sd, speech_feat)
    f0_up = F.interpolate(f0[:, None], scale_factor=float(scale), mode="nearest").transpose(1, 2)
    if rand_ini is None:
        rand_ini = torch.zeros(1, cfg.harmonics + 1)
    if noise is None:
        noise = torch.zeros(1, L, cfg.harmonics + 1)
    s = source_module(sd, cfg, f0_up, rand_ini, noise).transpose(1, 2)
    if cache_source is not None and cache_source.shape[2] != 0:
        s[:, :, : cache_source.shape[2]] = cache_source
    return decode(sd, cfg, speech_feat, s), s


# ---------------------------------------------------------------------------------------------------------------------------------
# CausalHiFTGenerator (Fun-CosyVoice3): cosyvoice/hifigan/generator.py:572-726, hifigan/f0_predictor.py:62-103, transformer/convolution.py:150-259.
# `finalize=False` = a streaming chunk whose last frames are only look-ahead context.  The reference runs the f0 predictor in float64
# (generator.py:716-717); `f0_dtype` selects that (the oracle default) or float32 (what the device computes).
# ---------------------------------------------------------------------------------------------------------------------------------
def _causal_conv(x, w, b, dilation=1, right=False, cache=None):
    """CausalConv1d.forward (convolution.py:176-187): pad causal_padding zeros (or `cache`) on the left ('left') or right ('right')."""
    k = w.shape[2]
    pad = int((k * dilation - dilation) / 2) * 2 + (k + 1) % 2
    c = torch.zeros(x.shape[0], x.shape[1], pad, dtype=x.dtype) if cache is None else cache
    assert c.shape[2] == pad
    x = torch.cat([x, c], 2) if right else torch.cat([c, x], 2)
    return F.conv1d(x, w, b, dilation=dilation)


def causal_f0_predictor(sd, mel, finalize=True, dtype=torch.float64):
    x = mel.to(dtype)
    for j in range(5):
        p = "f0_predictor.condnet.%d." % (2 * j)
        w, b = fold_weight_norm(sd, p).to(dtype), sd[p + "bias"].to(dtype)
        if j == 0:
            pad = w.shape[2] - 1                              # causal_padding of the k = 4 'right' conv = 3
            x = _causal_conv(x, w, b, right=True) if finalize else _causal_conv(x[:, :, :-pad], w, b, right=True, cache=x[:, :, -pad:])
        else:
            x = _causal_conv(x, w, b)
        x = F.elu(x)
    return torch.abs(F.linear(x.transpose(1, 2), sd["f0_predictor.classifier.weight"].to(dtype), sd["f0_predictor.classifier.bias"].to(dtype)).squeeze(-1)).float()


def causal_sine_gen2(cfg, f0, rand_ini, noise):
    """SineGen2 with causal=True in eval mode (generator.py:233-258, 289-317): fixed rand_ini / uniform noise buffers, NEAREST phase upsampling."""
    H = cfg.harmonics + 1
    scale = int(np.prod(cfg.ups) * cfg.hop)
    fn = f0 * torch.arange(1, H + 1, dtype=torch.float32).reshape(1, 1, H)
    rad = (fn / cfg.sr) % 1
    rad[:, 0, :] = rad[:, 0, :] + rand_ini
    rad = F.interpolate(rad.transpose(1, 2), scale_factor=1 / scale, mode="linear").transpose(1, 2)
    phase = torch.cumsum(rad, dim=1) * 2 * np.pi
    phase = F.interpolate(phase.transpose(1, 2) * scale, scale_factor=scale, mode="nearest").transpose(1, 2)
    sines = torch.sin(phase) * cfg.nsf_alpha
    uv = (f0 > cfg.voiced_thr).float()
    noise_amp = uv * cfg.nsf_sigma + (1 - uv) * cfg.nsf_alpha / 3
    return sines * uv + noise_amp * noise


def causal_resblock(sd, cfg, p, x, k):
    for j, d in enumerate(cfg.res_d):
        xt = snake(x, sd[p + "activations1.%d.alpha" % j])
        xt = _causal_conv(xt, fold_weight_norm(sd, p + "convs1.%d." % j), sd[p + "convs1.%d.bias" % j], dilation=d)
        xt = snake(xt, sd[p + "activations2.%d.alpha" % j])
        xt = _causal_conv(xt, fold_weight_norm(sd, p + "convs2.%d." % j), sd[p + "convs2.%d.bias" % j])
        x = xt + x
    return x


def causal_decode(sd, cfg, x, s, finalize=True):
    """CausalHiFTGenerator.decode (generator.py:684-726)."""
    win = hann16()
    spec = torch.view_as_real(torch.stft(s.squeeze(1), cfg.n_fft, cfg.hop, cfg.n_fft, window=win, return_complex=True))
    re, im = spec[..., 0], spec[..., 1]
    lr, up = cfg.look_right, int(np.prod(cfg.ups))
    w, b = fold_weight_norm(sd, "conv_pre."), sd["conv_pre.bias"]
    if finalize:
        x = _causal_conv(x, w, b, right=True)
    else:
        x = _causal_conv(x[:, :, :-lr], w, b, right=True, cache=x[:, :, -lr:])
        re, im = re[:, :, :-up * lr], im[:, :, :-up * lr]
    s_stft = torch.cat([re, im], dim=1)
    rates = np.cumprod([1] + cfg.ups[::-1][:-1])[::-1]
    nk = len(cfg.res_k)
    for i, (u, k) in enumerate(zip(cfg.ups, cfg.up_k)):
        x = F.leaky_relu(x, cfg.lrelu)
        x = F.interpolate(x, scale_factor=float(u), mode="nearest")                       # CausalConv1dUpsample (convolution.py:248-259)
        x = F.conv1d(F.pad(x, (k - 1, 0)), fold_weight_norm(sd, "ups.%d." % i), sd["ups.%d.bias" % i])
        if i == len(cfg.ups) - 1:
            x = F.pad(x, (1, 0), mode="reflect")
        r = int(rates[i])
        if r == 1:
            si = F.conv1d(s_stft, sd["source_downs.%d.weight" % i], sd["source_downs.%d.bias" % i])
        else:                                                                             # CausalConv1dDownSample: left pad stride - 1
            si = F.conv1d(F.pad(s_stft, (r - 1, 0)), sd["source_downs.%d.weight" % i], sd["source_downs.%d.bias" % i], stride=r)
        si = causal_resblock(sd, cfg, "source_resblocks.%d." % i, si, cfg.src_k[i])
        x = x + si
        xs = None
        for j in range(nk):
            y = causal_resblock(sd, cfg, "resblocks.%d." % (i * nk + j), x, cfg.res_k[j])
            xs = y if xs is None else xs + y
        x = xs / nk
    x = F.leaky_relu(x)
    x = _causal_conv(x, fold_weight_norm(sd, "conv_post."), sd["conv_post.bias"])
    nb = cfg.n_fft // 2 + 1
    mag = torch.clip(torch.exp(x[:, :nb, :]), max=1e2)
    phase = torch.sin(x[:, nb:, :])
    y = torch.istft(torch.complex(mag * torch.cos(phase), mag * torch.sin(phase)), cfg.n_fft, cfg.hop, cfg.n_fft, window=win)
    if not finalize:
        y = y[:, :-up * cfg.hop]
    return torch.clamp(y, -cfg.audio_limit, cfg.audio_limit)


def causal_inference(sd, cfg, speech_feat, finalize=True, rand_ini=None, noise=None, f0_dtype=torch.float64):
    """CausalHiFTGenerator.inference (generator.py:713-726) -> (speech, source [1,1,L]).  noise [1, >= L, 9] = the model's fixed uniform buffer."""
    scale = int(np.prod(cfg.ups) * cfg.hop)
    f0 = causal_f0_predictor(sd, speech_feat, finalize, f0_dtype)
    L = f0.shape[1] * scale
    f0_up = F.interpolate(f0[:, None], scale_factor=float(scale), mode="nearest").transpose(1, 2)
    rand_ini = torch.zeros(1, cfg.harmonics + 1) if rand_ini is None else rand_ini
    noise = torch.zeros(1, L, cfg.harmonics + 1) if noise is None else noise[:, :L]
    s = torch.tanh(F.linear(causal_sine_gen2(cfg, f0_up, rand_ini, noise), sd["m_source.l_linear.weight"], sd["m_source.l_linear.bias"])).transpose(1, 2)
    x = speech_feat if finalize else speech_feat[:, :, :-3]
    return causal_decode(sd, cfg, x, s, finalize), s
