"""Architecture constants of the CosyVoice models served by this package (the reference keeps them in
hyperpyyaml files, examples/libritts/cosyvoice2/conf/cosyvoice2.yaml).  `cv2()` = CosyVoice2-0.5B; `tiny()` = same topology with
small dims, used by the emulator-sized tests."""
from dataclasses import dataclass, field
from typing import List


@dataclass
class LLMConfig:                      # Appendix A.1 (cosyvoice2.yaml:23-36 + Qwen2.5-0.5B config)
    hidden: int = 896
    layers: int = 24
    heads: int = 14
    kv_heads: int = 2
    head_dim: int = 64
    inter: int = 4864
    text_vocab: int = 151936
    speech_token_size: int = 6561
    rms_eps: float = 1e-6
    rope_theta: float = 1e6
    # CosyVoice3LM (llm/llm.py:664-706): sos / eos / task_id / fill are rows speech_token_size + {0,1,2,3} of speech_embedding
    # (no llm_embedding table), the head is Linear(hidden, speech_token_size + 200, bias=False) and all 200 extra ids stop decoding.
    cv3: bool = False
    n_special: int = 3                # ids >= speech_token_size: 3 for Qwen2LM (eos, +1, fill), 200 for CosyVoice3LM
    endofprompt_id: int = 151646      # <|endofprompt|>, required in the text of CosyVoice3 requests (llm/llm.py:478-480)


@dataclass
class FlowConfig:                     # Appendix A.2 / A.3 (cosyvoice2.yaml:38-87)
    vocab: int = 6561
    dim: int = 512                    # encoder width
    enc_heads: int = 8
    ffn: int = 2048
    enc_blocks: int = 6
    up_blocks: int = 4
    spk_dim: int = 192
    mel: int = 80
    est_ch: int = 256                 # estimator channels
    est_heads: int = 8
    est_blocks: int = 4               # transformer blocks per resnet
    est_mid: int = 12
    pre_lookahead: int = 3
    chunk: int = 25                   # static_chunk_size in tokens (estimator: chunk * 2 frames)
    cfg_rate: float = 0.7
    n_timesteps: int = 10
    # estimator family: "unet" = CausalConditionalDecoder (CosyVoice2), "dit" = DiT (Fun-CosyVoice3, flow/DiT/dit.py:104-176).  For "dit" the fields
    # are read as: dim = flow input_size = mu_dim (80), ffn = PreLookaheadLayer channels (1024), est_ch = DiT width (1024), est_heads (16),
    # est_blocks = depth (22), est_mid = ff_mult (2); there is no conformer encoder (enc_blocks = up_blocks = 0), chunk = static_chunk_size / 2.
    estimator: str = "unet"


@dataclass
class HiftConfig:                     # Appendix A.4 (cosyvoice2.yaml:89-111)
    mel: int = 80
    base: int = 512
    harmonics: int = 8
    sr: int = 24000
    ups: List[int] = field(default_factory=lambda: [8, 5, 3])
    up_k: List[int] = field(default_factory=lambda: [16, 11, 7])
    res_k: List[int] = field(default_factory=lambda: [3, 7, 11])
    res_d: List[int] = field(default_factory=lambda: [1, 3, 5])
    src_k: List[int] = field(default_factory=lambda: [7, 7, 11])
    n_fft: int = 16
    hop: int = 4
    f0_ch: int = 512
    nsf_alpha: float = 0.1
    nsf_sigma: float = 0.003
    voiced_thr: float = 10.0
    lrelu: float = 0.1
    audio_limit: float = 0.99
    # CausalHiFTGenerator (Fun-CosyVoice3, hifigan/generator.py:572-726): every conv causal, conv_pre / the first f0 conv look `look_right` / 3
    # frames ahead, nearest-neighbour upsampling convs instead of transposed convs, nearest phase interpolation in the harmonic source
    causal: bool = False
    look_right: int = 4


@dataclass
class CV1Config:                      # CosyVoice-300M (examples/libritts/cosyvoice/conf/cosyvoice.yaml), SURVEY.md section 8 row a18
    text_vocab: int = 51866
    speech_token_size: int = 4096
    text_enc_in: int = 512            # text_encoder_input_size
    llm_dim: int = 1024               # text encoder output = llm_input_size = llm_output_size
    text_heads: int = 16
    text_ffn: int = 4096
    text_blocks: int = 6
    llm_heads: int = 16
    llm_ffn: int = 4096
    llm_blocks: int = 14
    spk_dim: int = 192
    flow_dim: int = 512               # MaskedDiffWithXvec input_size = encoder output
    flow_heads: int = 8
    flow_ffn: int = 2048
    flow_blocks: int = 6
    mel: int = 80
    regulator_layers: int = 4         # InterpolateRegulator sampling_ratios [1, 1, 1, 1]
    est_ch: List[int] = field(default_factory=lambda: [256, 256])
    est_heads: int = 8
    est_head_dim: int = 64
    est_blocks: int = 4
    est_mid: int = 12
    input_frame_rate: int = 50


def cv1():
    """CosyVoice-300M: (CV1Config, HiftConfig of the 22.05 kHz HiFTGenerator, cosyvoice.yaml:113-140)."""
    return CV1Config(), HiftConfig(sr=22050, ups=[8, 8], up_k=[16, 16], src_k=[7, 11])


def tiny_cv1():
    return (CV1Config(text_vocab=50, speech_token_size=40, text_enc_in=32, llm_dim=64, text_heads=4, text_ffn=128, text_blocks=2, llm_heads=4, llm_ffn=128,
                      llm_blocks=3, spk_dim=16, flow_dim=64, flow_heads=4, flow_ffn=128, flow_blocks=2, est_ch=[32, 32], est_heads=2, est_head_dim=16,
                      est_blocks=1, est_mid=2),
            HiftConfig(sr=22050, ups=[8, 8], up_k=[16, 16], src_k=[7, 11], base=32, f0_ch=32))


def tiny_cv1_k():
    """tiny_cv1 with 64-wide attention heads everywhere (what cv_attention serves, and what CosyVoice-300M has: 1024 / 16, 512 / 8, head_dim 64):
    the emulator-sized configuration of the kernel-backed path (cosyvoice1_hip.py)."""
    return (CV1Config(text_vocab=50, speech_token_size=40, text_enc_in=32, llm_dim=128, text_heads=2, text_ffn=128, text_blocks=2, llm_heads=2, llm_ffn=128,
                      llm_blocks=2, spk_dim=16, flow_dim=128, flow_heads=2, flow_ffn=128, flow_blocks=2, est_ch=[32, 32], est_heads=1, est_head_dim=64,
                      est_blocks=1, est_mid=2),
            HiftConfig(sr=22050, ups=[8, 8], up_k=[16, 16], src_k=[7, 11], base=32, f0_ch=32))


def cv2():
    return LLMConfig(), FlowConfig(), HiftConfig()


def tiny_cv3_llm():
    """Emulator-sized CosyVoice3LM; the text vocabulary still has to reach the hard-coded <|endofprompt|> id 151646."""
    return LLMConfig(hidden=128, layers=2, heads=2, kv_heads=1, inter=256, text_vocab=151650, speech_token_size=60, cv3=True, n_special=200)


def cv3_llm():
    """The LM of Fun-CosyVoice3-0.5B (cosyvoice3.yaml:23-36): same Qwen2.5-0.5B backbone, CosyVoice3LM head / embedding layout."""
    return LLMConfig(cv3=True, n_special=200)


def cv3_hift():
    """CausalHiFTGenerator of Fun-CosyVoice3-0.5B (cosyvoice3.yaml:77-100)."""
    return HiftConfig(causal=True)


def cv3_flow():
    """CausalMaskedDiffWithDiT of Fun-CosyVoice3-0.5B (cosyvoice3.yaml:38-75)."""
    return FlowConfig(vocab=6561, dim=80, enc_heads=0, ffn=1024, enc_blocks=0, up_blocks=0, est_ch=1024, est_heads=16, est_blocks=22, est_mid=2,
                      chunk=25, estimator="dit")


def tiny_cv3_flow():
    return FlowConfig(vocab=60, dim=80, enc_heads=0, ffn=64, enc_blocks=0, up_blocks=0, spk_dim=32, est_ch=128, est_heads=2, est_blocks=2, est_mid=2,
                      chunk=5, n_timesteps=2, estimator="dit")


def tiny():
    llm = LLMConfig(hidden=128, layers=2, heads=2, kv_heads=1, inter=256, text_vocab=300, speech_token_size=60)
    flow = FlowConfig(vocab=60, dim=128, enc_heads=2, ffn=256, enc_blocks=2, up_blocks=1, spk_dim=32, est_ch=64, est_heads=1,
                      est_blocks=1, est_mid=2)
    hift = HiftConfig(base=64, f0_ch=32)
    return llm, flow, hift
