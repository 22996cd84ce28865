"""CPU oracle of the CosyVoice2 synthesis hot path — TEST INFRASTRUCTURE, not product code.

A plain-torch fp32 restatement of the reference algorithm (FunAudioLLM/CosyVoice, /root/reference), every function
citing the reference file:line it follows.  Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may
import this package, and only as the checker / reported baseline — the product path (cosyvoice_amd/) never does.

Pinning status: the reference ships NO golden vectors, known-answer tests or fixtures for this path.
  * Pieces whose reference code imports in the build container (Qwen2LM over transformers' Qwen2, ras_sampling,
    UpsampleConformerEncoder, HiFTGenerator, masks, CosyVoice2Model.token2wav glue, and — through a stub of the
    Matcha primitives — the reference's own flow/decoder.py + flow/flow_matching.py control flow) are pinned by
    golden vectors generated from the REAL reference: tests/golden/make_golden.py -> tests/golden/*.npz,
    checked by tests/test_oracle_golden.py.
  * The Matcha-TTS primitives (SinusoidalPosEmb, TimestepEmbedding, ResnetBlock1D, BasicTransformerBlock incl.
    diffusers-0.29 Attention) are an un-vendored, un-pinned submodule: PARITY UNPINNED for those — they are restated
    from the published upstream code (tests/golden/matcha_stub.py) and cross-checked only by
    state-dict key/shape compatibility.
  * transformers.Qwen2ForCausalLM (pinned 4.51.3 in the reference, 5.15 installed here) is restated in oracle/llm.py and
    pinned against the installed implementation through the same golden vectors.

Precision policy shared with the HIP path ("W16A32"): LLM and flow weights are rounded to bf16 once (weights.py),
all arithmetic is fp32; HiFT is fp32 throughout (the reference always runs it in fp32, cli/model.py:312).
`oracle.flow.bf16_act()` additionally mirrors the product's "bf16" mode (operands of the flow's Linear / Conv1d / attention
products rounded to bfloat16, fp32 accumulation) for the statistical checks of that mode.

The reference is Python: there is no compiled reference to build (`oracle/_ref` does not apply); it is imported in the build
container only by tests/golden/make_golden.py, which wrote the committed fixtures.
"""
