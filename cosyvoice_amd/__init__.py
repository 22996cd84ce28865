"""cosyvoice_amd — MI355X-native (gfx950) CosyVoice2 synthesis hot path.

LLM speech-token decode -> flow-matching mel decoder -> HiFT vocoder, as hand-written HIP kernels
behind a C ABI (include/cosyvoice_amd.h), with a Python host mirroring the reference's
cosyvoice/cli/model.py objects.  See DESIGN.md.
"""
__version__ = "0.1.0"
