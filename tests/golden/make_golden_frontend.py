"""Golden vectors for the request assembly of the front end ("spk2info.pt caching as the reference does"), made by the REAL
cosyvoice.cli.frontend.CosyVoiceFrontEnd - build container only:

    python tests/golden/make_golden_frontend.py

The class is imported from /root/reference (whisper / inflect / onnxruntime / torchaudio are import-time stubs, never called here) and instantiated WITHOUT its
constructor - that one opens two .onnx files and a tokenizer vocabulary no box has - with the tokenizer and the three prompt extractors replaced by the deterministic
stand-ins of tests/frontend_fakes.py.  What runs from the reference is what is under test: frontend_sft / frontend_zero_shot (the forced 2:1 mel / token ratio at
24 kHz, the cached-speaker branch) / frontend_cross_lingual / frontend_instruct / frontend_instruct2 / frontend_vc (cli/frontend.py:157-224), _extract_text_token with a
text generator (:86-101), text_normalize's pass-through paths (:127-133) and the speaker registration of cli/cosyvoice.py:69-75.  Every model_input dict is stored
(tests/golden/frontend_requests.npz + the key lists in frontend_requests.json); tests/test_frontend_requests.py holds cosyvoice_amd.frontend.CosyVoiceFrontEnd to them."""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(REPO, "tests"))
import ref_import  # noqa: E402

ref_import.install()
for _n in ("whisper", "inflect"):
    sys.modules.setdefault(_n, types.ModuleType(_n))
import cosyvoice.cli.frontend as F  # noqa: E402
import frontend_fakes as FK  # noqa: E402


def real_front_end():
    fe = object.__new__(F.CosyVoiceFrontEnd)
    fe.tokenizer, fe.device, fe.allowed_special, fe.spk2info, fe.text_frontend = FK.FakeTokenizer(), torch.device("cpu"), "all", {}, ""
    return FK.install(fe)


def add_zero_shot_spk(fe, prompt_text, prompt_wav, spk_id):
    """cosyvoice/cli/cosyvoice.py:69-75 (CosyVoice.add_zero_shot_spk), statement by statement on the front end it edits."""
    assert spk_id != ""
    model_input = fe.frontend_zero_shot("", prompt_text, prompt_wav, 24000, "")
    del model_input["text"]
    del model_input["text_len"]
    fe.spk2info[spk_id] = model_input
    return True


def main():
    fe = real_front_end()
    got = FK.cases(fe, add_zero_shot_spk)
    arrs, keys = {}, {}
    for case, d in got.items():
        keys[case] = sorted(d)
        for k, v in d.items():
            assert torch.is_tensor(v), (case, k, type(v))
            arrs[case + "/" + k] = v.numpy()
    np.savez_compressed(os.path.join(HERE, "frontend_requests.npz"), **arrs)
    norm = FK.normalize_cases(real_front_end())
    with open(os.path.join(HERE, "frontend_requests.json"), "w") as f:
        json.dump({"keys": keys, "dtypes": {k: str(v.dtype) for k, v in arrs.items()}, "text_normalize": norm}, f, indent=1, sort_keys=True)
    print("wrote frontend_requests.npz (%d arrays, %.0f KB) and frontend_requests.json; cases: %s"
          % (len(arrs), os.path.getsize(os.path.join(HERE, "frontend_requests.npz")) / 1024, ", ".join(keys)))


if __name__ == "__main__":
    main()
