"""Build the gfx950 shared library (hipcc) in-tree.

`build_hip()` cross-compiles every HIP source under cosyvoice_amd/csrc for gfx950 and links
cosyvoice_amd/libcosyvoice_amd.so (works without a GPU).  The CPU emulator build used by the
test-suite lives in tests/emu/build_emu.py and is never produced or loaded from here.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(ROOT, "csrc")
LIB = os.path.join(ROOT, "libcosyvoice_amd.so")
OBJ_DIR = os.path.join(ROOT, "build")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def hip_sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps_mtime():
    m = 0.0
    for d in (CSRC, os.path.join(CSRC, "experiments"), os.path.join(ROOT, "..", "include")):
        for f in os.listdir(d):
            m = max(m, os.path.getmtime(os.path.join(d, f)))
    return m


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("command failed: %s\n%s" % (" ".join(cmd), r.stdout))
    return r.stdout


def build_hip(force=False, verbose=False):
    srcs = hip_sources()
    newest = _deps_mtime()
    os.makedirs(OBJ_DIR, exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
             "-I", CSRC, "-I", os.path.join(ROOT, "..", "include"), "-Wno-unused-result"]
    if os.environ.get("CV_BUILD_EXPERIMENTS") == "1":       # the measured no-go variants (csrc/experiments/, attn_flow_kernel, qkv_attn_kernel ...): A/B builds only
        flags.append("-DCV_BUILD_EXPERIMENTS")
    stamp = os.path.join(OBJ_DIR, ".flags")                 # a changed flag set rebuilds everything (the objects are otherwise judged by their age alone)
    if not os.path.exists(stamp) or open(stamp).read() != " ".join(flags):
        force = True
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= newest:
        return LIB

    def compile_one(src):
        obj = os.path.join(OBJ_DIR, os.path.basename(src) + ".o")
        if not force and os.path.exists(obj) and os.path.getmtime(obj) >= newest:
            return obj
        out = _run([HIPCC, *flags, "-c", src, "-o", obj])
        if verbose and out.strip():
            print(out)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(compile_one, srcs))
    _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs])
    open(stamp, "w").write(" ".join(flags))
    return LIB


if __name__ == "__main__":
    print(build_hip(force="--force" in sys.argv, verbose=True))
