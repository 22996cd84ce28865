"""TEST INFRASTRUCTURE: compile the UNMODIFIED kernel sources of cosyvoice_amd/csrc with the host
clang++ against the HIP emulator in this directory -> tests/emu/libcosyvoice_amd_emu.so.

Used only by the test-suite (this container has no GPU); see hip/hip_runtime.h.
"""
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(REPO, "cosyvoice_amd", "csrc")
LIB = os.path.join(HERE, "libcosyvoice_amd_emu.so")
OBJ_DIR = os.path.join(HERE, "build")
CXX = os.environ.get("EMU_CXX", "/opt/rocm/lib/llvm/bin/clang++")


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("command failed: %s\n%s" % (" ".join(cmd), r.stdout))
    return r.stdout


def build_emu(force=False):
    """Serialised across processes (pytest-xdist workers start together on a fresh checkout): one builds, the others wait and find it up to date."""
    import fcntl
    os.makedirs(OBJ_DIR, exist_ok=True)
    with open(os.path.join(OBJ_DIR, ".lock"), "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        return _build_emu(force)


def _build_emu(force=False):
    srcs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    srcs.append(os.path.join(HERE, "emu_runtime.cpp"))
    newest = 0.0
    for d in (CSRC, os.path.join(CSRC, "experiments"), os.path.join(REPO, "include"), HERE, os.path.join(HERE, "hip")):
        for f in os.listdir(d):
            if f.endswith((".h", ".hip", ".cpp")):
                newest = max(newest, os.path.getmtime(os.path.join(d, f)))
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= newest:
        return LIB
    os.makedirs(OBJ_DIR, exist_ok=True)
    flags = ["-O2", "-g", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-stack-protector", "-DCV_BUILD_EXPERIMENTS",      # the CPU suite keeps every variant under test
             "-I", HERE, "-I", CSRC, "-I", os.path.join(REPO, "include"),
             "-Wno-unused-value", "-Wno-unknown-attributes"]

    def compile_one(src):
        obj = os.path.join(OBJ_DIR, os.path.basename(src) + ".o")
        if not force and os.path.exists(obj) and os.path.getmtime(obj) >= newest:
            return obj
        _run([CXX, *flags, "-x", "c++", "-c", src, "-o", obj])
        return obj

    with ThreadPoolExecutor(max_workers=8) as ex:
        objs = list(ex.map(compile_one, srcs))
    _run([CXX, "-shared", "-fPIC", "-o", LIB, *objs, "-lpthread"])
    return LIB


if __name__ == "__main__":
    print(build_emu(force=True))
