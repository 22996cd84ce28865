"""Static per-kernel resource table of the gfx950 library (no GPU needed): every .hip of cosyvoice_amd/csrc compiled with the build's own flags plus
`-Rpass-analysis=kernel-resource-usage`, the remarks folded into one line per kernel - VGPRs / AGPRs / SGPRs, scratch bytes per lane, spills, LDS bytes per
workgroup, the occupancy the compiler derives from them.

    python tools/kernel_resources.py > profiles/r5_kernel_resources.txt

What it is for: the claims DESIGN.md makes about register arrays and scratch (section 8 item 11) can be read off a file instead of taken on trust, and a change that
pushes a hot kernel over a register or LDS step shows up in a diff of this table before any GPU minute is spent."""
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "cosyvoice_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FIELDS = [("VGPRs", "vgpr"), ("AGPRs", "agpr"), ("TotalSGPRs", "sgpr"), ("ScratchSize [bytes/lane]", "scratch"), ("VGPRs Spill", "vspill"), ("SGPRs Spill", "sspill"),
          ("LDS Size [bytes/block]", "lds"), ("Occupancy [waves/SIMD]", "occ")]


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), stdout=subprocess.PIPE, text=True, check=True).stdout.split("\n")
        return dict(zip(names, out))
    except Exception:                                           # noqa: BLE001
        return {n: n for n in names}


def remarks(src):
    with tempfile.TemporaryDirectory() as d:
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-I", CSRC, "-I", os.path.join(ROOT, "include"),
                            "-Wno-unused-result", "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", os.path.join(d, "o.o")],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr[-2000:])
    rows, cur = [], None
    for line in r.stderr.split("\n"):
        m = re.search(r"remark:\s+Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1), "where": os.path.basename(line.split(":")[0]) + ":" + line.split(":")[1]}
            rows.append(cur)
            continue
        for label, key in FIELDS:
            m = re.search(r"remark:\s+" + re.escape(label) + r": (\d+)", line)
            if m and cur is not None:
                cur[key] = int(m.group(1))
    return rows


def short(name):
    name = re.sub(r"\(.*\)$", "", name)                         # drop the argument list
    name = name.replace("cv::(anonymous namespace)::", "").replace("(anonymous namespace)::", "").replace("cv::", "").replace("void ", "")
    return name


def main():
    srcs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    with ThreadPoolExecutor(max_workers=8) as ex:
        per = list(ex.map(remarks, srcs))
    names = demangle(sorted({r["name"] for rows in per for r in rows}))
    print("# gfx950 kernel resources (hipcc -O3, the library's build flags; `-Rpass-analysis=kernel-resource-usage`).  One line per kernel INSTANTIATION per translation unit.")
    print("# vgpr + agpr <= 512 per lane; occupancy = waves per SIMD the register / LDS use allows; scratch > 0 = private memory in HBM (spills or an indexed local array)")
    print("%-16s %-92s %5s %5s %5s %8s %7s %7s %8s %4s" % ("unit", "kernel", "vgpr", "agpr", "sgpr", "scratch", "vspill", "sspill", "lds", "occ"))
    n_scratch = 0
    for src, rows in zip(srcs, per):
        for r in sorted(rows, key=lambda r: names[r["name"]]):
            n_scratch += r.get("scratch", 0) > 0
            print("%-16s %-92s %5d %5d %5d %8d %7d %7d %8d %4d" % (os.path.basename(src), short(names[r["name"]])[:92], r.get("vgpr", -1), r.get("agpr", -1), r.get("sgpr", -1),
                                                                  r.get("scratch", -1), r.get("vspill", -1), r.get("sspill", -1), r.get("lds", -1), r.get("occ", -1)))
    total = sum(len(rows) for rows in per)
    print("# %d kernel instantiations, %d with scratch > 0" % (total, n_scratch))


if __name__ == "__main__":
    sys.exit(main())
