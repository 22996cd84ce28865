#!/bin/bash
# Round-3 call: full -m gpu suite, the decode weight-prefetch A/B (tools/probe_prefetch.py), the default bench line.
#   /usr/local/graft/bin/gpurun --timeout 1000 -- 'bash tools/gpu_r3_prefetch.sh <tag>'
set -u
TAG=${1:-r3p}
O=gpurun_out/$TAG; mkdir -p $O
run() { local name=$1; shift; local t0=$(date +%s); echo "== $name"; timeout -k 5 "$@" > $O/$name.log 2>&1; echo "   rc=$? $(( $(date +%s) - t0 ))s ($(tail -1 $O/$name.log | cut -c1-160))"; }
run probe_prefetch 240 python tools/probe_prefetch.py
grep -E "prefetch=" $O/probe_prefetch.log
run pytest_gpu     600 python -X faulthandler -m pytest tests -q -m gpu -p no:cacheprovider --timeout 500 -x
run bench_default  600 python bench.py
python - "$O/bench_default.log" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        d = json.loads(line)
        print("bench", d["value"], d["ms_per_step"], "first chunk", d.get("first_chunk_ms_p50"), d.get("stages"))
        for k in ("batched_decode", "batched_decode_16", "streaming_clients", "mixed64", "cosyvoice3"):
            print("  ", k, d.get(k))
        r = d["roofline"]; print("   roofline", {k: r.get(k) for k in ("achieved", "frac", "avg_launch_us", "traffic", "decode_stage", "decode_step_us_from_chains")})
        print("   per_kernel", {k[:28]: v.get("chain_avg_us") for k, v in r["per_kernel"].items()})
PY
