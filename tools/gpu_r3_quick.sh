#!/bin/bash
# Short A/B call: flow tail probe, HiFT two-sided split on / off, the affected GPU tests, the default bench line.
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/gpu_r3_quick.sh <tag>'
set -u
TAG=${1:-r3q}
O=gpurun_out/$TAG; mkdir -p $O
run() { local name=$1; shift; local t0=$(date +%s); echo "== $name"; timeout -k 5 "$@" > $O/$name.log 2>&1; echo "   rc=$? $(( $(date +%s) - t0 ))s ($(tail -1 $O/$name.log | cut -c1-160))"; }
run probe_flow_tail 200 python tools/probe_flow.py tail
grep -E "ms per flow|max \|" $O/probe_flow_tail.log
run pytest_subset   500 python -X faulthandler -m pytest tests/test_flow.py tests/test_hift.py tests/test_causal_hift.py tests/test_ops.py tests/test_zz_fullsize.py tests/test_zz_llm_batch.py tests/test_model_batch.py -q -m gpu -p no:cacheprovider --timeout 400
run bench_hift_chain 120 env CV_GEMM_WX3=0 python bench.py --no-extras --steps 3 --warmup 1 --no-cpu-baseline --first-chunk-reps 1
run bench_default   600 python bench.py
for f in bench_hift_chain bench_default; do python - "$O/$f.log" "$f" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        d = json.loads(line)
        print(sys.argv[2], d["value"], d["ms_per_step"], "first chunk", d.get("first_chunk_ms_p50"), d.get("stages"))
        for k in ("batched_decode", "batched_decode_16", "streaming_clients", "mixed64", "cosyvoice3"):
            if k in d: print("  ", k, d.get(k))
        r = d["roofline"]; print("   roofline", {k: r.get(k) for k in ("achieved", "frac", "avg_launch_us", "decode_step_us_from_chains")})
        print("   per_kernel", {k[:28]: v.get("chain_avg_us") for k, v in r["per_kernel"].items()})
        if "cpu_baseline" in d: print("   cpu", {k: d["cpu_baseline"].get(k) for k in ("value", "cores", "kind", "stage_seconds")})
PY
done
