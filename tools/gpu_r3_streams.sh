#!/bin/bash
# Two-stream estimator A/B: flow.inference at the U10 size, batched passes, the flow GPU tests, stage times of the bench.
set -u
TAG=${1:-r3s}
O=gpurun_out/$TAG; mkdir -p $O
run() { local name=$1; shift; local t0=$(date +%s); echo "== $name"; timeout -k 5 "$@" > $O/$name.log 2>&1; echo "   rc=$? $(( $(date +%s) - t0 ))s ($(tail -1 $O/$name.log | cut -c1-200))"; }
run probe_flow_streams 200 python tools/probe_flow.py streams
grep -E "ms per flow|identical" $O/probe_flow_streams.log
run probe_flow_batch 200 python tools/probe_flow_batch.py
grep -E "flow pass" $O/probe_flow_batch.log
run pytest_flow 500 python -X faulthandler -m pytest tests/test_flow.py tests/test_model_batch.py tests/test_zz_fullsize.py -q -m gpu -p no:cacheprovider --timeout 400 -k "not llm"
run bench_quick 150 python bench.py --no-extras --steps 5 --warmup 2 --no-cpu-baseline --first-chunk-reps 3
python - "$O/bench_quick.log" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        d = json.loads(line)
        print("bench", d["value"], d["ms_per_step"], "first chunk", d.get("first_chunk_ms_p50"), d.get("stages"))
PY
