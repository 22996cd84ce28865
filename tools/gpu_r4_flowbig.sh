#!/bin/bash
# Round 4, second call: the large-M flow kernels (flow_big.h) on the hardware - bit-identity tests, wall-clock A/B per utterances-per-pass and tile shape, per-kernel rocprof.
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/gpu_r4_flowbig.sh'
set -u
O=gpurun_out/r4b; mkdir -p $O
R=$GRAFT_REPO_ROOT
run() { local name=$1; shift; local t0=$(date +%s); echo "== $name"; timeout -k 5 "$@" > $O/$name.log 2>&1; echo "   rc=$? $(( $(date +%s) - t0 ))s ($(tail -1 $O/$name.log | cut -c1-160))"; }
run pytest_flow 400 python -m pytest tests/test_flow.py tests/test_ops.py -q -m gpu -p no:cacheprovider -x
run probe_big 300 python tools/probe_flow_big.py check
cat $O/probe_big.log | grep -E "nu=|bit-identical"
( cd /tmp && export TMPDIR=/tmp && timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_big -- python $R/tools/probe_flow_big.py profile > $R/$O/prof_big.log 2>&1; echo "== rocprof probe_flow_big rc=$?" )
f=$(find $O/prof_big -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/rocprof_flow_big_kernel_stats.csv && head -24 "$f" | cut -c1-200
rm -rf $O/prof_big
run bench_headline 300 python bench.py --gpus 1 --steps 8 --warmup 3 --no-extras --no-cpu-baseline
python - "$O/bench_headline.log" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        d = json.loads(line); print("bench", d["value"], d["ms_per_step"], "first chunk", d.get("first_chunk_ms_p50"), d.get("stages"))
PY
