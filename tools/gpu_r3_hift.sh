#!/bin/bash
# HiFT A/B call: Snake once per value vs prologue Snake, kernel stats, affected GPU tests.
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/gpu_r3_hift.sh <tag>'
set -u
TAG=${1:-r3h}
O=gpurun_out/$TAG; mkdir -p $O
R=$GRAFT_REPO_ROOT
run() { local name=$1; shift; local t0=$(date +%s); echo "== $name"; timeout -k 5 "$@" > $O/$name.log 2>&1; echo "   rc=$? $(( $(date +%s) - t0 ))s ($(tail -1 $O/$name.log | cut -c1-200))"; }
run pytest_hift 400 python -X faulthandler -m pytest tests/test_hift.py tests/test_causal_hift.py tests/test_zz_fullsize.py tests/test_llm.py -q -m gpu -p no:cacheprovider --timeout 300
run bench_snake_pro 150 env CV_HIFT_SNAKE_ONCE=0 python bench.py --no-extras --steps 5 --warmup 2 --no-cpu-baseline --first-chunk-reps 2
run bench_snake_once 150 python bench.py --no-extras --steps 5 --warmup 2 --no-cpu-baseline --first-chunk-reps 2
run bench_snake_once_chain 150 env CV_GEMM_WX3=0 python bench.py --no-extras --steps 5 --warmup 2 --no-cpu-baseline --first-chunk-reps 2
for f in bench_snake_pro bench_snake_once bench_snake_once_chain; do python - "$O/$f.log" "$f" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        d = json.loads(line)
        print(sys.argv[2], d["value"], d["ms_per_step"], "first chunk", d.get("first_chunk_ms_p50"), d.get("stages"))
PY
done
( cd /tmp && export TMPDIR=/tmp && timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_hift -- python $R/tools/profile_small.py hift > $R/$O/prof_hift.log 2>&1; echo "== rocprof hift rc=$?" )
f=$(find $O/prof_hift -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/rocprof_hift_snake_once_kernel_stats.csv && head -14 "$f" | cut -c1-220
rm -rf $O/prof_hift
