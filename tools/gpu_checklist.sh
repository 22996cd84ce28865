#!/bin/bash
# One gpurun call that validates everything added after the last hardware run of round 1 and collects the first numbers for it:
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/gpu_checklist.sh'
# Writes under gpurun_out/checklist/.  Order: cheapest / most important first; every step has its own timeout.
set -u
O=gpurun_out/checklist; mkdir -p $O
run() { local name=$1; shift; echo "== $name"; timeout -k 5 "$@" > $O/$name.log 2>&1; echo "   rc=$? ($(tail -1 $O/$name.log | cut -c1-150))"; }
run pytest_gpu            300 python -X faulthandler -m pytest tests -q -m gpu -s
run pytest_batch_decode   120 env CV_TEST_BATCH_DECODE=1 python -X faulthandler -m pytest tests/test_zz_llm_batch.py -q -m gpu
run smoke                  60 python -c "import __graft_entry__ as g; g.smoke()"
run bench_default         240 python bench.py
run bench_batch8          240 python bench.py --no-cpu-baseline --first-chunk-reps 1 --batch 8
run bench_batch4          200 python bench.py --no-cpu-baseline --first-chunk-reps 1 --batch 4
run bench_fp32            200 python bench.py --no-cpu-baseline --first-chunk-reps 1 --flow-precision fp32
run chain_probe            60 tools/ubench/chain_probe
( cd /tmp && export TMPDIR=/tmp && timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/prof_bench -- \
    python $OLDPWD/bench.py --steps 1 --warmup 1 --no-cpu-baseline --first-chunk-reps 1 > $OLDPWD/$O/prof_bench.log 2>&1; echo "== rocprof bench rc=$?" )
grep -h "rel L2\|SNR\|golden" $O/pytest_gpu.log | head -30
for f in bench_default bench_batch8 bench_batch4 bench_fp32; do python - "$O/$f.log" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        d = json.loads(line); print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], d.get("first_chunk_ms_p50"), d.get("batched_decode"))
PY
done
