#!/bin/bash
# Round 4, ninth call: scalar activation dispatch (act4_call) - phases, per-kernel times at 8 utterances per pass and at batch 1, headline.
set -u
O=gpurun_out/r4i; mkdir -p $O
R=$GRAFT_REPO_ROOT
run() { local name=$1; shift; local t0=$(date +%s); echo "== $name"; timeout -k 5 "$@" > $O/$name.log 2>&1; echo "   rc=$? $(( $(date +%s) - t0 ))s ($(tail -1 $O/$name.log | cut -c1-200))"; }
run pytest_act 400 python -m pytest tests/test_ops.py tests/test_flow.py tests/test_hift.py tests/test_dit.py -q -m gpu -p no:cacheprovider -x
run phases 300 python tools/probe_flow_phases.py
grep -E "64x64 lds_epilogue=1|QG=1" $O/phases.log | cut -c1-260
run probe 100 python tools/probe_flow_big2.py cfg=-1,3,3
grep "nu=" $O/probe.log
for nu in 1 8; do
  ( cd /tmp && export TMPDIR=/tmp && timeout -k 5 150 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_fb$nu -- python $R/tools/profile_flow_batch.py $nu > $R/$O/prof_fb$nu.log 2>&1; echo "== rocprof flow batch $nu rc=$? $(tail -1 $R/$O/prof_fb$nu.log)" )
  f=$(find $O/prof_fb$nu -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/rocprof_flow_batch${nu}_kernel_stats.csv && head -12 "$f" | cut -c1-190
  rm -rf $O/prof_fb$nu
done
run bench_headline 300 python bench.py --gpus 1 --steps 8 --warmup 3 --no-extras --no-cpu-baseline
python - "$O/bench_headline.log" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        d = json.loads(line); print("bench", d["value"], d["ms_per_step"], "first chunk", d.get("first_chunk_ms_p50"), d.get("stages"))
PY
