#!/bin/bash
# Round-3 validation call on one MI355X:  /usr/local/graft/bin/gpurun --timeout 1100 -- 'bash tools/gpu_r3_validate.sh [tag]'
# Full -m gpu suite, smoke, the persistent-pair probe, the default bench line (with its extras: batched 8 / 16, 8 streaming clients, mixed64,
# cosyvoice3), the rocprof kernel stats of the headline command and the FETCH_SIZE pass behind `roofline.traffic`.
# Everything lands in gpurun_out/<tag>/; what is worth keeping is copied to profiles/ by hand afterwards.
set -u
TAG=${1:-r3}
O=gpurun_out/$TAG; mkdir -p $O
R=$GRAFT_REPO_ROOT
run() { local name=$1; shift; local t0=$(date +%s); echo "== $name"; timeout -k 5 "$@" > $O/$name.log 2>&1; echo "   rc=$? $(( $(date +%s) - t0 ))s ($(tail -1 $O/$name.log | cut -c1-160))"; }
run pytest_gpu     600 python -X faulthandler -m pytest tests -q -m gpu -p no:cacheprovider --timeout 500
[ -f gpurun_out/r3_fullsize_errors.json ] && cp gpurun_out/r3_fullsize_errors.json $O/
run smoke           90 python -c "import __graft_entry__ as g; g.smoke()"
run persist_probe  120 tools/ubench/persist_probe
grep -E "RESULT|PASS|FAIL|us per layer|stamps|give-up" $O/persist_probe.log | cut -c1-230
run probe_flow_tail 200 python tools/probe_flow.py tail
grep -E "ms per flow|max \|" $O/probe_flow_tail.log
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
        if "cpu_baseline" in d: print("   cpu", {k: d["cpu_baseline"].get(k) for k in ("value", "cores", "kind", "stage_seconds")})
PY
CV_DOWN_DEEP=0 run bench_b8_down_split 200 env CV_DOWN_DEEP=0 python bench.py --no-extras --batch 8 --steps 4 --warmup 1 --no-cpu-baseline --first-chunk-reps 1
python - "$O/bench_b8_down_split.log" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        print("batch 8 with the round-2 down (split-K + sum_partials):", json.loads(line).get("batched_decode"))
PY
( cd /tmp && export TMPDIR=/tmp && timeout -k 5 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_bench -- \
    python $R/bench.py --no-extras --steps 2 --warmup 1 --no-cpu-baseline --first-chunk-reps 1 > $R/$O/prof_bench.log 2>&1; echo "== rocprof bench rc=$?" )
f=$(find $O/prof_bench -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/rocprof_bench_kernel_stats.csv && head -12 "$f" | cut -c1-170
rm -rf $O/prof_bench
( cd /tmp && export TMPDIR=/tmp && timeout -k 5 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_llm -- python $R/tools/profile_small.py llm > $R/$O/pmc_llm.log 2>&1; echo "== pmc rc=$?" )
python tools/pmc_summary.py $O/pmc_gemv_fetch.json $O/pmc_llm -- gemv | head -40
rm -rf $O/pmc_llm
( cd /tmp && export TMPDIR=/tmp && timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_flow -- python $R/tools/probe_flow.py profile > $R/$O/prof_flow.log 2>&1; echo "== rocprof flow rc=$?" )
f=$(find $O/prof_flow -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/rocprof_flow_kernel_stats.csv && head -10 "$f" | cut -c1-170
rm -rf $O/prof_flow
( cd /tmp && export TMPDIR=/tmp && timeout -k 5 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/$O/pmc_flow_sq -- python $R/tools/probe_flow.py profile > $R/$O/pmc_flow_sq.log 2>&1; echo "== pmc flow sq rc=$?" )
( cd /tmp && export TMPDIR=/tmp && timeout -k 5 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_flow_fetch -- python $R/tools/probe_flow.py profile > $R/$O/pmc_flow_fetch.log 2>&1; echo "== pmc flow fetch rc=$?" )
python tools/pmc_summary.py $O/pmc_flow.json $O/pmc_flow_sq $O/pmc_flow_fetch -- flow_tail flow_gemm attn_flow gemm_conv norm_rows | head -70
rm -rf $O/pmc_flow_sq $O/pmc_flow_fetch
