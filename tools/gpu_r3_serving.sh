#!/bin/bash
# Round-3 call: 8 streaming clients with the scheduler's shared flow passes off / 4 / 8 (CV_CHUNK_BATCH) and a shorter interpreter switch interval,
# the flow tail per batch size, then the profiles behind DESIGN section 6 (rocprof kernel stats of the headline command, flow PMC passes).
#   /usr/local/graft/bin/gpurun --timeout 1100 -- 'bash tools/gpu_r3_serving.sh <tag>'
set -u
TAG=${1:-r3s}
O=gpurun_out/$TAG; mkdir -p $O
R=$GRAFT_REPO_ROOT
run() { local name=$1; shift; local t0=$(date +%s); echo "== $name"; timeout -k 5 "$@" > $O/$name.log 2>&1; echo "   rc=$? $(( $(date +%s) - t0 ))s ($(tail -1 $O/$name.log | cut -c1-100))"; }
S="python bench.py --no-extras --stream-clients 8 --steps 1 --warmup 1 --no-cpu-baseline --first-chunk-reps 1 --lanes 4"
run stream_cb1   200 env CV_CHUNK_BATCH=1 $S
run stream_cb4   200 env CV_CHUNK_BATCH=4 $S
run stream_cb8   200 env CV_CHUNK_BATCH=8 $S
run stream_cb4_sw 200 env CV_CHUNK_BATCH=4 CV_SWITCH_INTERVAL=0.0005 $S
run stream_cb4_l2 200 env CV_CHUNK_BATCH=4 python bench.py --no-extras --stream-clients 8 --steps 1 --warmup 1 --no-cpu-baseline --first-chunk-reps 1 --lanes 2
for f in stream_cb1 stream_cb4 stream_cb8 stream_cb4_sw stream_cb4_l2; do python - "$O/$f.log" "$f" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        d = json.loads(line); print(sys.argv[2], d.get("streaming_clients"))
PY
done
run probe_flow_batch_tail 240 python tools/probe_flow_batch.py tail
grep -E "fused_tail|flow pass" $O/probe_flow_batch_tail.log
run probe_flow_tail 200 python tools/probe_flow.py tail
grep -E "ms per flow|max \|" $O/probe_flow_tail.log
( cd /tmp && export TMPDIR=/tmp && timeout -k 5 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_bench -- \
    python $R/bench.py --no-extras --steps 2 --warmup 1 --no-cpu-baseline --first-chunk-reps 1 > $R/$O/prof_bench.log 2>&1; echo "== rocprof bench rc=$?" )
f=$(find $O/prof_bench -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/rocprof_bench_kernel_stats.csv && head -12 "$f" | cut -c1-170
rm -rf $O/prof_bench
for TAIL in 0 1; do
( cd /tmp && export TMPDIR=/tmp && timeout -k 5 200 env TAIL=$TAIL rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_flow -- python $R/tools/probe_flow.py profile > $R/$O/prof_flow_tail$TAIL.log 2>&1; echo "== rocprof flow tail=$TAIL rc=$?" )
f=$(find $O/prof_flow -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/rocprof_flow_tail${TAIL}_kernel_stats.csv && head -10 "$f" | cut -c1-170
rm -rf $O/prof_flow
( cd /tmp && export TMPDIR=/tmp && timeout -k 5 200 env TAIL=$TAIL rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/$O/pmc_flow_sq -- python $R/tools/probe_flow.py profile > $R/$O/pmc_flow_sq.log 2>&1; echo "== pmc flow sq rc=$?" )
( cd /tmp && export TMPDIR=/tmp && timeout -k 5 200 env TAIL=$TAIL rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_flow_fetch -- python $R/tools/probe_flow.py profile > $R/$O/pmc_flow_fetch.log 2>&1; echo "== pmc flow fetch rc=$?" )
python tools/pmc_summary.py $O/pmc_flow_tail$TAIL.json $O/pmc_flow_sq $O/pmc_flow_fetch -- flow_tail flow_gemm attn_flow gemm_conv norm_rows | head -40
rm -rf $O/pmc_flow_sq $O/pmc_flow_fetch
done
