#!/bin/bash
# Round 4, seventh call: the depth-4 register ring of the large-M GEMMs.
set -u
O=gpurun_out/r4g; mkdir -p $O
R=$GRAFT_REPO_ROOT
run() { local name=$1; shift; local t0=$(date +%s); echo "== $name"; timeout -k 5 "$@" > $O/$name.log 2>&1; echo "   rc=$? $(( $(date +%s) - t0 ))s ($(tail -1 $O/$name.log | cut -c1-200))"; }
run pytest_big 300 python -m pytest tests/test_flow.py -q -m gpu -p no:cacheprovider -x -k "big_m or ragged or batch_equals"
run probe_sweep 400 python tools/probe_flow_big2.py
grep -E "nu=|bit-identical" $O/probe_sweep.log
for cfg in -1,3,3 -1,2,2 0,3,3; do
  ( cd /tmp && export TMPDIR=/tmp && timeout -k 5 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$cfg -- python $R/tools/probe_flow_big2.py profile cfg=$cfg > $R/$O/prof_$cfg.log 2>&1; echo "== rocprof cfg=$cfg rc=$? $(grep 'nu=' $R/$O/prof_$cfg.log | tail -1)" )
  f=$(find $O/prof_$cfg -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/rocprof_big_ring_${cfg}_kernel_stats.csv && grep -E "flow_gemm_big|attn_flow|ln_bf16" "$f" | cut -c1-170
  rm -rf $O/prof_$cfg
done
