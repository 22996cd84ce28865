#!/bin/bash
# Flow / HiFT A/B call: attention key splits, block tail at batch 1 / 4 / 8, HiFT kernel stats with the two-sided split on.
set -u
TAG=${1:-r3f}
O=gpurun_out/$TAG; mkdir -p $O
R=$GRAFT_REPO_ROOT
run() { local name=$1; shift; local t0=$(date +%s); echo "== $name"; timeout -k 5 "$@" > $O/$name.log 2>&1; echo "   rc=$? $(( $(date +%s) - t0 ))s ($(tail -1 $O/$name.log | cut -c1-160))"; }
run probe_flow_attn 200 python tools/probe_flow.py attn
grep -E "ms per flow|max \|" $O/probe_flow_attn.log
run probe_flow_batch_tail 300 python tools/probe_flow_batch.py tail
grep -E "nu=|flow pass" $O/probe_flow_batch_tail.log
( cd /tmp && export TMPDIR=/tmp && timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_hift -- python $R/tools/profile_small.py hift > $R/$O/prof_hift.log 2>&1; echo "== rocprof hift rc=$?" )
f=$(find $O/prof_hift -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/rocprof_hift_wx3_kernel_stats.csv && head -14 "$f" | cut -c1-200
rm -rf $O/prof_hift
( cd /tmp && export TMPDIR=/tmp && CV_GEMM_WX3=0 timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_hift0 -- python $R/tools/profile_small.py hift > $R/$O/prof_hift0.log 2>&1; echo "== rocprof hift (fp32 chain) rc=$?" )
f=$(find $O/prof_hift0 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/rocprof_hift_chain_kernel_stats.csv && head -14 "$f" | cut -c1-200
rm -rf $O/prof_hift0
