#!/bin/bash
# GPU idle gaps of the headline hot path (rocprofv3 kernel trace -> tools/trace_gaps.py).
set -u
TAG=${1:-r3g}
O=gpurun_out/$TAG; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout 200 python tools/profile_utt.py > $O/utt_plain.log 2>&1; tail -1 $O/utt_plain.log
( cd /tmp && export TMPDIR=/tmp && timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace -- \
    python $R/tools/profile_utt.py > $R/$O/trace_bench.log 2>&1; echo "== rocprof trace rc=$?" )
tail -1 $O/trace_bench.log
find $O/trace -name "*.csv" | head; f=$(find $O/trace -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && head -2 "$f" | cut -c1-400
python tools/trace_gaps.py $O/trace 30 420 > $O/gaps.txt 2>&1; head -150 $O/gaps.txt
rm -rf $O/trace
