#!/bin/bash
# Round 4, first call: per-kernel rocprof + PMC of the batched flow pass at 1 / 4 / 8 utterances, CosyVoice-300M stage times (first hardware timing).
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/gpu_r4_first.sh'
set -u
O=gpurun_out/r4a; mkdir -p $O
R=$GRAFT_REPO_ROOT
run() { local name=$1; shift; local t0=$(date +%s); echo "== $name"; timeout -k 5 "$@" > $O/$name.log 2>&1; echo "   rc=$? $(( $(date +%s) - t0 ))s ($(tail -1 $O/$name.log | cut -c1-160))"; }
for nu in 1 4 8; do
  ( cd /tmp && export TMPDIR=/tmp && timeout -k 5 150 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_fb$nu -- python $R/tools/profile_flow_batch.py $nu > $R/$O/prof_fb$nu.log 2>&1; echo "== rocprof flow batch $nu rc=$? $(tail -1 $R/$O/prof_fb$nu.log)" )
  f=$(find $O/prof_fb$nu -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/rocprof_flow_batch${nu}_kernel_stats.csv && head -14 "$f" | cut -c1-220
  rm -rf $O/prof_fb$nu
done
for nu in 1 8; do
  ( cd /tmp && export TMPDIR=/tmp && timeout -k 5 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/$O/pmc_a$nu -- python $R/tools/profile_flow_batch.py $nu > $R/$O/pmc_a$nu.log 2>&1; echo "== pmc mfma $nu rc=$?" )
  ( cd /tmp && export TMPDIR=/tmp && timeout -k 5 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE WRITE_SIZE --output-format csv -d $R/$O/pmc_b$nu -- python $R/tools/profile_flow_batch.py $nu > $R/$O/pmc_b$nu.log 2>&1; echo "== pmc fetch $nu rc=$?" )
  python tools/pmc_summary.py $O/pmc_flow_batch$nu.json $O/pmc_a$nu $O/pmc_b$nu > $O/pmc_flow_batch$nu.txt 2>&1
  rm -rf $O/pmc_a$nu $O/pmc_b$nu
done
run probe_cv1             300 python tools/probe_cv1.py
grep -E "LM:|flow|HiFT|host" $O/probe_cv1.log | head -20
run probe_cv1_graphs      300 python tools/probe_cv1.py graphs
grep -E "LM:|flow|HiFT|host" $O/probe_cv1_graphs.log | head -20
