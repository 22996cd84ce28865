#!/bin/bash
# Round 4, eleventh call: CosyVoice-300M's LM decode step as one library call (cv_lm1_step: 73 launches in one hipGraph) against the launch-per-operator tape.
set -u
O=gpurun_out/r4k; mkdir -p $O
R=$GRAFT_REPO_ROOT
run() { local name=$1; shift; local t0=$(date +%s); echo "== $name"; timeout -k 5 "$@" > $O/$name.log 2>&1; echo "   rc=$? $(( $(date +%s) - t0 ))s ($(tail -1 $O/$name.log | cut -c1-300))"; }
run pytest_cv1 600 python -m pytest tests/test_zzz_cosyvoice1_hip.py -q -m gpu -p no:cacheprovider -x
run probe_fused 300 python tools/probe_cv1.py
grep -E "^LM|^flow|^hift|^one|^host" $O/probe_fused.log
run probe_nograph 300 python tools/probe_cv1.py nograph
grep -E "^LM" $O/probe_nograph.log
run probe_nofused 300 python tools/probe_cv1.py nofused
grep -E "^LM" $O/probe_nofused.log
( cd /tmp && export TMPDIR=/tmp && timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_cv1 -- python $R/tools/probe_cv1.py profile > $R/$O/prof_cv1.log 2>&1; echo "== rocprof cv1 rc=$? $(tail -1 $R/$O/prof_cv1.log)" )
f=$(find $O/prof_cv1 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/rocprof_cv1_kernel_stats.csv && head -14 "$f" | cut -c1-200
rm -rf $O/prof_cv1
run bench_cv1 400 python bench.py --only-extra cosyvoice300m --steps 8
