#!/bin/bash
# Round 4, last profiles: rocprofv3 kernel stats of the throughput configurations end to end - 16 requests per lock-step batch (configs[2]-style) and the 64-utterance mixed
# workload (configs[3] on one GPU) - so that the next round's kernel work on them starts from per-kernel evidence.
set -u
O=gpurun_out/r4t; mkdir -p $O
R=$GRAFT_REPO_ROOT
for name in batched_decode_16 mixed64; do
  ( cd /tmp && export TMPDIR=/tmp && timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$name -- python $R/bench.py --only-extra $name --steps 6 > $R/$O/prof_$name.log 2>&1; echo "== rocprof $name rc=$? $(grep -c . $R/$O/prof_$name.log) lines" )
  f=$(find $O/prof_$name -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/rocprof_${name}_kernel_stats.csv && head -16 "$f" | cut -c1-150
  rm -rf $O/prof_$name
done
