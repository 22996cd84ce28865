#!/bin/bash
# Round 4, twelfth and thirteenth calls: cv_lm1_step after the one-pass attention (8 key ranges per head, rows and query in one round trip), LayerNorm row requested before the weights,
# logits through a pinned buffer.
set -u
O=gpurun_out/r4m; mkdir -p $O
R=$GRAFT_REPO_ROOT
run() { local name=$1; shift; local t0=$(date +%s); echo "== $name"; timeout -k 5 "$@" > $O/$name.log 2>&1; echo "   rc=$? $(( $(date +%s) - t0 ))s ($(tail -1 $O/$name.log | cut -c1-300))"; }
run pytest_cv1 600 python -m pytest tests/test_zzz_cosyvoice1_hip.py -q -m gpu -p no:cacheprovider -x
run probe_fused 300 python tools/probe_cv1.py
grep -E "^LM|^one" $O/probe_fused.log
run probe_nograph 300 python tools/probe_cv1.py nograph
grep -E "^LM" $O/probe_nograph.log
( cd /tmp && export TMPDIR=/tmp && timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_cv1 -- python $R/tools/probe_cv1.py profile nograph > $R/$O/prof_cv1.log 2>&1; echo "== rocprof cv1 rc=$? $(tail -1 $R/$O/prof_cv1.log)" )
f=$(find $O/prof_cv1 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/rocprof_cv1_kernel_stats.csv && grep lm1 "$f" | cut -c1-200
f=$(find $O/prof_cv1 -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
# per (kernel, grid) durations of the lm1 kernels + the gaps between consecutive launches of one decode step
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "lm1_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
by = collections.defaultdict(list)
for r in rows:
    by[(r["Kernel_Name"].split("(")[0][-28:], r["Grid_Size_X"], r.get("Grid_Size_Y", ""))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(by.items()):
    v.sort(); print("%-30s grid %6s x %2s  n=%5d  median %6.2f us  p10 %6.2f  p90 %6.2f" % (k[0], k[1], k[2], len(v), v[len(v)//2] / 1e3, v[len(v)//10] / 1e3, v[len(v)*9//10] / 1e3))
gaps = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(rows, rows[1:])]
gaps = [g for g in gaps if g < 20000]
gaps.sort(); print("gaps between consecutive lm1 kernels (< 20 us): n=%d median %.2f us p90 %.2f us, sum per 73 launches %.1f us" % (len(gaps), gaps[len(gaps)//2] / 1e3, gaps[len(gaps)*9//10] / 1e3, sum(gaps) / len(gaps) * 73 / 1e3))
PY
rm -rf $O/prof_cv1
run bench_cv1 400 python bench.py --only-extra cosyvoice300m --steps 8
