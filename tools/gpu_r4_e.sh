#!/bin/bash
# Round 4, fifth call: LDS-DMA staging through inline asm (hand-placed waits), graph policy for large passes on mixed64, batched HiFT A/B.
set -u
O=gpurun_out/r4e; mkdir -p $O
R=$GRAFT_REPO_ROOT
run() { local name=$1; shift; local t0=$(date +%s); echo "== $name"; timeout -k 5 "$@" > $O/$name.log 2>&1; echo "   rc=$? $(( $(date +%s) - t0 ))s ($(tail -1 $O/$name.log | cut -c1-200))"; }
show() { python - "$O/$1.log" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        d = json.loads(line)
        if "extra" in d:
            r = d["result"]; print("    ", {k: r.get(k) for k in ("audio_s_per_s", "wall_s", "pipeline_audio_s_per_s", "lm_us_per_step", "utterance_hashes_sha1")}); continue
        print("    value", d["value"], "ms", d["ms_per_step"], "| batched", {k: d["batched_decode"].get(k) for k in ("audio_s_per_s", "pipeline_audio_s_per_s", "lm_us_per_step")} if d.get("batched_decode") else None)
PY
}
CV_FLOW_BIG_GLDS=1 run pytest_glds 300 python -m pytest tests/test_flow.py tests/test_zz_fullsize.py -q -m gpu -p no:cacheprovider -x -k "big_m or ras_replay or ragged or batch_equals"
for cfg in -1,3,3 -1,2,2 -1,1,2 0,1,2 0,1,1 2,1,1; do
  run probe_reg_$cfg 100 python tools/probe_flow_big2.py cfg=$cfg
  CV_FLOW_BIG_GLDS=1 run probe_glds_$cfg 100 python tools/probe_flow_big2.py cfg=$cfg
done
grep -h "nu=" $O/probe_reg_*.log | sed 's/^/reg  /'; grep -h "nu=" $O/probe_glds_*.log | sed 's/^/glds /'
( cd /tmp && export TMPDIR=/tmp && CV_FLOW_BIG_GLDS=1 timeout -k 5 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_glds -- python $R/tools/probe_flow_big2.py profile cfg=0,1,2 > $R/$O/prof_glds.log 2>&1; echo "== rocprof glds rc=$?" )
f=$(find $O/prof_glds -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/rocprof_big_glds_0,1,2_kernel_stats.csv && grep -E "flow_gemm_big|attn_flow|ln_bf16" "$f" | cut -c1-170
rm -rf $O/prof_glds
E="python bench.py --steps 20"
run mixed_default 300 $E --only-extra mixed64; show mixed_default
CV_FLOW_GRAPH_MAX_ROWS=0 run mixed_graph_all 300 $E --only-extra mixed64; show mixed_graph_all
CV_FLOW_GRAPH_MAX_ROWS=0 CV_FLOW_BIG_ROWS=0 run mixed_r3_like 300 $E --only-extra mixed64; show mixed_r3_like
run mixed_fb8 300 $E --only-extra mixed64 --flow-batch 8; show mixed_fb8
B="python bench.py --no-extras --steps 4 --warmup 1 --no-cpu-baseline --first-chunk-reps 1"
run b16_fb8 200 $B --batch 16 --flow-batch 8; show b16_fb8
run b16_fb8_hiftbatch 200 $B --batch 16 --flow-batch 8 --hift-batch; show b16_fb8_hiftbatch
