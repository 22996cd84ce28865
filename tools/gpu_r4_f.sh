#!/bin/bash
# Round 4, sixth call: row-wise (LDS-staged) output stores of the large-M GEMMs A/B + per-kernel times; 32 lock-step slots on the hardware.
set -u
O=gpurun_out/r4f; mkdir -p $O
R=$GRAFT_REPO_ROOT
run() { local name=$1; shift; local t0=$(date +%s); echo "== $name"; timeout -k 5 "$@" > $O/$name.log 2>&1; echo "   rc=$? $(( $(date +%s) - t0 ))s ($(tail -1 $O/$name.log | cut -c1-200))"; }
show() { python - "$O/$1.log" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        d = json.loads(line)
        if "extra" in d:
            r = d["result"]; print("    ", {k: r.get(k) for k in ("audio_s_per_s", "wall_s", "pipeline_audio_s_per_s", "lm_us_per_step", "tokens_equal_oracle_all_slots", "utterance_hashes_sha1")}); continue
        print("    value", d["value"], "ms", d["ms_per_step"], "| batched", {k: d["batched_decode"].get(k) for k in ("audio_s_per_s", "pipeline_audio_s_per_s", "lm_us_per_step", "tokens_equal_oracle_all_slots")} if d.get("batched_decode") else None)
PY
}
run pytest_new 400 python -m pytest tests/test_flow.py tests/test_zz_llm_batch.py -q -m gpu -p no:cacheprovider -x -k "big_m or sixteen or batch_matches or batch_of_eight or continuous"
for epi in 1 0; do
  for cfg in -1,3,3 -1,2,2 -1,1,2; do CV_FLOW_BIG_LDS_EPI=$epi run probe_epi${epi}_$cfg 100 python tools/probe_flow_big2.py cfg=$cfg; done
done
grep -h "nu=" $O/probe_epi1_*.log | sed 's/^/lds-epilogue  /'; grep -h "nu=" $O/probe_epi0_*.log | sed 's/^/lane stores   /'
( cd /tmp && export TMPDIR=/tmp && timeout -k 5 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_epi -- python $R/tools/probe_flow_big2.py profile cfg=-1,3,3 > $R/$O/prof_epi.log 2>&1; echo "== rocprof lds epilogue rc=$?" )
f=$(find $O/prof_epi -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/rocprof_big_ldsepi_kernel_stats.csv && grep -E "flow_gemm_big|attn_flow|ln_bf16|norm_rows|cvt_bf16" "$f" | cut -c1-170
rm -rf $O/prof_epi
B="python bench.py --no-extras --steps 4 --warmup 1 --no-cpu-baseline --first-chunk-reps 1"
run b16 200 $B --batch 16; show b16
run b32 300 $B --batch 32; show b32
run b32_l3 300 $B --batch 32 --lanes 3; show b32_l3
E="python bench.py --steps 20"
run mixed_16slots 300 $E --only-extra mixed64; show mixed_16slots
CV_BENCH_MIXED_SLOTS=32 run mixed_32slots 300 $E --only-extra mixed64; show mixed_32slots
CV_BENCH_MIXED_SLOTS=32 run mixed_32slots_l3 300 $E --only-extra mixed64 --lanes 3; show mixed_32slots_l3
