# Round-2 validation run on one MI355X: full GPU test-suite, smoke, the bench lines (headline with cpu_baseline, serving extras, mixed64) and the rocprof
# kernel stats of the headline command.  Everything lands in gpurun_out/; the summaries worth keeping are copied to profiles/ afterwards.
# (The FETCH_SIZE pass behind `roofline.traffic` is tools/profile_small.py llm under rocprofv3 --pmc, summarised by tools/pmc_summary.py:
#  profiles/r2_pmc_gemv_fetch.json - the gate/up kernel has not changed since.)
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 900 > gpurun_out/r2_tests_final.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_tests_final.log; tail -4 gpurun_out/r2_tests_final.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
timeout 600 python bench.py > gpurun_out/r2_bench_final.json 2> gpurun_out/r2_bench_final.err; echo "bench rc=$?"; tail -3 gpurun_out/r2_bench_final.err
timeout 400 python bench.py --steps 4 --warmup 2 --batch 8 --lanes 4 --stream-clients 8 --stream-requests 104 --no-cpu-baseline --first-chunk-reps 3 > gpurun_out/r2_bench_serving.json 2> gpurun_out/r2_bench_serving.err; echo "serving rc=$?"
python - <<'PY'
import json
for f in ("r2_bench_final", "r2_bench_serving"):
    try:
        d = json.loads([l for l in open("gpurun_out/%s.json" % f) if l.startswith("{")][-1])
        print(f, {k: d.get(k) for k in ("value", "ms_per_step", "first_chunk_ms_p50", "stages", "batched_decode", "streaming_clients")})
        r = d["roofline"]; print("   roofline", {k: r.get(k) for k in ("achieved", "frac", "avg_launch_us", "traffic", "decode_stage")})
        if "cpu_baseline" in d: print("   cpu", {k: d["cpu_baseline"].get(k) for k in ("value", "cores", "kind")})
    except Exception as e:
        print(f, "unreadable", e)
PY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --first-chunk-reps 1 > $R/gpurun_out/r2_prof_bench.log 2>&1
cd $R
f=$(find gpurun_out/prof_bench -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r2_rocprof_bench_final_kernel_stats.csv; head -8 "$f" | cut -c1-150
rm -rf gpurun_out/prof_bench
timeout 300 python bench.py --workload mixed64 --steps 2 --warmup 1 --lanes 3 --no-cpu-baseline --first-chunk-reps 1 > gpurun_out/r2_bench_mixed64.json 2> gpurun_out/r2_bench_mixed64.err; echo "mixed rc=$?"
python - <<'PY'
import json
try:
    d = json.loads([l for l in open("gpurun_out/r2_bench_mixed64.json") if l.startswith("{")][-1])
    print("mixed64", {k: d.get(k) for k in ("value", "ms_per_step", "utterance_hashes_sha1")})
except Exception as e:
    print("mixed64 unreadable", e)
PY
