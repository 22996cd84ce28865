# Round-2 validation run on one MI355X: full GPU test-suite, smoke, the bench lines (headline, serving extras, mixed64), rocprof kernel stats and the
# FETCH_SIZE pass behind `roofline.traffic`.  Everything lands in gpurun_out/; the summaries worth keeping are copied to profiles/ afterwards.
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 900 > gpurun_out/r2_tests_final.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_tests_final.log; tail -4 gpurun_out/r2_tests_final.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/r2_bench_final.json 2> gpurun_out/r2_bench_final.err; echo "bench rc=$?"; tail -4 gpurun_out/r2_bench_final.err
timeout 900 python bench.py --steps 6 --warmup 2 --batch 8 --lanes 4 --stream-clients 8 --stream-requests 104 --no-cpu-baseline --first-chunk-reps 3 > gpurun_out/r2_bench_serving.json 2> gpurun_out/r2_bench_serving.err; echo "serving rc=$?"
timeout 900 python bench.py --steps 2 --warmup 1 --batch 16 --lanes 4 --no-cpu-baseline --first-chunk-reps 1 > gpurun_out/r2_bench_b16.json 2> gpurun_out/r2_bench_b16.err; echo "b16 rc=$?"
timeout 900 python bench.py --workload mixed64 --steps 2 --warmup 1 --lanes 4 --no-cpu-baseline --first-chunk-reps 1 > gpurun_out/r2_bench_mixed64.json 2> gpurun_out/r2_bench_mixed64.err; echo "mixed rc=$?"
python - <<'PY'
import json
for f in ("r2_bench_final", "r2_bench_serving", "r2_bench_b16", "r2_bench_mixed64"):
    try:
        d = json.loads([l for l in open("gpurun_out/%s.json" % f) if l.startswith("{")][-1])
        print(f, {k: d.get(k) for k in ("value", "ms_per_step", "first_chunk_ms_p50", "batched_decode", "streaming_clients", "utterance_hashes_sha1")})
        r = d["roofline"]; print("   roofline", {k: r.get(k) for k in ("achieved", "frac", "avg_launch_us", "traffic", "decode_stage")})
        if "cpu_baseline" in d: print("   cpu", {k: d["cpu_baseline"].get(k) for k in ("value", "cores", "kind")})
    except Exception as e:
        print(f, "unreadable", e)
PY
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --first-chunk-reps 1 > $R/gpurun_out/r2_prof_bench.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_llm -- python $R/tools/profile_small.py llm > $R/gpurun_out/r2_pmc_llm.log 2>&1
cd $R
f=$(find gpurun_out/prof_bench -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r2_rocprof_bench_final_kernel_stats.csv; head -8 "$f" | cut -c1-150
python tools/pmc_summary.py gpurun_out/r2_pmc_gemv_fetch.json gpurun_out/pmc_llm -- gemv attn_decode
rm -rf gpurun_out/prof_bench gpurun_out/pmc_llm
python - <<'PY'
import json
d = json.load(open("gpurun_out/r2_pmc_gemv_fetch.json"))
for k, v in d.items(): print(k[:70], v.get("n"), round(v.get("hbm_read_bytes_corrected", 0) / 1e6, 2), "MB")
PY
