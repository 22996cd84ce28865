mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 900 > gpurun_out/r2_tests_2.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_tests_2.log; tail -6 gpurun_out/r2_tests_2.log
timeout 900 python bench.py --steps 10 --warmup 3 --batch 8 --stream-clients 8 --stream-requests 104 > gpurun_out/r2_bench_3.json 2> gpurun_out/r2_bench_3.err; echo "bench rc=$?"; tail -8 gpurun_out/r2_bench_3.err
timeout 600 python bench.py --workload mixed64 --steps 2 --warmup 1 --no-cpu-baseline --first-chunk-reps 1 > gpurun_out/r2_bench_mixed64.json 2> gpurun_out/r2_bench_mixed64.err; echo "mixed rc=$?"; tail -3 gpurun_out/r2_bench_mixed64.err
python - <<'PY'
import json
for f in ("gpurun_out/r2_bench_3.json", "gpurun_out/r2_bench_mixed64.json"):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f, {k: d.get(k) for k in ("value", "ms_per_step", "first_chunk_ms_p50", "batched_decode", "streaming_clients", "utterance_hashes_sha1", "scaling")})
        print("   roofline:", {k: d["roofline"].get(k) for k in ("achieved", "frac", "avg_launch_us", "decode_stage")})
    except Exception as e:
        print(f, "unreadable", e)
PY
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --first-chunk-reps 1 > $R/gpurun_out/r2_prof_bench.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_batch -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --first-chunk-reps 1 --batch 8 > $R/gpurun_out/r2_prof_batch.log 2>&1
cd $R
for n in bench batch; do f=$(find gpurun_out/prof_$n -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r2_rocprof_${n}_kernel_stats.csv; echo "== $n"; head -16 "$f" | cut -c1-140; done
rm -rf gpurun_out/prof_bench gpurun_out/prof_batch
