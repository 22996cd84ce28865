mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_zz_fullsize.py -m gpu -q -p no:cacheprovider --timeout 600 -k cv3 2>&1 | tail -12
for st in 4 10; do
timeout 900 python bench.py --steps 3 --warmup 1 --cv3 --cv3-steps $st --lanes 4 --no-cpu-baseline --first-chunk-reps 1 > gpurun_out/r2_bench_cv3_$st.json 2> gpurun_out/r2_bench_cv3_$st.err; echo "rc=$?"; tail -2 gpurun_out/r2_bench_cv3_$st.err | cut -c1-400
done
python - <<'PY'
import json
for st in (4, 10):
    try:
        d = json.loads([l for l in open("gpurun_out/r2_bench_cv3_%d.json" % st) if l.startswith("{")][-1]); print(st, d.get("cosyvoice3"))
    except Exception as e: print(st, "unreadable", e)
PY
cat gpurun_out/r2_fullsize_errors.json | grep cv3
