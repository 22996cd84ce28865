mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_serving.py tests/test_frontend.py tests/test_model.py tests/test_flow.py -m gpu -q -p no:cacheprovider --timeout 600 2>&1 | tail -8
timeout 900 python bench.py --steps 10 --warmup 3 --stream-clients 8 --stream-requests 104 > gpurun_out/r2_bench_2.json 2> gpurun_out/r2_bench_2.err; echo "bench rc=$?"
tail -6 gpurun_out/r2_bench_2.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r2_bench_2.json") if l.startswith("{")][-1])
print({k: d[k] for k in ("value", "ms_per_step", "first_chunk_ms_p50", "self_check")})
print(d.get("streaming_clients")); print(d["cpu_baseline"])
PY
