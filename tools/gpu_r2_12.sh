mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_zz_llm_batch.py tests/test_model.py tests/test_cosyvoice1.py tests/test_serving.py -m gpu -q -p no:cacheprovider --timeout 600 2>&1 | tail -3
for w in 4 5 4 5; do
CV_GEMV_GATEUP_WAVES=$w timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --first-chunk-reps 7 > gpurun_out/r2_gw$w.json 2> gpurun_out/r2_gw$w.err
python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/r2_gw$w.json") if l.startswith("{")][-1])
pk = d["roofline"]["per_kernel"]
print("gate_up waves=$w", d.get("value"), d.get("ms_per_step"), "first chunk", d.get("first_chunk_ms_p50"), "gate_up chain us", [v["chain_avg_us"] for k, v in pk.items() if "gate_up" in k], d["roofline"].get("decode_stage", {}).get("us_per_token_from_chains"))
PY
done | tee gpurun_out/r2_gateup_waves_ab.txt
timeout 600 python bench.py --steps 4 --warmup 2 --batch 8 --lanes 4 --stream-clients 8 --stream-requests 104 --no-cpu-baseline --first-chunk-reps 1 > gpurun_out/r2_bench_b8_l4.json 2> gpurun_out/r2_bench_b8_l4.err; echo "rc=$?"; tail -1 gpurun_out/r2_bench_b8_l4.err
python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/r2_bench_b8_l4.json") if l.startswith("{")][-1])
print("b8 lanes 4 (batched prefill)", d.get("batched_decode")); print("8 clients lanes 4 (first-chunk lane)", d.get("streaming_clients"))
PY
