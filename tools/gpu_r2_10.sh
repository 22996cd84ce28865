mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_model.py -m gpu -q -p no:cacheprovider --timeout 600 -k "lanes or concurrent or matches_oracle" 2>&1 | tail -2
for ln in 1 2 3 4; do
timeout 600 python bench.py --steps 4 --warmup 2 --batch 8 --lanes $ln --no-cpu-baseline --first-chunk-reps 1 > gpurun_out/r2_bench_b8_l$ln.json 2> gpurun_out/r2_bench_b8_l$ln.err; echo "rc=$?"; tail -1 gpurun_out/r2_bench_b8_l$ln.err
python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/r2_bench_b8_l$ln.json") if l.startswith("{")][-1])
print("lanes $ln", d.get("value"), d.get("batched_decode"))
PY
done | tee gpurun_out/r2_lanes_ab.txt
for ln in 1 3; do
timeout 600 python bench.py --steps 2 --warmup 1 --stream-clients 8 --stream-requests 56 --lanes $ln --no-cpu-baseline --first-chunk-reps 1 > gpurun_out/r2_bench_sc_l$ln.json 2> gpurun_out/r2_bench_sc_l$ln.err; echo "rc=$?"; tail -1 gpurun_out/r2_bench_sc_l$ln.err
python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/r2_bench_sc_l$ln.json") if l.startswith("{")][-1])
print("lanes $ln", d.get("streaming_clients"))
PY
done | tee -a gpurun_out/r2_lanes_ab.txt
timeout 600 python bench.py --steps 2 --warmup 1 --batch 16 --lanes 3 --no-cpu-baseline --first-chunk-reps 1 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('b16 lanes3', d.get('batched_decode'))" | tee -a gpurun_out/r2_lanes_ab.txt
