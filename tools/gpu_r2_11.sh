mkdir -p gpurun_out
for ln in 3 6; do
timeout 600 python bench.py --steps 2 --warmup 1 --stream-clients 8 --stream-requests 56 --batch 8 --lanes $ln --no-cpu-baseline --first-chunk-reps 1 > gpurun_out/r2_bench_sc_l$ln.json 2> gpurun_out/r2_bench_sc_l$ln.err; echo "rc=$?"; tail -1 gpurun_out/r2_bench_sc_l$ln.err
python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/r2_bench_sc_l$ln.json") if l.startswith("{")][-1])
print("prio lanes $ln", d.get("streaming_clients"), d.get("batched_decode"))
PY
done | tee gpurun_out/r2_lanes_prio_ab.txt
