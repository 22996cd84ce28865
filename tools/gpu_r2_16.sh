mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --first-chunk-reps 1 > gpurun_out/r2_torchrun_w1.json 2> gpurun_out/r2_torchrun_w1.err; echo "torchrun u10 rc=$?"; tail -2 gpurun_out/r2_torchrun_w1.err | cut -c1-300
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --workload mixed64 --steps 1 --warmup 0 --no-cpu-baseline --first-chunk-reps 1 > gpurun_out/r2_torchrun_mixed_w1.json 2> gpurun_out/r2_torchrun_mixed_w1.err; echo "torchrun mixed rc=$?"; tail -2 gpurun_out/r2_torchrun_mixed_w1.err | cut -c1-300
python - <<'PY'
import json
for f in ("r2_torchrun_w1", "r2_torchrun_mixed_w1"):
    try:
        d = json.loads([l for l in open("gpurun_out/%s.json" % f) if l.startswith("{")][-1]); print(f, d["n_gpus"], d["value"], d["scaling"], d.get("utterance_hashes_sha1"))
    except Exception as e: print(f, "unreadable", e)
PY
