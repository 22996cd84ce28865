mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_frontend.py tests/test_serving.py -m gpu -q -p no:cacheprovider --timeout 300 2>&1 | tail -3
run() {
  timeout 300 python bench.py --steps 2 --warmup 1 --stream-clients 8 --stream-requests 56 --batch 8 --lanes $2 --lane-cus $1 --no-cpu-baseline --first-chunk-reps 1 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); b = d['streaming_clients']; q = d['batched_decode']
print('lane_cus=$1 lanes=$2 | batch 8:', q['audio_s_per_s'], q['pipeline_audio_s_per_s'], '| clients 8:', json.dumps(b))"
}
( run 0 4; run 224 4; run 192 4; run 160 4; run 192 6 ) | tee gpurun_out/r2_lane_cu_mask_ab.txt
