mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_model.py tests/test_serving.py -m gpu -q -p no:cacheprovider --timeout 300 2>&1 | tail -3
run() {
  timeout 300 python bench.py --steps 1 --warmup 1 --stream-clients 8 --stream-requests 56 --lanes $2 --no-cpu-baseline --first-chunk-reps 1 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); b = d['streaming_clients']; print('clients 8 first-chunk-priority=$1 lanes=$2', json.dumps(b))"
}
( CV_FIRST_CHUNK_PRIORITY=0 run 0 4; CV_FIRST_CHUNK_PRIORITY=1 run 1 4; CV_FIRST_CHUNK_PRIORITY=1 run 1 3; CV_FIRST_CHUNK_PRIORITY=1 run 1 6 ) | tee gpurun_out/r2_first_chunk_priority_ab.txt
