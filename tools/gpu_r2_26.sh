mkdir -p gpurun_out
timeout 500 python -m pytest tests/test_dit.py tests/test_model.py -m gpu -q -p no:cacheprovider --timeout 400 2>&1 | tail -2
run() {
  timeout 400 python bench.py --steps 2 --warmup 1 --cv3 --cv3-steps 4 --flow-batch $1 --lanes $2 --no-cpu-baseline --first-chunk-reps 1 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); q = d['cosyvoice3']
print('cv3 4-step flow_batch=$1 lanes=$2:', 'alone', q['batch1_audio_s_per_s'], '16 per GPU', q['batch16_audio_s_per_s'], 'audio_s/s', q['batch16_ms_per_batch'], 'ms')"
}
( run 1 3; run 4 3; run 4 2 ) | tee gpurun_out/r2_cv3_flow_batch_ab.txt
