mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_zz_llm_batch.py tests/test_frontend.py -m gpu -q -p no:cacheprovider --timeout 300 2>&1 | tail -3
run() {
  CV_SKINNY_X3=$1 CV_DOWN_FUSED_SUM=$2 timeout 300 python bench.py --steps 2 --warmup 1 --batch $3 --lanes 2 --no-cpu-baseline --first-chunk-reps 1 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); q = d['batched_decode']
print('x3=$1 fused_sum=$2 batch $3:', q['audio_s_per_s'], q['pipeline_audio_s_per_s'], 'lm_us_per_step', q['lm_us_per_step'], 'tokens ok', q['tokens_equal_oracle_all_slots'])"
}
( run 0 0 8; run 1 0 8; run 0 1 8; run 1 1 8; run 0 0 8; run 1 1 8; run 0 0 16; run 1 1 16 ) | tee gpurun_out/r2_skinny_x3_ab.txt
timeout 300 python tools/probe_flow_batch.py sweep 2>&1 | grep -v Warn | tail -14 | tee gpurun_out/r2_flow_batch_sweep.txt
