mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_zz_llm_batch.py -m gpu -q -p no:cacheprovider --timeout 300 2>&1 | tail -2
run() {
  CV_ATTN_BATCH_WAVES=$1 CV_SKINNY_NARROW_WAVES=$2 timeout 300 python bench.py --steps 2 --warmup 1 --batch 8 --lanes 2 --no-cpu-baseline --first-chunk-reps 1 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); q = d['batched_decode']
print('attn_waves=$1 narrow_waves=$2 batch 8:', q['audio_s_per_s'], q['pipeline_audio_s_per_s'], 'lm_us_per_step', q['lm_us_per_step'], 'tokens ok', q['tokens_equal_oracle_all_slots'])"
}
( run 4 4; run 8 8; run 8 4; run 4 8 ) | tee gpurun_out/r2_batch_waves_ab.txt
