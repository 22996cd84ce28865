mkdir -p gpurun_out
timeout 500 python -m pytest tests/test_ops.py tests/test_llm.py tests/test_flow.py tests/test_dit.py tests/test_zz_llm_batch.py -m gpu -q -p no:cacheprovider --timeout 400 2>&1 | tail -3
timeout 200 python tools/probe_prefill_x3.py 2>&1 | grep -v Warn | tail -6 | tee gpurun_out/r2_prefill_x3_ab.txt
for v in 0 1 0 1; do
CV_GEMM_X3=$v timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --first-chunk-reps 5 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('CV_GEMM_X3=$v', d['value'], 'audio_s/s', d['ms_per_step'], 'ms/utt first chunk', d['first_chunk_ms_p50'], 'tokens ok', d['self_check']['tokens_equal_oracle'])"
done | tee -a gpurun_out/r2_prefill_x3_ab.txt
