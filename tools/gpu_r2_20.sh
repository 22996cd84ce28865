mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_flow.py tests/test_model.py tests/test_zz_llm_batch.py -m gpu -q -p no:cacheprovider --timeout 600 2>&1 | tail -3
for cfg in "1 4" "2 4" "4 4" "8 4" "4 2" "8 2" "8 1"; do
set -- $cfg
timeout 600 python bench.py --steps 2 --warmup 1 --batch 8 --flow-batch $1 --lanes $2 --no-cpu-baseline --first-chunk-reps 1 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); b = d['batched_decode']; print('batch 8 flow_batch=$1 lanes=$2', b['audio_s_per_s'], b['pipeline_audio_s_per_s'], b['ms_per_batch'])"
done | tee gpurun_out/r2_flow_batch_ab.txt
timeout 600 python bench.py --steps 2 --warmup 1 --batch 16 --flow-batch 4 --lanes 4 --no-cpu-baseline --first-chunk-reps 1 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); b = d['batched_decode']; print('batch 16 flow_batch=4 lanes=4', b['audio_s_per_s'], b['pipeline_audio_s_per_s'], b['ms_per_batch'])" | tee -a gpurun_out/r2_flow_batch_ab.txt
timeout 600 python tools/probe_flow_batch.py 2>&1 | tail -6 | tee -a gpurun_out/r2_flow_batch_ab.txt
