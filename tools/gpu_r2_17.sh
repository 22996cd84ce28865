mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_llm_fp8.py tests/test_zz_llm_batch.py tests/test_hift.py -m gpu -q -p no:cacheprovider --timeout 600 2>&1 | tail -4
for nb in 8 16; do
timeout 600 python bench.py --steps 2 --warmup 1 --batch $nb --lanes 4 --llm-fp8 --no-cpu-baseline --first-chunk-reps 1 2>gpurun_out/r2_fp8_b$nb.err | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('fp8 batch $nb', d.get('batched_decode'))"
done | tee gpurun_out/r2_fp8_ab.txt
timeout 900 python bench.py --steps 2 --warmup 1 --cv3 --cv3-steps 4 --llm-fp8 --lanes 4 --no-cpu-baseline --first-chunk-reps 1 2>gpurun_out/r2_fp8_cv3.err | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('fp8 cv3', d.get('cosyvoice3'))" | tee -a gpurun_out/r2_fp8_ab.txt
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_fp8 -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --first-chunk-reps 1 --batch 8 --lanes 1 --llm-fp8 > $R/gpurun_out/r2_prof_fp8.log 2>&1
cd $R
f=$(find gpurun_out/prof_fp8 -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r2_rocprof_batch8_fp8_kernel_stats.csv; grep "skinny\|attn_decode_batch\|sum_partials" "$f" | cut -c1-150
rm -rf gpurun_out/prof_fp8
tail -3 gpurun_out/r2_fp8_b8.err
