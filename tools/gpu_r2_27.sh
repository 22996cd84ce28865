mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_llm_fp8.py tests/test_ops.py -m gpu -q -p no:cacheprovider --timeout 300 2>&1 | tail -3
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_b8 -- python $R/bench.py --steps 2 --warmup 1 --batch 8 --lanes 2 --no-cpu-baseline --first-chunk-reps 1 > $R/gpurun_out/r2_prof_b8.log 2>&1
cd $R
f=$(find gpurun_out/prof_b8 -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r2_rocprof_batch8_kernel_stats.csv; grep -E "skinny|attn_decode_batch|sum_partials" "$f" | cut -c1-150
rm -rf gpurun_out/prof_b8
timeout 300 python bench.py --steps 2 --warmup 1 --cv3 --cv3-steps 4 --no-cpu-baseline --first-chunk-reps 1 > gpurun_out/r2_bench_cosyvoice3.json 2>/dev/null; python -c "
import json; d = json.loads([l for l in open('gpurun_out/r2_bench_cosyvoice3.json') if l.startswith('{')][-1]); print(d['cosyvoice3'])"
timeout 200 python bench.py --steps 2 --warmup 1 --batch 16 --lanes 3 --no-cpu-baseline --first-chunk-reps 1 > gpurun_out/r2_bench_b16.json 2>/dev/null; python -c "
import json; d = json.loads([l for l in open('gpurun_out/r2_bench_b16.json') if l.startswith('{')][-1]); print(d['batched_decode'])"
