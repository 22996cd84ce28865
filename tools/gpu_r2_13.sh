mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_zz_llm_batch.py -m gpu -q -p no:cacheprovider --timeout 600 2>&1 | tail -2
for nb in 8 16; do
timeout 600 python bench.py --steps 4 --warmup 2 --batch $nb --lanes 4 --no-cpu-baseline --first-chunk-reps 1 > gpurun_out/r2_bench_b$nb.json 2> gpurun_out/r2_bench_b$nb.err; echo "bench b$nb rc=$?"; tail -1 gpurun_out/r2_bench_b$nb.err
python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/r2_bench_b$nb.json") if l.startswith("{")][-1])
print($nb, d.get("value"), d.get("batched_decode"))
PY
done
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_batch -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --first-chunk-reps 1 --batch 8 --lanes 1 > $R/gpurun_out/r2_prof_batch2.log 2>&1
cd $R
f=$(find gpurun_out/prof_batch -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r2_rocprof_batch8_mfma_kernel_stats.csv; grep "skinny\|attn_decode_batch\|sum_partials\|sample_kernel" "$f" | cut -c1-150
rm -rf gpurun_out/prof_batch
