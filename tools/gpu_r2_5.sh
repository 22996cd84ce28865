mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_llm.py tests/test_zz_llm_batch.py tests/test_zz_fullsize.py tests/test_model.py -m gpu -q -p no:cacheprovider --timeout 600 2>&1 | tail -5
for cfg in "CV_DECODE_FUSED_QKV=1 CV_GEMV_SHARED_NORM=1" "CV_DECODE_FUSED_QKV=0 CV_GEMV_SHARED_NORM=1" "CV_DECODE_FUSED_QKV=0 CV_GEMV_SHARED_NORM=0"; do
  echo "== $cfg"
  env $cfg timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --first-chunk-reps 3 2> gpurun_out/r2_bench_ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], d['ms_per_step'], d['first_chunk_ms_p50'], r['decode_stage'], {k:v.get('chain_avg_us') for k,v in r['per_kernel'].items()})"
done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_llm -- python $GRAFT_REPO_ROOT/tools/profile_small.py llm > $GRAFT_REPO_ROOT/gpurun_out/r2_pmc_llm.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py gpurun_out/pmc_llm gpurun_out/r2_pmc_gemv_fetch.json gemv qkv_attn
rm -rf gpurun_out/pmc_llm
