# Last validation of the round on the final tree: full GPU suite, the padded-pass probe, the default bench line.
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 900 > gpurun_out/r2_tests_final.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_tests_final.log; tail -4 gpurun_out/r2_tests_final.log
timeout 200 python tools/probe_flow_batch.py ragged 2>&1 | grep "^tokens" | tee gpurun_out/r2_flow_ragged.txt
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r2_bench_final2.json 2> gpurun_out/r2_bench_final2.err; echo "bench rc=$?"; python -c "
import json; d = json.loads([l for l in open('gpurun_out/r2_bench_final2.json') if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['first_chunk_ms_p50'], d['self_check']['tokens_equal_oracle'], d['stages'])"
