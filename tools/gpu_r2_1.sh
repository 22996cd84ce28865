set -x
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 900 > gpurun_out/r2_tests_1.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_tests_1.log
tail -30 gpurun_out/r2_tests_1.log
timeout 600 python bench.py --steps 10 --warmup 3 --batch 8 > gpurun_out/r2_bench_1.json 2> gpurun_out/r2_bench_1.err; echo "bench rc=$?"
tail -5 gpurun_out/r2_bench_1.err; cat gpurun_out/r2_bench_1.json | head -c 3000
