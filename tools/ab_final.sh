#!/bin/bash
# Last GPU call of the round: validate the committed default on hardware, then A/B the two-wave 16x32 GEMM tile on ONE box.
mkdir -p gpurun_out/v22
timeout -k 5 70 python -X faulthandler -m pytest tests -x -q -m gpu > gpurun_out/v22/pytest.log 2>&1; echo "pytest rc=$?"; tail -1 gpurun_out/v22/pytest.log
timeout -k 5 40 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/v22/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/v22/smoke.log
show() { python -c "
import json,sys; d=json.loads(open(sys.argv[1]).read()); print(sys.argv[2], d['value'], d['ms_per_step'], d.get('first_chunk_ms_p50'))" "$1" "$2"; }
CV_GEMM_TWO_WAVE_TILE=0 timeout -k 5 40 python bench.py --no-cpu-baseline --first-chunk-reps 1 --steps 4 > gpurun_out/v22/bench_A1.json 2> gpurun_out/v22/bench_A1.err; show gpurun_out/v22/bench_A1.json A1
timeout -k 5 90 python bench.py > gpurun_out/v22/bench_B_full.json 2> gpurun_out/v22/bench_B_full.err; show gpurun_out/v22/bench_B_full.json B_full
CV_GEMM_TWO_WAVE_TILE=0 timeout -k 5 40 python bench.py --no-cpu-baseline --first-chunk-reps 1 --steps 4 > gpurun_out/v22/bench_A2.json 2> gpurun_out/v22/bench_A2.err; show gpurun_out/v22/bench_A2.json A2
timeout -k 5 40 python bench.py --no-cpu-baseline --first-chunk-reps 1 --steps 4 > gpurun_out/v22/bench_B2.json 2> gpurun_out/v22/bench_B2.err; show gpurun_out/v22/bench_B2.json B2
