#!/bin/bash
# Batched-decode A/B: fragment-ordered weights + wave-private activation staging (skinny_pk_kernel) against the round-2 row-major kernels.
set -u
TAG=${1:-r3m}
O=gpurun_out/$TAG; mkdir -p $O
run() { local name=$1; shift; local t0=$(date +%s); echo "== $name"; timeout -k 5 "$@" > $O/$name.log 2>&1; echo "   rc=$? $(( $(date +%s) - t0 ))s ($(tail -1 $O/$name.log | cut -c1-200))"; }
run pytest_llm 500 python -X faulthandler -m pytest tests/test_zz_llm_batch.py tests/test_llm.py tests/test_llm_fp8.py tests/test_model_batch.py tests/test_serving.py -q -m gpu -p no:cacheprovider --timeout 400
run pytest_fullsize_llm 300 python -X faulthandler -m pytest tests/test_zz_fullsize.py -q -m gpu -p no:cacheprovider --timeout 280 -k "llm or batch"
for pk in 0 1; do
  run bench_b8_pk$pk 200 env CV_BATCH_PACKED=$pk python bench.py --no-extras --batch 8 --steps 2 --warmup 1 --no-cpu-baseline --first-chunk-reps 1
  run bench_b16_pk$pk 200 env CV_BATCH_PACKED=$pk python bench.py --no-extras --batch 16 --steps 2 --warmup 1 --no-cpu-baseline --first-chunk-reps 1
done
run bench_stream8 300 python bench.py --no-extras --stream-clients 8 --stream-requests 56 --lanes 4 --steps 2 --warmup 1 --no-cpu-baseline --first-chunk-reps 1
for f in bench_b8_pk0 bench_b8_pk1 bench_b16_pk0 bench_b16_pk1 bench_stream8; do python - "$O/$f.log" "$f" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        d = json.loads(line)
        print(sys.argv[2], d["value"], {k: d[k] for k in d if k.startswith("batched") or k.startswith("streaming")})
PY
done
