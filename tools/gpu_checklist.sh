#!/bin/bash
# One gpurun call for the start of a round: validates the tree on hardware and collects the reference numbers of every bench leg
# (round-2 end values in brackets):
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/gpu_checklist.sh'
# Writes under gpurun_out/checklist/.  Order: cheapest / most important first; every step has its own timeout.  ~7 GPU-minutes.
set -u
O=gpurun_out/checklist; mkdir -p $O
run() { local name=$1; shift; echo "== $name"; timeout -k 5 "$@" > $O/$name.log 2>&1; echo "   rc=$? ($(tail -1 $O/$name.log | cut -c1-150))"; }
run pytest_gpu            400 python -X faulthandler -m pytest tests -q -m gpu                         # [166 passed, 2 skipped, ~3 min]
run smoke                  60 python -c "import __graft_entry__ as g; g.smoke()"
run bench_default         240 python bench.py                                                          # [54.0 audio-s/s, first chunk 66.6 ms, gate/up 0.397]
run bench_serving         240 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --first-chunk-reps 3 --batch 8 --lanes 4 --stream-clients 8 --stream-requests 104   # [189 / 203; 105 audio-s/s, p50 176 ms]
run bench_batch16         200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --first-chunk-reps 1 --batch 16 --lanes 3    # [251 / 271]
run bench_mixed64         240 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --first-chunk-reps 1 --workload mixed64 --lanes 3   # [249]
run bench_cv3             240 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --first-chunk-reps 1 --cv3 --cv3-steps 4     # [55 alone, 278 at 16]
run probe_flow_ragged     120 python tools/probe_flow_batch.py ragged                                  # [x1.5 - x2.3, bit-identical]
run probe_prefill_x3      120 python tools/probe_prefill_x3.py                                         # [4.36 -> 3.07 ms]
( cd /tmp && export TMPDIR=/tmp && timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/prof_bench -- \
    python $OLDPWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline --first-chunk-reps 1 > $OLDPWD/$O/prof_bench.log 2>&1; echo "== rocprof bench rc=$?" )
for f in bench_default bench_serving bench_batch16 bench_mixed64 bench_cv3; do python - "$O/$f.log" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        d = json.loads(line)
        print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], d.get("first_chunk_ms_p50"), d.get("batched_decode"), d.get("streaming_clients"), d.get("cosyvoice3"))
PY
done
