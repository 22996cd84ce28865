#!/bin/bash
# Round-3 A/B of the offline batch paths: sequences per flow pass (--flow-batch 4 | 8) x token2wav lanes, batch 16 and the mixed64 workload.
#   /usr/local/graft/bin/gpurun --timeout 700 -- 'bash tools/gpu_r3_flowbatch.sh <tag>'
set -u
TAG=${1:-r3b}
O=gpurun_out/$TAG; mkdir -p $O
run() { local name=$1; shift; echo "== $name"; timeout -k 5 "$@" > $O/$name.log 2>&1; echo "   rc=$?"; python - "$O/$name.log" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        d = json.loads(line)
        print("   ", d["value"], d.get("batched_decode"), d.get("config", {}).get("workload", "")[:40])
PY
}
B="python bench.py --no-extras --steps 4 --warmup 1 --no-cpu-baseline --first-chunk-reps 1"
run b16_fb4_l3 200 $B --batch 16 --flow-batch 4 --lanes 3
run b16_fb8_l3 200 $B --batch 16 --flow-batch 8 --lanes 3
run b16_fb8_l2 200 $B --batch 16 --flow-batch 8 --lanes 2
run mixed_fb4_l3 200 $B --workload mixed64 --flow-batch 4 --lanes 3
run mixed_fb8_l3 200 $B --workload mixed64 --flow-batch 8 --lanes 3
run mixed_fb8_l2 200 $B --workload mixed64 --flow-batch 8 --lanes 2
