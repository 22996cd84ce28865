#!/bin/bash
# Round 4: mixed64 (configs[3] on one GPU) with the requests admitted as listed (fifo) or longest first, 16 and 32 sequences in flight.
set -u
O=gpurun_out/r4s; mkdir -p $O
for slots in 16 32; do for order in fifo longest_first; do
  CV_BENCH_MIXED_SLOTS=$slots CV_BENCH_MIXED_ORDER=$order timeout 300 python bench.py --only-extra mixed64 --steps 20 > $O/mixed_${slots}_$order.log 2>&1
  python - "$O/mixed_${slots}_$order.log" "$slots" "$order" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        r = json.loads(line)["result"]; print("slots", sys.argv[2], sys.argv[3], r["audio_s_per_s"], "wall", r["wall_s"], r["utterance_hashes_sha1"][:12], r["token_check"]["identical_to_oracle"])
PY
done; done
