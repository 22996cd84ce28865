set -u
O=gpurun_out/r3x; mkdir -p $O
S="python bench.py --no-extras --steps 1 --warmup 1 --no-cpu-baseline --first-chunk-reps 1 --lanes 4"
timeout 200 $S --stream-clients 8 > $O/s_alone.log 2>&1
timeout 300 $S --stream-clients 8 --batch 16 > $O/s_after_b16.log 2>&1
timeout 300 $S --stream-clients 8 --batch 8 > $O/s_after_b8.log 2>&1
for f in s_alone s_after_b16 s_after_b8; do python - $O/$f.log $f <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        d = json.loads(line); print(sys.argv[2], d.get("streaming_clients"))
PY
done
