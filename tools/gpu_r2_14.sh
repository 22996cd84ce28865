mkdir -p gpurun_out
for rt in 2 3 2 3; do
CV_SKINNY_WIDE_RT=$rt timeout 600 python bench.py --steps 2 --warmup 1 --batch 8 --lanes 4 --no-cpu-baseline --first-chunk-reps 1 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('wide_rt=$rt', d.get('batched_decode'))"
done | tee gpurun_out/r2_skinny_rt_ab.txt
