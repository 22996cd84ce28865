#!/bin/bash
# Round 4, fourth call: hardware check of what the session added after the flow kernels (RAS replay, padded-mask B3, single-row GEMV under CosyVoice-300M, fp8 sub-line)
# and the A/B runs that decide defaults: flow_batch 4 | 8 with the large-M kernels, persistent flow GEMMs next to the LM (overlap), LDS-DMA staging.
set -u
O=gpurun_out/r4d; mkdir -p $O
R=$GRAFT_REPO_ROOT
run() { local name=$1; shift; local t0=$(date +%s); echo "== $name"; timeout -k 5 "$@" > $O/$name.log 2>&1; echo "   rc=$? $(( $(date +%s) - t0 ))s ($(tail -1 $O/$name.log | cut -c1-200))"; }
show() { python - "$O/$1.log" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        d = json.loads(line)
        print("    value", d["value"], "ms", d["ms_per_step"], "| batched", {k: d["batched_decode"].get(k) for k in ("audio_s_per_s", "pipeline_audio_s_per_s", "lm_us_per_step")} if d.get("batched_decode") else None,
              "| ras", (d.get("self_check") or {}).get("ras"))
PY
}
run pytest_new 500 python -m pytest tests/test_flow.py tests/test_ops.py tests/test_zz_fullsize.py tests/test_zzz_cosyvoice1_hip.py tests/test_zzz_cosyvoice1_hip_model.py tests/test_zzz_round3_hooks.py -q -m gpu -p no:cacheprovider -x -k "padded_mask or big_m or single_row or ras_replay or cosyvoice1 or forward_one_step or f0_float64 or cv3_causal_hift"
run probe_cv1 200 python tools/probe_cv1.py
grep -E "LM:|flow:|one 500" $O/probe_cv1.log
CV_GEMV_F32=0 run probe_cv1_tiled 200 python tools/probe_cv1.py
grep -E "LM:" $O/probe_cv1_tiled.log
B="python bench.py --no-extras --steps 4 --warmup 1 --no-cpu-baseline --first-chunk-reps 1"
run b16_fb4 200 $B --batch 16 --flow-batch 4; show b16_fb4
run b16_fb8 200 $B --batch 16 --flow-batch 8; show b16_fb8
CV_FLOW_BIG_PERSIST=3 run b16_fb8_persist3 200 $B --batch 16 --flow-batch 8; show b16_fb8_persist3
CV_FLOW_BIG_PERSIST=2 run b16_fb8_persist2 200 $B --batch 16 --flow-batch 8; show b16_fb8_persist2
CV_FLOW_BIG_ROWS=0 run b16_fb8_small 200 $B --batch 16 --flow-batch 8; show b16_fb8_small
run mixed_fb4 200 $B --workload mixed64 --flow-batch 4; show mixed_fb4
run mixed_fb8 200 $B --workload mixed64 --flow-batch 8; show mixed_fb8
CV_FLOW_BIG_PERSIST=3 run mixed_fb8_persist3 200 $B --workload mixed64 --flow-batch 8; show mixed_fb8_persist3
run probe_glds 200 python tools/probe_flow_big2.py cfg=-1,3,3
CV_FLOW_BIG_GLDS=1 run probe_glds_on 200 python tools/probe_flow_big2.py cfg=-1,3,3
CV_FLOW_BIG_GLDS=1 run probe_glds_on_128 200 python tools/probe_flow_big2.py cfg=0,1,2
grep -h "nu=" $O/probe_glds.log $O/probe_glds_on.log $O/probe_glds_on_128.log
run extra_cv3 400 python bench.py --only-extra cosyvoice3 --steps 4
tail -2 $O/extra_cv3.log | cut -c1-1500
run extra_cv1 400 python bench.py --only-extra cosyvoice300m --steps 4
tail -2 $O/extra_cv1.log | cut -c1-1200
