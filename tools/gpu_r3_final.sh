#!/bin/bash
# Round-3 end-of-round validation on one MI355X:  /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash tools/gpu_r3_final.sh [tag]'
# Full -m gpu suite, smoke, the driver's bench command (with its extras), rocprofv3 kernel stats of the headline hot path.
set -u
TAG=${1:-r3z}
O=gpurun_out/$TAG; mkdir -p $O
R=$GRAFT_REPO_ROOT
run() { local name=$1; shift; local t0=$(date +%s); echo "== $name"; timeout -k 5 "$@" > $O/$name.log 2>&1; echo "   rc=$? $(( $(date +%s) - t0 ))s ($(tail -1 $O/$name.log | cut -c1-160))"; }
run pytest_gpu     700 python -X faulthandler -m pytest tests -q -m gpu -p no:cacheprovider --timeout 500
[ -f gpurun_out/r3_fullsize_errors.json ] && cp gpurun_out/r3_fullsize_errors.json $O/
run smoke           90 python -c "import __graft_entry__ as g; g.smoke()"
run bench_driver   900 python bench.py --gpus 1 --steps 20 --warmup 5
python - "$O/bench_driver.log" "$O/bench_driver.json" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        d = json.loads(line)
        json.dump(d, open(sys.argv[2], "w"), indent=1)
        print("bench", d["value"], d["ms_per_step"], "first chunk", d.get("first_chunk_ms_p50"), d.get("stages"))
        for k in ("batched_decode", "batched_decode_16", "streaming_clients", "mixed64", "cosyvoice3"):
            print("  ", k, d.get(k))
        r = d["roofline"]; print("   roofline", {k: r.get(k) for k in ("achieved", "frac", "avg_launch_us", "traffic", "decode_stage", "decode_step_us_from_chains")})
        print("   per_kernel", {k[:28]: v.get("chain_avg_us") for k, v in r["per_kernel"].items()})
        if "cpu_baseline" in d: print("   cpu", {k: d["cpu_baseline"].get(k) for k in ("value", "cores", "kind", "stage_seconds")})
PY
( cd /tmp && export TMPDIR=/tmp && timeout -k 5 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_utt -- python $R/tools/profile_utt.py > $R/$O/prof_utt.log 2>&1; echo "== rocprof utt rc=$?" )
f=$(find $O/prof_utt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/rocprof_utt_kernel_stats.csv && head -12 "$f" | cut -c1-170
rm -rf $O/prof_utt
