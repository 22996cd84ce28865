#!/bin/bash
# Round-4 end-of-round validation on one MI355X:  /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/gpu_r4_final.sh [tag]'
# Full -m gpu suite, smoke, the driver's bench command (with its extras), rocprofv3 kernel stats of the bench command, the FETCH_SIZE pass behind roofline.traffic,
# MFMA-busy and FETCH_SIZE passes of the 8-utterance flow pass on the large-M kernel set.
set -u
TAG=${1:-r4z}
O=gpurun_out/$TAG; mkdir -p $O
R=$GRAFT_REPO_ROOT
run() { local name=$1; shift; local t0=$(date +%s); echo "== $name"; timeout -k 5 "$@" > $O/$name.log 2>&1; echo "   rc=$? $(( $(date +%s) - t0 ))s ($(tail -1 $O/$name.log | cut -c1-160))"; }
run pytest_gpu     900 python -X faulthandler -m pytest tests -q -m gpu -p no:cacheprovider --timeout 500
run smoke           90 python -c "import __graft_entry__ as g; g.smoke()"
run bench_driver  1000 python bench.py --gpus 1 --steps 20 --warmup 5
python - "$O/bench_driver.log" "$O/bench_driver.json" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        d = json.loads(line)
        json.dump(d, open(sys.argv[2], "w"), indent=1)
        print("bench", d["value"], d["ms_per_step"], "first chunk", d.get("first_chunk_ms_p50"), d.get("stages"))
        print("   self_check", {k: (v if not isinstance(v, dict) else {kk: vv for kk, vv in v.items() if not isinstance(vv, (list, dict))}) for k, v in d.get("self_check", {}).items()})
        for k in ("batched_decode", "batched_decode_16", "batched_decode_32", "streaming_clients", "mixed64", "cosyvoice3", "cosyvoice300m"):
            print("  ", k, json.dumps(d.get(k))[:700])
        r = d["roofline"]; print("   roofline", {k: r.get(k) for k in ("achieved", "frac", "avg_launch_us", "traffic", "decode_stage", "decode_step_us_from_chains")})
        if "cpu_baseline" in d: print("   cpu", {k: d["cpu_baseline"].get(k) for k in ("value", "cores", "kind", "stage_seconds")})
PY
( cd /tmp && export TMPDIR=/tmp && timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_bench -- python $R/bench.py --gpus 1 --steps 5 --warmup 2 --no-extras --no-cpu-baseline > $R/$O/prof_bench.log 2>&1; echo "== rocprof bench rc=$?" )
f=$(find $O/prof_bench -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/rocprof_bench_kernel_stats.csv && head -12 "$f" | cut -c1-170
rm -rf $O/prof_bench
( cd /tmp && export TMPDIR=/tmp && timeout -k 5 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_llm -- python $R/tools/profile_small.py llm > $R/$O/pmc_llm.log 2>&1; echo "== pmc llm rc=$?" )
python tools/pmc_summary.py $O/pmc_gemv_fetch.json $O/pmc_llm -- gemv | head -12
rm -rf $O/pmc_llm
( cd /tmp && export TMPDIR=/tmp && timeout -k 5 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/$O/pmc_fb8_sq -- python $R/tools/profile_flow_batch.py 8 > $R/$O/pmc_fb8_sq.log 2>&1; echo "== pmc flow batch 8 sq rc=$?" )
( cd /tmp && export TMPDIR=/tmp && timeout -k 5 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_fb8_fetch -- python $R/tools/profile_flow_batch.py 8 > $R/$O/pmc_fb8_fetch.log 2>&1; echo "== pmc flow batch 8 fetch rc=$?" )
python tools/pmc_summary.py $O/pmc_flow_batch8_end.json $O/pmc_fb8_sq $O/pmc_fb8_fetch -- flow_gemm attn_flow ln_bf16 gemm_conv norm_rows | head -40
rm -rf $O/pmc_fb8_sq $O/pmc_fb8_fetch
