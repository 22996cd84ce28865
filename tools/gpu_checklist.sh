#!/bin/bash
# One gpurun call for the start of a round: validates the tree on hardware and collects the reference numbers (round-4 end values in brackets; the end-of-round
# validation with the rocprof / PMC passes is tools/gpu_r4_final.sh):
#   /usr/local/graft/bin/gpurun --timeout 2000 -- 'bash tools/gpu_checklist.sh'
# Writes under gpurun_out/checklist/.  Order: cheapest / most important first; every step has its own timeout.  ~20 GPU-minutes (pytest ~5, bench ~7 with the cosyvoice300m extra, the CosyVoice-300M probes ~6).
set -u
O=gpurun_out/checklist; mkdir -p $O
R=$GRAFT_REPO_ROOT
run() { local name=$1; shift; echo "== $name"; timeout -k 5 "$@" > $O/$name.log 2>&1; echo "   rc=$? ($(tail -1 $O/$name.log | cut -c1-150))"; }
run pytest_gpu            800 python -X faulthandler -m pytest tests -q -m gpu -p no:cacheprovider         # [241 passed, 3 skipped, 4-10 min by the box]
run smoke                  60 python -c "import __graft_entry__ as g; g.smoke()"
# the driver's command; the line carries every other BASELINE.json configuration as an extra (each in a process of its own, 2 lanes):
# [56.2 audio-s/s, first chunk 58.4 ms, gate/up 0.39-0.40; 8 streaming clients 173 audio-s/s at p50 117 ms; batch 8 / 16 / 32 226 / 367 / 416 (246 / 384 / 441 overlapped);
#  mixed64 432 (32 in flight, longest first); cosyvoice3 60 / 368, fp8 330; cosyvoice300m 28.3-28.9]
run bench_driver          600 python bench.py --gpus 1 --steps 20 --warmup 5
python - "$O/bench_driver.log" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        d = json.loads(line)
        print("bench", d["value"], d["ms_per_step"], "first chunk", d.get("first_chunk_ms_p50"), d.get("stages"))
        for k in ("streaming_clients", "batched_decode", "batched_decode_16", "batched_decode_32", "mixed64", "cosyvoice3", "cosyvoice300m"):
            print("  ", k, d.get(k))
        r = d["roofline"]; print("   roofline", {k: r.get(k) for k in ("achieved", "frac", "avg_launch_us", "decode_step_us_from_chains")})
PY
# batched vocoding A/B (cv_hift_inference_batch, opt-in since the last session of round 3: bit-identical per utterance, never timed): batch 16 with and without
run bench_b16_hift_batch  300 python bench.py --gpus 1 --steps 6 --warmup 2 --no-extras --no-cpu-baseline --batch 16 --hift-batch
run bench_b16_hift_single 300 python bench.py --gpus 1 --steps 6 --warmup 2 --no-extras --no-cpu-baseline --batch 16
# CosyVoice-300M on the kernels: stage times, host share, fp32 chain vs split3  [LM 0.54 ms per token through cv_lm1_step (nofused: 0.96), flow 77-79 ms, HiFT 8.4 ms]
run probe_cv1             300 python tools/probe_cv1.py
run probe_cv1_split3      300 python tools/probe_cv1.py split3
run probe_cv1_graphs      300 python tools/probe_cv1.py graphs                                      # the estimator tape as a hipGraph (opt-in, never run on hardware yet)
run probe_flow_ragged     120 python tools/probe_flow_batch.py ragged                                  # [x1.5 - x2.3, bit-identical]
( cd /tmp && export TMPDIR=/tmp && timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_utt -- python $R/tools/profile_utt.py > $R/$O/prof_utt.log 2>&1; echo "== rocprof hot path rc=$?" )
f=$(find $O/prof_utt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/rocprof_utt_kernel_stats.csv && head -12 "$f" | cut -c1-170
rm -rf $O/prof_utt
( cd /tmp && export TMPDIR=/tmp && timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_cv1 -- python $R/tools/probe_cv1.py profile > $R/$O/prof_cv1.log 2>&1; echo "== rocprof cosyvoice-300m rc=$?" )
f=$(find $O/prof_cv1 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/rocprof_cv1_kernel_stats.csv && head -12 "$f" | cut -c1-170
rm -rf $O/prof_cv1
[ -f gpurun_out/r3_cv1_fullsize_errors.json ] && cp gpurun_out/r3_cv1_fullsize_errors.json $O/   # measured full-size errors of tests/test_zzzz_cosyvoice1_fullsize.py: tighten its bounds to 3x these
