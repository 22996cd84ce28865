#!/bin/bash
# Round 4, tenth call: the conformer encoder over the stacked utterances of a pass (enc_batch) A/B, the float64 f0 default's cost (CosyVoice3 extra), batch extras.
set -u
O=gpurun_out/r4j; mkdir -p $O
run() { local name=$1; shift; local t0=$(date +%s); echo "== $name"; timeout -k 5 "$@" > $O/$name.log 2>&1; echo "   rc=$? $(( $(date +%s) - t0 ))s ($(tail -1 $O/$name.log | cut -c1-300))"; }
run pytest_flow 400 python -m pytest tests/test_flow.py -q -m gpu -p no:cacheprovider -x
for e in 0 1; do
  CV_FLOW_ENC_BATCH=$e run probe_enc$e 100 python tools/probe_flow_big2.py cfg=-1,3,3
  grep "nu=" $O/probe_enc$e.log
done
for f in 0 1; do
  CV_BENCH_CV3_F0_F64=$f run cv3_f0_$f 300 python bench.py --only-extra cosyvoice3 --steps 8
done
run batch16 300 python bench.py --only-extra batched_decode_16 --steps 8
run mixed64 300 python bench.py --only-extra mixed64 --steps 20
