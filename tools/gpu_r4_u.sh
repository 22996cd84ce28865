#!/bin/bash
# Round 4: the batched decode attention with the heads of a kv group on one XCD (r4u), then in one workgroup (r4v) - LM step at 16 / 32 slots, mixed64, and its kernel stats.
set -u
O=gpurun_out/r4v; mkdir -p $O
R=$GRAFT_REPO_ROOT
run() { local name=$1; shift; local t0=$(date +%s); echo "== $name"; timeout -k 5 "$@" > $O/$name.log 2>&1; echo "   rc=$? $(( $(date +%s) - t0 ))s"; }
run pytest_batch 400 python -m pytest tests/test_zz_llm_batch.py tests/test_llm_fp8.py -q -m gpu -p no:cacheprovider -x
tail -1 $O/pytest_batch.log
for name in batched_decode_16 batched_decode_32 mixed64; do
  run $name 300 python bench.py --only-extra $name --steps 8
  python - "$O/$name.log" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        r = json.loads(line)["result"]; print("  ", {k: r[k] for k in ("audio_s_per_s", "pipeline_audio_s_per_s", "lm_tokens_per_s", "lm_us_per_step", "wall_s", "tokens_equal_oracle_all_slots") if k in r}, r.get("token_check", ""), r.get("utterance_hashes_sha1", "")[:12])
PY
done
( cd /tmp && export TMPDIR=/tmp && timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_mixed64 -- python $R/bench.py --only-extra mixed64 --steps 6 > $R/$O/prof_mixed64.log 2>&1; echo "== rocprof mixed64 rc=$?" )
f=$(find $O/prof_mixed64 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/rocprof_mixed64_kernel_stats.csv && head -8 "$f" | cut -c1-150
rm -rf $O/prof_mixed64
