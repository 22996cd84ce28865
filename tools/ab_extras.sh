#!/bin/bash
# A/B of bench extras over environment knobs and token2wav lanes:  bash tools/ab_extras.sh <tag> "<extra> ..." "<ENV=V,ENV=V|-> ..." "<lanes> ..."
# one line per run: the throughput / latency fields of the extra's record; logs under gpurun_out/<tag>/
TAG=$1; O=gpurun_out/$TAG; mkdir -p $O
for x in $2; do for envs in $3; do for l in $4; do
  name=$(echo "${x}_${envs}_lanes${l}" | tr -c 'A-Za-z0-9_\n' '_')
  ( [ "$envs" != "-" ] && for kv in $(echo "$envs" | tr ',' ' '); do export "$kv"; done; timeout -k 5 600 python bench.py --only-extra $x --steps ${STEPS:-4} --lanes $l > $O/$name.log 2>&1 )
  python - "$O/$name.log" "$x" "$envs" "$l" <<'P'
import json, sys
f, x, envs, l = sys.argv[1:5]
rec = None
for line in open(f):
    if line.startswith('{"extra"'):
        rec = json.loads(line)["result"]
keys = ("audio_s_per_s", "pipeline_audio_s_per_s", "ms_per_batch", "lm_us_per_step", "first_chunk_ms_p50", "wall_s", "utterance_hashes_sha1", "tokens_equal_oracle_all_slots", "tokens_equal_oracle_all_requests")
print("%-18s %-28s lanes %s  %s" % (x, envs, l, {k: (rec[k][:10] if isinstance(rec[k], str) else rec[k]) for k in keys if rec and k in rec} if rec else "FAILED (see %s)" % f), flush=True)
P
done; done; done
