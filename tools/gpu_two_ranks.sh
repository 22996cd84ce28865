#!/bin/bash
# Dev tool (round 6): bench.py's N > 1 branch on the hardware that is available - TWO ranks, launched exactly as the driver launches N > 1 (torch.distributed.run, rendezvous
# on 127.0.0.1), both pinned to the box's ONE GPU through an explicit device list (CV_BENCH_RANK_DEVICES=0,0: rank i takes the i-th entry, as it does with a HIP_VISIBLE_DEVICES list the launcher restricts).  The value of such a line
# means nothing (two replicas share a GPU); what it shows is that the real rank body - pinning before the first HIP call, gloo control plane, barrier, max-over-ranks, hash
# gather, the JSON line - executes at full size on an MI355X.   gpurun -- 'bash tools/gpu_two_ranks.sh <tag>'
TAG=${1:-two}; O=gpurun_out/$TAG; mkdir -p $O
for wl in u10 mixed64; do
  CV_BENCH_RANK_DEVICES=0,0 timeout -k 5 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 --workload $wl --no-cpu-baseline > $O/two_ranks_$wl.json 2> $O/two_ranks_$wl.log
  echo "== two ranks $wl rc=$? $(cut -c1-700 $O/two_ranks_$wl.json)"
done
timeout -k 5 600 python bench.py --gpus 1 --steps 3 --warmup 1 --workload mixed64 --no-cpu-baseline --no-extras > $O/one_rank_mixed64.json 2> $O/one_rank_mixed64.log
echo "== one rank mixed64 rc=$? $(cut -c1-400 $O/one_rank_mixed64.json)"
python - <<PY
import json
a = json.load(open("$O/two_ranks_mixed64.json")); b = json.load(open("$O/one_rank_mixed64.json"))
print("== hash over the 64 utterances: two ranks %s  one rank %s  equal %s" % (a["utterance_hashes_sha1"][:12], b["utterance_hashes_sha1"][:12], a["utterance_hashes_sha1"] == b["utterance_hashes_sha1"]))
print("== rank devices:", a["rank_devices"])
PY
