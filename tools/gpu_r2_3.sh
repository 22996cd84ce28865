set -x
mkdir -p gpurun_out
timeout 300 python tools/probe_flow.py all 2>&1 | tee gpurun_out/r2_probe_flow_1.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_flow_fused -- python $GRAFT_REPO_ROOT/tools/probe_flow.py profile > $GRAFT_REPO_ROOT/gpurun_out/r2_prof_flow.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_flow_fused -name "*kernel_stats*" | head; f=$(find gpurun_out/prof_flow_fused -name "*kernel_stats.csv" | head -1); head -14 "$f" | cut -c1-170
timeout 600 python -m pytest tests/test_flow.py -m gpu -q -p no:cacheprovider -k "fused or bf16" 2>&1 | tail -5
