#!/bin/bash
# Dev tool (round 6): tools/ubench/attn_probe, then rocprofv3 PMC passes (own runs, --kernel-trace only) over one case / variant.
#   gpurun -- 'bash tools/gpu_attn_pmc.sh <tag> <case> "<variant substring>"'
TAG=$1; CASE=${2:-0}; VAR=${3:-"<4,3,3> 128"}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O
$R/tools/ubench/attn_probe > $O/attn_probe.log 2>&1; cat $O/attn_probe.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters.txt 2>&1
pass() { local n=$1; shift; timeout -k 5 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc_$n -- $R/tools/ubench/attn_probe $CASE "$VAR" > $O/pmc_$n.log 2>&1; echo "== pmc $n rc=$?";
  f=$(find $O/pmc_$n -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python3 $R/tools/pmc_avg.py "$f" attn_flow32 | tee $O/pmc_$n.txt; rm -rf $O/pmc_$n; }
pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE
pass b SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM
pass c SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_BF16
