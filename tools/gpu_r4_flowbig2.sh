#!/bin/bash
# Round 4, third call: persistent large-M GEMMs - sweep, per-kernel rocprof of the best few, SQ counters of attention / GEMM at 8 utterances per pass.
set -u
O=gpurun_out/r4c; mkdir -p $O
R=$GRAFT_REPO_ROOT
run() { local name=$1; shift; local t0=$(date +%s); echo "== $name"; timeout -k 5 "$@" > $O/$name.log 2>&1; echo "   rc=$? $(( $(date +%s) - t0 ))s ($(tail -1 $O/$name.log | cut -c1-160))"; }
run pytest_flow_big 200 python -m pytest tests/test_flow.py -q -m gpu -p no:cacheprovider -x -k "big_m"
run probe_big2 400 python tools/probe_flow_big2.py
grep -E "nu=|bit-identical" $O/probe_big2.log
for cfg in 0,3,3 0,2,2 0,1,2 1,1,1 2,2,2; do
  ( cd /tmp && export TMPDIR=/tmp && timeout -k 5 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$cfg -- python $R/tools/probe_flow_big2.py profile cfg=$cfg > $R/$O/prof_$cfg.log 2>&1; echo "== rocprof cfg=$cfg rc=$? $(grep 'nu=' $R/$O/prof_$cfg.log | tail -1)" )
  f=$(find $O/prof_$cfg -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/rocprof_big_${cfg}_kernel_stats.csv && grep -E "flow_gemm_big|attn_flow|ln_bf16" "$f" | cut -c1-160
  rm -rf $O/prof_$cfg
done
# SQ counters (own pass, kernel-trace only besides --pmc): where the waves of the attention / GEMM kernels spend their cycles
( cd /tmp && export TMPDIR=/tmp && timeout -k 5 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $R/$O/pmc_sq -- python $R/tools/probe_flow_big2.py profile cfg=0,3,3 > $R/$O/pmc_sq.log 2>&1; echo "== pmc sq rc=$?" )
python tools/pmc_summary.py $O/pmc_sq_big.json $O/pmc_sq -- flow_gemm_big attn_flow ln_bf16 gemm_conv norm_rows > $O/pmc_sq_big.txt 2>&1; tail -60 $O/pmc_sq_big.txt
rm -rf $O/pmc_sq
( cd /tmp && export TMPDIR=/tmp && timeout -k 5 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU --output-format csv -d $R/$O/pmc_sq2 -- python $R/tools/probe_flow_big2.py profile cfg=0,3,3 > $R/$O/pmc_sq2.log 2>&1; echo "== pmc sq2 rc=$?" )
python tools/pmc_summary.py $O/pmc_sq2_big.json $O/pmc_sq2 -- flow_gemm_big attn_flow ln_bf16 > $O/pmc_sq2_big.txt 2>&1; tail -40 $O/pmc_sq2_big.txt
rm -rf $O/pmc_sq2
