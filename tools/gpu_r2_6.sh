mkdir -p gpurun_out
timeout 300 python tools/probe_flow.py all 2>&1 | tee gpurun_out/r2_probe_flow_3.txt
timeout 600 python -m pytest tests/test_flow.py -m gpu -q -p no:cacheprovider -k "fused or bf16" 2>&1 | tail -3
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_flow3 -- python $R/tools/probe_flow.py profile > $R/gpurun_out/r2_prof_flow3.log 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_flow_sq -- python $R/tools/probe_flow.py profile > $R/gpurun_out/r2_pmc_flow_sq.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_flow_fetch -- python $R/tools/probe_flow.py profile > $R/gpurun_out/r2_pmc_flow_fetch.log 2>&1
cd $R
python tools/pmc_summary.py gpurun_out/r2_pmc_flow.json gpurun_out/pmc_flow_sq gpurun_out/pmc_flow_fetch -- flow_gemm attn_flow gemm_conv norm_rows attention | head -80
f=$(find gpurun_out/prof_flow3 -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r2_rocprof_flow_fused_v3_kernel_stats.csv; head -10 "$f" | cut -c1-150
tail -3 gpurun_out/r2_pmc_flow_sq.log
rm -rf gpurun_out/pmc_flow_sq gpurun_out/pmc_flow_fetch gpurun_out/prof_flow3
