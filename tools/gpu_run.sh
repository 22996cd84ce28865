#!/bin/bash
# One parametrised GPU call (round 5; replaces the per-measurement gpu_r3_* / gpu_r4_* scripts):
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/gpu_run.sh <tag> <plan> [<plan> ...]'
# Every plan writes its logs under gpurun_out/<tag>/ and prints one summary line per step.  Plans:
#   tests [pytest args]   the -m gpu suite (or the files / -k expression given in PYTEST_ARGS)
#   smoke                 __graft_entry__.smoke()
#   bench                 the driver's bench command (all extras)
#   extra:<name>[:ENV=V,ENV=V...]   one bench extra in a process of its own with A/B knobs in its environment
#   prof:<name>           rocprofv3 --kernel-trace --stats of one bench extra -> kernel stats csv
#   profbench             rocprofv3 --kernel-trace --stats of the bench command without extras
#   pmcgemv / pmcflow     the FETCH_SIZE / MFMA-busy passes behind the roofline records
#   py:<script>[:args[:ENV=V,...]]      python tools/<script> args  (a probe; args separated by spaces inside the quoted plan)
#   profpy:<script>[:args[:ENV=V,...]]  the same under rocprofv3 --kernel-trace --stats
#   bin:<path>            a microbenchmark binary (tools/ubench/*)
set -u
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
STEPS=${STEPS:-4}
run() { local name=$1; shift; local t0=$(date +%s); timeout -k 5 "$@" > $O/$name.log 2>&1; echo "== $name rc=$? $(( $(date +%s) - t0 ))s  $(grep -v '^\s*$' $O/$name.log | tail -1 | cut -c1-600)"; }
prof() { local name=$1; shift; ( cd /tmp && export TMPDIR=/tmp && timeout -k 5 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$name -- "$@" > $R/$O/prof_$name.log 2>&1; echo "== rocprof $name rc=$?" )
  local f=$(find $O/prof_$name -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/rocprof_${name}_kernel_stats.csv && head -${PROF_LINES:-14} "$f" | cut -c1-200; rm -rf $O/prof_$name; }
for plan in "$@"; do
  case "$plan" in
    tests)     run pytest_gpu 1200 python -X faulthandler -m pytest ${PYTEST_ARGS:-tests} -q -m gpu -p no:cacheprovider --timeout 500 ;;
    smoke)     run smoke 120 python -c "import __graft_entry__ as g; g.smoke()" ;;
    bench)     run bench_driver 1200 python bench.py --gpus 1 --steps 20 --warmup 5
               python tools/bench_summary.py $O/bench_driver.log $O/bench_driver.json ;;
    extra:*)   IFS=: read -r _ name envs <<< "$plan"; envs=${envs#:}; tagname=$(echo "${name}_${envs:-default}" | tr -c 'A-Za-z0-9_\n' '_')
               ( for kv in $(echo "${envs:-}" | tr ',' ' '); do export "$kv"; done; run extra_$tagname 900 python bench.py --only-extra $name --steps $STEPS ) ;;
    prof:*)    IFS=: read -r _ name envs <<< "$plan"; envs=${envs#:}; ( for kv in $(echo "${envs:-}" | tr ',' ' '); do export "$kv"; done; prof $name python $R/bench.py --only-extra $name --steps 2 ) ;;
    profbench) prof bench python $R/bench.py --gpus 1 --steps 5 --warmup 2 --no-extras --no-cpu-baseline ;;
    pmcgemv)   ( cd /tmp && export TMPDIR=/tmp && timeout -k 5 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_llm -- python $R/tools/profile_small.py llm > $R/$O/pmc_llm.log 2>&1; echo "== pmc llm rc=$?" )
               python tools/pmc_summary.py $O/pmc_gemv_fetch.json $O/pmc_llm -- gemv | head -12; rm -rf $O/pmc_llm ;;
    pmcflow)   ( cd /tmp && export TMPDIR=/tmp && timeout -k 5 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/$O/pmc_fb8_sq -- python $R/tools/profile_flow_batch.py 8 > $R/$O/pmc_fb8_sq.log 2>&1; echo "== pmc flow batch 8 sq rc=$?" )
               ( cd /tmp && export TMPDIR=/tmp && timeout -k 5 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_fb8_fetch -- python $R/tools/profile_flow_batch.py 8 > $R/$O/pmc_fb8_fetch.log 2>&1; echo "== pmc flow batch 8 fetch rc=$?" )
               python tools/pmc_summary.py $O/pmc_flow_batch8.json $O/pmc_fb8_sq $O/pmc_fb8_fetch -- flow_gemm flow_band flow_lnqkv attn_flow ln_bf16 gemm_conv norm_rows | head -40; rm -rf $O/pmc_fb8_sq $O/pmc_fb8_fetch ;;
    bin:*)     IFS=: read -r _ exe <<< "$plan"; run bin_$(basename $exe) 600 $exe; cat $O/bin_$(basename $exe).log ;;
    profpy:*)  IFS=: read -r _ script pargs envs <<< "$plan"; envs=${envs#:}; ( for kv in $(echo "${envs:-}" | tr ',' ' '); do export "$kv"; done; prof $(basename $script .py)_$(echo "${pargs:-}_${envs:-}" | tr -c 'A-Za-z0-9_\n' '_') python $R/tools/$script ${pargs:-} ) ;;
    py:*)      IFS=: read -r _ script pargs envs <<< "$plan"; envs=${envs#:}; ( for kv in $(echo "${envs:-}" | tr ',' ' '); do export "$kv"; done; run py_$(basename $script .py)_$(echo "${pargs:-}_${envs:-}" | tr -c 'A-Za-z0-9_\n' '_') 900 python tools/$script ${pargs:-} ) ;;
    *)         echo "unknown plan $plan" ;;
  esac
done
