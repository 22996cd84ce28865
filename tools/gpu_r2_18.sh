mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_hift -- python $R/tools/profile_small.py hift > $R/gpurun_out/r2_prof_hift.log 2>&1
cd $R
f=$(find gpurun_out/prof_hift -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r2_rocprof_hift_kernel_stats.csv; head -30 "$f" | cut -c1-170
t=$(find gpurun_out/prof_hift -name "*kernel_trace.csv" | head -1); python - "$t" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if "gemm_conv" in r["Kernel_Name"]]
print(len(rows), "gemm launches")
for r in rows[:80]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print("%8.1f us grid %s x %s x %s  %s" % (d, r.get("Grid_Size_X", r.get("Grid_Size")), r.get("Grid_Size_Y"), r.get("Grid_Size_Z"), r["Kernel_Name"][17:75]))
PY
rm -rf gpurun_out/prof_hift
