"""Dev tool: GPU idle gaps in a rocprofv3 --kernel-trace CSV (kernel_trace.csv: Kernel_Name, Start_Timestamp, End_Timestamp in ns).
    python tools/trace_gaps.py <dir or csv> [min_gap_us=30] [last_ms=400]
Prints every gap between consecutive dispatches (all streams merged, sorted by start) above the threshold inside the last `last_ms` of the trace,
with the kernels on either side, and the sum of gaps / busy time there."""
import csv, glob, os, sys

src = sys.argv[1]
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
last_ms = float(sys.argv[3]) if len(sys.argv) > 3 else 400.0
files = [src] if src.endswith(".csv") else glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
if not rows:
    sys.exit("no dispatches found")
t_end = max(r[1] for r in rows)
rows = [r for r in rows if r[0] >= t_end - last_ms * 1e6]
print("%d dispatches in the last %.0f ms" % (len(rows), last_ms))
busy_end = rows[0][1]
gaps, total_gap = [], 0
for i in range(1, len(rows)):
    s, e, name = rows[i]
    if s > busy_end:
        g = (s - busy_end) / 1e3
        total_gap += s - busy_end
        if g >= thr:
            gaps.append((g, (busy_end - rows[0][0]) / 1e6, rows[i - 1][2][:60], name[:60]))
    busy_end = max(busy_end, e)
span = (busy_end - rows[0][0]) / 1e6
print("span %.2f ms, idle %.2f ms in all gaps (%.1f %%)" % (span, total_gap / 1e6, 100 * total_gap / 1e6 / span))
small = total_gap / 1e3 - sum(g[0] for g in gaps)
print("gaps >= %.0f us: %d, %.2f ms in total; smaller gaps: %.2f ms" % (thr, len(gaps), sum(g[0] for g in gaps) / 1e3, small / 1e3))
for g, at, a, b in gaps[:80]:
    print("  %8.1f us at +%8.2f ms   after %-60s before %s" % (g, at, a, b))
