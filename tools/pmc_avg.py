"""Dev tool: per-kernel averages of a rocprofv3 counter_collection.csv (one row per dispatch and counter).  usage: pmc_avg.py <csv> <kernel substring>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
sel = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r.get("Kernel_Name", "")
    if sel in k:
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k[:110])
    for c, v in sorted(cs.items()):
        print("   %-36s n=%4d  avg %16.1f" % (c, len(v), sum(v) / len(v)))
