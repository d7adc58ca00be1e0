"""Per-kernel averages of the SQ counters of one rocprofv3 --pmc pass (diagnostics): python scripts/pmc_small_report.py <dir>"""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void mi::", "").replace("mi::", "")[:40]
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    if k.startswith("void at") or k.startswith("__amd"):
        continue
    print("%-42s n=%3d " % (k, len(next(iter(v.values())))) + "  ".join("%s=%.3g" % (c.replace("SQ_", ""), sum(x) / len(x)) for c, x in sorted(v.items())))
