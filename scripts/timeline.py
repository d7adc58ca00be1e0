"""Prints the per-dispatch timeline of the last query batch from a rocprofv3 kernel trace CSV."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "ingest" in r["Kernel_Name"]][-1]
t0 = int(rows[idx]["Start_Timestamp"])
tot = {}
for r in rows[idx:]:
    n = r["Kernel_Name"].split("(")[0].replace("void mi::", "").replace("mi::", "")
    if n.startswith("__amd"): continue
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot[n] = tot.get(n, 0) + d
    print("%-36s start=%8.1f us dur=%8.1f us" % (n[:36], (int(r["Start_Timestamp"]) - t0) / 1e3, d))
print("---- per kernel:", {k: round(v, 1) for k, v in tot.items()}, "sum=%.1f us" % sum(tot.values()))
