"""Prints the last N dispatches of a rocprofv3 kernel trace with start / end relative to the first of them (overlaps between
streams become visible).  Usage: python scripts/trace_tail.py <trace dir> [N=40]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if not r["Kernel_Name"].startswith("__amd")][-n:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    nme = r["Kernel_Name"].split("(")[0].replace("void mi::", "").replace("mi::", "")
    print("%-40s q=%-3s start=%9.1f end=%9.1f us" % (nme[:40], r.get("Queue_Id", "?"), (int(r["Start_Timestamp"]) - t0) / 1e3,
                                                     (int(r["End_Timestamp"]) - t0) / 1e3))
