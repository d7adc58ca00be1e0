"""Fabric traffic of the gallery ingest kernels from the rocprofv3 --pmc passes of scripts/profile_record.sh stage 8.
gfx950 corrections (MI355X_MICROARCH.md, HBM): FETCH_SIZE and WRITE_SIZE are in KiB; FETCH_SIZE counts a wide coalesced read
stream at half its bytes (x 2); WRITE_SIZE is exact for 16-byte-per-lane stores.  Usage: python scripts/ingest_pmc_report.py <tag>"""
import csv, glob, os, sys
tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
alg = {"read": 1005994 * 2048 * 4, "write": 1005994 * 2048 * 4 + 1006080 * 2048 * 2 + 1006080 * 12}
for layout in ("rows", "cols"):
    vals = {}
    for kind, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        fs = glob.glob(f"{root}/gpurun_out/{tag}_ingest_{layout}_{kind}/*/*counter_collection.csv")
        per, name = [], ""
        for f in fs:
            for r in csv.DictReader(open(f)):
                if "ingest_" in r["Kernel_Name"] and r["Counter_Name"] == counter:
                    per.append(float(r["Counter_Value"]))
                    name = r["Kernel_Name"].split("(")[0].replace("void mi::", "")[:60]
        if per:
            vals[kind] = (sum(per) / len(per), len(per), name)
    if "fetch" in vals and "write" in vals:
        rd, wr = vals["fetch"][0] * 1024 * 2, vals["write"][0] * 1024
        print("%s: %s  launches %d: read %.3f GB (algorithmic %.3f: x %.3f)  written %.3f GB (algorithmic %.3f: x %.3f)  total %.2f GB vs 20.6"
              % (layout, vals["fetch"][2], vals["fetch"][1], rd / 1e9, alg["read"] / 1e9, rd / alg["read"], wr / 1e9, alg["write"] / 1e9,
                 wr / alg["write"], (rd + wr) / 1e9))
