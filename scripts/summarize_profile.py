"""Summarises a scripts/profile_round.sh run into profiles/<tag>_*.  Usage: python scripts/summarize_profile.py <tag>"""
import collections, csv, glob, json, os, shutil, sys
tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
go, pr = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles")
os.makedirs(pr, exist_ok=True)
shutil.copy(glob.glob(f"{go}/{tag}_trace/*/*kernel_stats.csv")[0], f"{pr}/{tag}_kernel_stats.csv")
bench = [l for l in open(f"{go}/{tag}_bench.json") if l.startswith("{")][-1]
open(f"{pr}/{tag}_bench.json", "w").write(bench)
out = {"bench": json.loads(bench)}
def counters(sub):
    agg = collections.defaultdict(list)
    for f in glob.glob(f"{go}/{tag}_{sub}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            agg[(r["Kernel_Name"].split("(")[0].replace("void mi::", "").replace("mi::", ""), r["Counter_Name"])].append(float(r["Counter_Value"]))
    return agg
pm = {}
for sub in ("pmc_fetch", "pmc_write", "pmc_l2", "pmc_sq"):
    for (k, c), v in counters(sub).items():
        pm.setdefault(k, {})[c] = {"launches": len(v), "sum": sum(v), "max": max(v)}
out["pmc"] = pm
# HBM traffic of the dominant kernel per launch (largest launch = the final chunk), gfx950 corrections from
# MI355X_MICROARCH.md: FETCH_SIZE is in KiB and reads exactly half of a wide coalesced stream -> x2; WRITE_SIZE exact.
g = pm.get("gemm_select_kernel<false, 0>", {})
if "FETCH_SIZE" in g:
    n = g["FETCH_SIZE"]["launches"]
    fetch = g["FETCH_SIZE"]["sum"] * 1024 * 2 / n
    write = g.get("WRITE_SIZE", {"sum": 0})["sum"] * 1024 / max(1, g.get("WRITE_SIZE", {"launches": 1})["launches"])
    out["gemm_select_hbm_bytes_per_launch"] = fetch + write
    out["gemm_select_hbm_bytes_per_batch"] = (fetch + write) * 3      # three filtered launches per batch
    if "TCC_HIT_sum" in g:
        out["gemm_select_l2_hit_rate"] = g["TCC_HIT_sum"]["sum"] / (g["TCC_HIT_sum"]["sum"] + g["TCC_MISS_sum"]["sum"])
json.dump(out, open(f"{pr}/{tag}_pmc_summary.json", "w"), indent=1)
json.dump({"gemm_select_hbm_bytes_per_launch": out.get("gemm_select_hbm_bytes_per_launch"), "source": f"profiles/{tag}_pmc_summary.json"},
          open(f"{pr}/pmc_traffic.json", "w"))
for r in csv.DictReader(open(f"{pr}/{tag}_kernel_stats.csv")):
    print("%-44s calls=%4s avg=%9.1f us  %5s%%" % (r["Name"].split("(")[0].replace("void mi::", "").replace("mi::", "")[:44], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
print({k: v for k, v in out.items() if k.startswith("gemm_select")})
for k, v in pm.items():
    if "gemm_select_kernel<false" in k or "rescore" in k:
        print(k, {c: ("%.4g" % (x["sum"] / x["launches"])) for c, x in v.items()})
