"""Summarises a scripts/profile_round.sh run into profiles/<tag>_*.  Usage: python scripts/summarize_profile.py <tag>"""
import collections, csv, glob, json, os, re, shutil, sys
tag = sys.argv[1]
# the filtered scoring launch of a batch (not the conditional repair instantiation, whose last bool is true)
MAIN = re.compile(r"gemm_(tile|select)_kernel<false, 0, (true|false), false")
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
go, pr = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles")
os.makedirs(pr, exist_ok=True)
newest = lambda pat: max(glob.glob(pat), key=os.path.getmtime)
shutil.copy(newest(f"{go}/{tag}_trace/*/*kernel_stats.csv"), f"{pr}/{tag}_kernel_stats.csv")
bench = [l for l in open(f"{go}/{tag}_bench.json") if l.startswith("{")][-1]
open(f"{pr}/{tag}_bench.json", "w").write(bench)
out = {"bench": json.loads(bench)}
def counters(sub):
    agg = collections.defaultdict(list)
    for f in [newest(f"{go}/{tag}_{sub}/*/*counter_collection.csv")] if glob.glob(f"{go}/{tag}_{sub}/*/*counter_collection.csv") else []:
        for r in csv.DictReader(open(f)):
            agg[(r["Kernel_Name"].split("(")[0].replace("void mi::", "").replace("mi::", ""), r["Counter_Name"])].append(float(r["Counter_Value"]))
    return agg
pm = {}
for sub in ("pmc_fetch", "pmc_write", "pmc_l2", "pmc_sq"):
    for (k, c), v in counters(sub).items():
        pm.setdefault(k, {})[c] = {"launches": len(v), "sum": sum(v), "max": max(v)}
out["pmc"] = pm
# HBM traffic of the dominant kernel, per launch = the big filtered scoring launch of a batch (conditional repair
# launches that exit immediately are ignored).  gfx950 corrections from MI355X_MICROARCH.md: FETCH_SIZE is in KiB and
# reads exactly half of a wide coalesced stream -> x2; WRITE_SIZE (KiB) exact.
def per_launch_values(sub, counter):
    vals = []
    for f in [newest(f"{go}/{tag}_{sub}/*/*counter_collection.csv")] if glob.glob(f"{go}/{tag}_{sub}/*/*counter_collection.csv") else []:
        for r in csv.DictReader(open(f)):
            if MAIN.search(r["Kernel_Name"]) and r["Counter_Name"] == counter:
                vals.append(float(r["Counter_Value"]))
    return vals
fv, wv = per_launch_values("pmc_fetch", "FETCH_SIZE"), per_launch_values("pmc_write", "WRITE_SIZE")
if fv:
    big_f = [v for v in fv if v > 0.5 * max(fv)]
    big_w = [v for v in wv if v > 0.5 * max(wv)] if wv else [0.0]
    out["gemm_select_hbm_bytes_per_launch"] = sum(big_f) / len(big_f) * 1024 * 2 + sum(big_w) / len(big_w) * 1024
    hv, mv = per_launch_values("pmc_l2", "TCC_HIT_sum"), per_launch_values("pmc_l2", "TCC_MISS_sum")
    if hv and mv:
        out["gemm_select_l2_hit_rate"] = sum(hv) / (sum(hv) + sum(mv))
json.dump(out, open(f"{pr}/{tag}_pmc_summary.json", "w"), indent=1)
json.dump({"gemm_select_hbm_bytes_per_launch": out.get("gemm_select_hbm_bytes_per_launch"), "source": f"profiles/{tag}_pmc_summary.json"},
          open(f"{pr}/pmc_traffic.json", "w"))
for r in csv.DictReader(open(f"{pr}/{tag}_kernel_stats.csv")):
    print("%-44s calls=%4s avg=%9.1f us  %5s%%" % (r["Name"].split("(")[0].replace("void mi::", "").replace("mi::", "")[:44], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
print({k: v for k, v in out.items() if k.startswith("gemm_select")})
for k, v in pm.items():
    if "gemm_tile_kernel<false" in k or "gemm_select_kernel<false" in k or "rescore" in k:
        print(k, {c: ("%.4g" % (x["sum"] / x["launches"])) for c, x in v.items()})
