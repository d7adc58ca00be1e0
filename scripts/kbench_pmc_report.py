"""Per-variant fabric traffic / L2 hit rate of a scripts/kbench_pmc.sh run.  FETCH_SIZE is in KiB and reads half of a wide
coalesced stream on gfx950 (MI355X_MICROARCH.md, HBM): bytes = FETCH_SIZE x 1024 x 2; WRITE_SIZE (KiB) is exact."""
import collections, csv, glob, os, re, sys
tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
go = os.path.join(root, "gpurun_out")
ALG = 1005994 * 2048 * 2 + 1024 * 2048 * 2


def rows(sub):
    fs = glob.glob(f"{go}/{tag}_{sub}/*/*counter_collection.csv")
    agg = collections.defaultdict(list)
    for f in fs:
        for r in csv.DictReader(open(f)):
            m = re.search(r"gemm_(tile|select)_kernel<([^>]*)>", r["Kernel_Name"])
            if m:
                agg[(m.group(1) + "<" + m.group(2) + ">", r["Counter_Name"])].append(float(r["Counter_Value"]))
    return agg


f, l2, w = rows("kb_fetch"), rows("kb_l2"), rows("kb_write")
names = sorted({k for k, _ in f} | {k for k, _ in l2})
print("# template arguments: FIRST, DBG, F16, REPAIR, ORDER, OPT, POL   (algorithmic bytes per launch %.3f GB)" % (ALG / 1e9))
print("%-52s %9s %7s %8s %8s" % ("kernel instantiation", "fetch GB", "x alg", "write MB", "L2 hit"))
for n in names:
    fv = [v for v in f.get((n, "FETCH_SIZE"), []) if v > 0]
    wv = w.get((n, "WRITE_SIZE"), [])
    h, m = sum(l2.get((n, "TCC_HIT_sum"), [])), sum(l2.get((n, "TCC_MISS_sum"), []))
    fb = (sum(fv) / len(fv) * 2048) if fv else float("nan")
    wb = (sum(wv) / len(wv) * 1024) if wv else float("nan")
    print("%-52s %9.3f %7.3f %8.1f %8.4f" % (n, fb / 1e9, fb / ALG, wb / 1e6, h / (h + m) if h + m else float("nan")))
