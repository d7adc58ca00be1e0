# LDS bank conflicts of the gallery ingest kernels, old source (ab/ingest_old.hip) against the tree's:
#   bash scripts/ingest_lds_conflicts.sh <tag>      (one rocprofv3 --pmc pass per build and layout)
set -e
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P=image-search-engine-for-historical-research_amd/build; C=image-search-engine-for-historical-research_amd/csrc; o=gpurun_out
mkdir -p $P $o
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -DMI_INGEST_PROBE=128 scripts/ingestbench.hip -o $P/ingestbench_new 2> /dev/null
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -DMI_INGEST_PROBE=128 -DMI_INGEST_SRC='"../ab/ingest_old.hip"' -I $C scripts/ingestbench.hip -o $P/ingestbench_old 2> /dev/null
for b in old new; do for l in rows cols; do
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $o/${tag}_lds_${b}_${l} -- $P/ingestbench_$b 1005994 2048 $l > $o/${tag}_lds_${b}_${l}.log 2>&1 || true
done; done
python3 - <<PY
import csv, glob
for b in ("old", "new"):
    for l in ("rows", "cols"):
        acc = {}
        for f in glob.glob("gpurun_out/${tag}_lds_%s_%s/*/*counter_collection.csv" % (b, l)):
            for r in csv.DictReader(open(f)):
                if "ingest_" in r["Kernel_Name"]:
                    acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        if acc:
            c = sum(acc["SQ_LDS_BANK_CONFLICT"]) / len(acc["SQ_LDS_BANK_CONFLICT"]); a = sum(acc["SQ_LDS_IDX_ACTIVE"]) / len(acc["SQ_LDS_IDX_ACTIVE"])
            print("%s %s: SQ_LDS_BANK_CONFLICT %.3e / SQ_LDS_IDX_ACTIVE %.3e = %.3f per launch (%d launches)" % (b, l, c, a, c / a, len(acc["SQ_LDS_IDX_ACTIVE"])))
PY
