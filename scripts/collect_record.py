"""Copies the small summaries of a scripts/profile_record.sh run from gpurun_out/ into profiles/ (tracked).
Usage: python scripts/collect_record.py <tag>"""
import glob, json, os, shutil, sys
tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
go, pr = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles")
for name in ("sync", "q1", "q70", "bf16", "aqe", "aqe_rparis", "10m"):
    src = os.path.join(go, f"{tag}_{name}_bench.json")
    if os.path.exists(src):
        lines = [l for l in open(src) if l.startswith("{")]
        if lines:
            open(os.path.join(pr, f"{tag}_{name}_bench.json"), "w").write(lines[-1])
    st = glob.glob(os.path.join(go, f"{tag}_{name}_trace", "*", "*kernel_stats.csv"))
    if st:
        shutil.copy(max(st, key=os.path.getmtime), os.path.join(pr, f"{tag}_{name}_kernel_stats.csv"))
for f in ("timeline_full.txt", "timeline_s8.txt", "rank_all.txt", "host_api.txt", "gallery_io.json", "xcc_report.txt",
          "diffusion_refsize.txt", "mfma_probe.txt", "kbench.txt", "rehearse2.txt", "rehearse4.txt", "ladder_probe.txt",
          "shard_model.txt", "protocol_rccl1.txt", "layout_model.txt", "rehearse4_rows_pipelined.txt", "tile4_probe.txt",
          "kbench_thr.txt", "bare_gpus2.txt", "bare_gpus4.txt", "tailbench.txt", "ab_r02.txt", "sweep_seeds.txt", "graph_replay.txt", "timeline_q1.txt", "timeline_q70.txt", "kbench_pmc.txt", "gap_probe.txt",
          "first_launches.txt", "shard_model_10m.txt", "ingestbench.txt", "ingest_pmc.txt"):
    src = os.path.join(go, f"{tag}_{f}")
    if os.path.exists(src) and os.path.getsize(src) < 200000:
        text = open(src, errors="replace").read()
        text = "\n".join(l for l in text.splitlines() if "amdgpu.ids" not in l and "at::native" not in l)
        open(os.path.join(pr, f"{tag}_{f}"), "w").write(text + "\n")
print(sorted(f for f in os.listdir(pr) if f.startswith(tag + "_")))
