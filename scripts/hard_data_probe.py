"""Which sticky flags the structured galleries of bench.py's `hard_data` block raise, stage by stage:
python scripts/hard_data_probe.py [kind ...]   (kinds: nonneg clustered near_duplicates)
Per (kind, batch size): one batch through the device entry point on the speculative schedule, then with speculative = 0
(rigorous chunk schedule), then with force_exact = 1 (f32 scorer); flags (1 survivor overflow, 2 candidate overflow, 4 record
overflow, 8 fp16 range, 16 speculative threshold failed), survivors / candidates per query."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import isehr_amd  # noqa: E402,F401
from isehr_amd import _lib  # noqa: E402
from isehr_amd.sharded import ShardedGallery  # noqa: E402
import bench  # noqa: E402

kinds = sys.argv[1:] or ["nonneg", "clustered", "near_duplicates"]
n, d, k = int(os.environ.get("PROBE_ROWS", 1005994)), 2048, 100
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
for ci, kind in enumerate(("nonneg", "clustered", "near_duplicates")):
    if kind not in kinds:
        continue
    raw, queries, desc = bench._hard_rows(kind, n, d, dev, 1234 + 500 + ci)
    torch.cuda.synchronize()
    gal = _lib.Gallery.from_device_ptr(raw.data_ptr(), n, d, norm_mode=_lib.NORM_L2, device=0)
    del raw
    print("==", kind, desc, "norm bounds", gal.norm_bounds(), "f16", gal.get_option("image_dtype"), flush=True)
    sg = ShardedGallery(gal)
    for nq in (1024, 70, 1):
        q = queries(nq).contiguous()
        stages = (("speculative", {}), ("chunks", {"speculative": 0}), ("f32", {"force_exact": 1}))
        if os.environ.get("PROBE_STAGES") == "sample_vs_chunk":
            # large shards (PROBE_ROWS > 1.3 M): the 24 576-row hashed sample against the chunk schedule (first rows as the sample)
            stages = (("hashed_sample", {}), ("chunk_schedule", {"spec_max_ratio": 20}))
        for name, opts in stages:
            for o, v in opts.items():
                gal.set_option(o, v)
            gal.flags()
            gal.status(reset=True)
            sg.search(q, k)
            torch.cuda.synchronize()
            fl = gal.flags()
            st = gal.status(reset=True)
            for o in opts:
                gal.set_option(o, {"speculative": 1, "force_exact": 0, "spec_max_ratio": 160}[o])
            print("  q%-5d %-12s flags=%2d survivors/q=%9.1f candidates/q=%8.1f" %
                  (nq, name, fl, st["survivors"] / max(1, st["queries"]), st["candidates"] / max(1, st["queries"])), flush=True)
            if name in ("speculative", "hashed_sample") and nq > 128:
                import numpy as np
                cyc = gal.debug_cycles(2048)
                rec = cyc[:, 3].astype(np.int64)
                top = np.argsort(-rec)[:8]
                print("        records per wave: mean %.0f max %d; top waves (workgroup, wave, records): %s" %
                      (rec.mean(), rec.max(), [(int(i) // 8, int(i) % 8, int(rec[i])) for i in top]), flush=True)
    gal.close()
    torch.cuda.empty_cache()
