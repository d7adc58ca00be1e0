#!/usr/bin/env python3
"""Gallery.from_host at BASELINE size under the host-ingest modes (mi_set_global_option "host_ingest": 0 = one copy of the
whole array + one ingest, 1 = row blocks copied by the runtime from the pageable array; the record profiles/r05f_host_ingest_modes.json
also holds mode 2 of the build it was taken with -- blocks through pinned buffers filled by host threads, since removed) for the reference's [D, N] layout (vecs.T) and for a row-major array, float32 and float64.
    python scripts/host_ingest_modes.py [--rows 1005994]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1005994)
    args = ap.parse_args()
    import numpy as np
    import torch
    import isehr_amd  # noqa: F401
    from isehr_amd import _lib
    d, n = 2048, args.rows
    stream = torch.cuda.current_stream().cuda_stream
    vecs = np.empty((d, n), dtype=np.float32)
    for r0 in range(0, n, 131072):
        m = min(131072, n - r0)
        raw = torch.empty((m, d), dtype=torch.float32, device="cuda")
        _lib.synth_fill_device(raw.data_ptr(), 1234, r0, m, d, stream)
        vecs[:, r0:r0 + m] = raw.t().contiguous().cpu().numpy()
    del raw
    rows = np.ascontiguousarray(vecs[:, :n // 2].T)                 # row-major, half the rows (4 GB)
    v64 = vecs[:, :131072].astype(np.float64)
    out = {"rows": n}
    cases = {"f32 [D,N].T %d rows" % n: vecs.T, "f32 row-major %d rows" % (n // 2): rows, "f64 [D,N].T 131072 rows": v64.T}
    _lib.Gallery.from_host(vecs[:, :4096].T).close()
    sums = {}
    for name, arr in cases.items():
        out[name] = {}
        for mode, threads in ((0, 4), (1, 4), (0, 4), (1, 4)):
            _lib.set_global_option("host_ingest", mode)
            t0 = time.perf_counter()
            g = _lib.Gallery.from_host(arr)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            first = g.get_rows(arr.shape[0] - 3, 3)
            g.close()
            key = "mode%d_threads%d" % (mode, threads)
            out[name].setdefault(key, []).append(round(dt, 4))
            sums.setdefault(name, []).append(first.tobytes())
        out[name]["GBps_best"] = {k: round(arr.nbytes / min(v) / 1e9, 1) for k, v in out[name].items() if isinstance(v, list)}
        out[name]["same_rows_in_every_mode"] = len(set(sums[name])) == 1
    _lib.set_global_option("host_ingest", 1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
