"""PCIe-inclusive rate: the host entry point mi_knn_search (host queries in, host results out, per-call staging)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import isehr_amd
from isehr_amd import _lib
from isehr_amd.synth import synth_rows
n, d, nq, k = 1005994, 2048, 1024, 100
dev = torch.device("cuda", 0); s = torch.cuda.current_stream().cuda_stream
raw = torch.empty((n, d), dtype=torch.float32, device=dev)
_lib.synth_fill_device(raw.data_ptr(), 1234, 0, n, d, s); torch.cuda.synchronize()
g = _lib.Gallery.from_device_ptr(raw.data_ptr(), n, d); del raw
q = synth_rows(99, 0, nq, d)
for _ in range(2): g.search(q, k)
t0 = time.time(); reps = 10
for _ in range(reps): idx, sc, secs = g.search(q, k)
dt = (time.time() - t0) / reps
print("host API (H2D of %.1f MB queries, search, D2H of results): %.3f ms per 1024-query batch = %.0f queries/s" % (q.nbytes / 1e6, dt * 1e3, nq / dt))
q1 = q[:1]
t0 = time.time()
for _ in range(50): g.search(q1, k)
dt1 = (time.time() - t0) / 50
print("host API, single query (online.py shape): %.3f ms per query" % (dt1 * 1e3))
g.close()
