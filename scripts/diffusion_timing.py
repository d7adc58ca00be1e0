"""Timing of the GPU diffusion at the reference's real small-database size (rOxford5k: N = 4993, D = 2048,
n_trunc = 2000, kd = 200, src/utils/Reranking.py:230-235) on clustered synthetic descriptors."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import isehr_amd
from isehr_amd.synth import synth_rows
from isehr_amd.diffusion import Diffusion
n, d = 4993, 2048
f = synth_rows(5, 0, n, d).astype(np.float64)
c = synth_rows(6, 0, 40, d).astype(np.float64)
f = 0.8 * f + 1.5 * c[np.arange(n) % 40]
f /= np.linalg.norm(f, axis=1, keepdims=True)
f = f.astype(np.float32)
dd = Diffusion(f)
t0 = time.time(); off = dd.get_offline_results(2000, 200); t1 = time.time()
q = f[:70] + 0.05 * synth_rows(7, 0, 70, d); q /= np.linalg.norm(q, axis=1, keepdims=True)
t2 = time.time(); ranks, sc = dd.search_online(q.astype(np.float32), 3, 2000); t3 = time.time()
print("offline (kNN graph 4993x4993 top-2000, Laplacian, 4993 CG solves of 2000x2000): %.3f s; nnz=%d" % (t1 - t0, off.nnz))
print("online (70 queries: top-3, combine, top-2000): %.4f s; self-retrieval rank0 ok: %s" % (t3 - t2, bool((ranks[0] == np.arange(70)).mean() > 0.9)))
dd.close()
