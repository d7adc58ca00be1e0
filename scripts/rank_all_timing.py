"""Timing of the full-length ranking (--mode mAP shape: 70 queries x 1,005,994 rows x 2048)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import isehr_amd
from isehr_amd import _lib
from isehr_amd.synth import synth_rows
n, d, nq = 1005994, 2048, 70
dev = torch.device("cuda", 0); s = torch.cuda.current_stream().cuda_stream
raw = torch.empty((n, d), dtype=torch.float32, device=dev)
_lib.synth_fill_device(raw.data_ptr(), 1234, 0, n, d, s); torch.cuda.synchronize()
g = _lib.Gallery.from_device_ptr(raw.data_ptr(), n, d); del raw
q = synth_rows(99, 0, nq, d)
idx, sc, secs = g.rank_all(q, return_scores=True)
idx, sc, secs = g.rank_all(q, return_scores=True)
top, tsc, _ = g.search(q, 100)
print("rank_all: %d queries x %d rows in %.3f s (incl. D2H of %.0f MB); sorted=%s; top-100 sets equal the top-K path: %s"
      % (nq, n, secs, idx.nbytes / 1e6 + sc.nbytes / 1e6, bool((np.diff(sc, axis=1) <= 0).all()),
         all(set(a) == set(b) for a, b in zip(idx[:, :100], top))))
g.close()
