import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import isehr_amd
from isehr_amd import _lib
n, d, nq, k = 1005994, 2048, 1024, 100
dev = torch.device("cuda", 0)
raw = torch.empty((n, d), dtype=torch.float32, device=dev)
s = torch.cuda.current_stream().cuda_stream
_lib.synth_fill_device(raw.data_ptr(), 1234, 0, n, d, s); torch.cuda.synchronize()
g = _lib.Gallery.from_device_ptr(raw.data_ptr(), n, d); del raw
q = torch.empty((nq, d), dtype=torch.float32, device=dev); _lib.synth_fill_device(q.data_ptr(), 99, 0, nq, d, s)
idx = torch.empty((nq, k), dtype=torch.int64, device=dev); sc = torch.empty((nq, k), dtype=torch.float32, device=dev)
for lad in (0, 1, 0, 1):
    g.set_option("ladder", lad)
    for it in range(3):
        g.search_device(q.data_ptr(), nq, k, idx.data_ptr(), sc.data_ptr(), None, s)
    torch.cuda.synchronize()
    c = g.debug_cycles().astype(np.float64)
    print("ladder=%d records per query emitted by the scoring launch: %.1f" % (lad, c[:, 3].sum() / nq), " flags", g.flags())
