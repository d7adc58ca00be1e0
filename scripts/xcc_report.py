"""Diagnostics: per-XCD loop time and tile share of the tile kernel, with and without the measured-speed split."""
import sys, os
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
for bal in (0, 1):
    g.set_option("xcc_balance", bal)
    for it in range(8):
        g.search_device(q.data_ptr(), nq, k, idx.data_ptr(), sc.data_ptr(), None, s)
        torch.cuda.synchronize()
        c = g.debug_cycles().astype(np.float64)[::8]          # wave 0 of every workgroup
        lab = np.arange(len(c)) % 8
        dur = [c[lab == x, 7].max() * 0.01 for x in range(8)]
        tiles = [c[lab == x, 5].sum() / 64 / 4 for x in range(8)]
        if it in (0, 1, 2, 4, 7):
            print("balance=%d launch %d: loop us per XCD label max %s | tiles %s | spread %.2f %%" % (
                bal, it, " ".join("%.0f" % v for v in dur), " ".join("%.0f" % v for v in tiles),
                100 * (max(dur) / np.mean(dur) - 1)))
