"""Diagnostics: where the scoring kernel's cycles go (stamped build, option debug=8)."""
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
g.set_option("debug", 8)
for _ in range(3):
    g.search_device(q.data_ptr(), nq, k, idx.data_ptr(), sc.data_ptr(), None, s)
torch.cuda.synchronize()
c = g.debug_cycles().astype(np.float64)     # last launch = the big final chunk
sl = c[:, 5]
ok = sl > 0
for grp, name in ((0, "group0 (A loader)"), (1, "group1 (B loader)")):
    m = ok.copy(); m &= ((np.arange(len(c)) % 8) // 4 == grp)
    per = c[m, :5].sum(0) / sl[m].sum()
    e2 = (c[m, 7].astype(np.uint64) & np.uint64((1 << 40) - 1)).astype(np.float64); hits = (c[m, 7].astype(np.uint64) >> np.uint64(40)).astype(np.float64)
    tiles = sl[m].sum() / 64
    print(name, "per TILE: filter decide=%.0f hit-path=%.0f cycles, hit (lane,block) pairs=%.1f" % (c[m, 6].sum() / tiles, e2.sum() / tiles, hits.sum() / tiles))
    print(name, "cycles per slice: load=%.0f barrier1=%.0f mfma+vmwait=%.0f barrier2=%.0f epilogue(per slice)=%.0f total=%.0f" % (*per, per.sum()))

g.set_option("debug", 24)
g.search_device(q.data_ptr(), nq, k, idx.data_ptr(), sc.data_ptr(), None, s)
torch.cuda.synchronize()
c = g.debug_cycles()
dc = c[:, 6].astype(np.float64); dr = c[:, 7].astype(np.float64)
ok = dr > 0
print("shader clock during the final scoring launch: %.0f MHz (median over waves; s_memtime / s_memrealtime x 100 MHz)" % np.median(dc[ok] / dr[ok] * 100.0))
