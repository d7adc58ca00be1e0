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
    print(name, "cycles per slice: load=%.0f barrier1=%.0f mfma+vmwait=%.0f barrier2=%.0f epilogue(per slice)=%.0f total=%.0f" % (*per, per.sum()))
