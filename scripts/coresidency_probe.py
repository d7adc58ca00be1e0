"""Does a memory-bound kernel on a second stream run BESIDE the persistent tile kernel (which holds one 512-thread workgroup
and ~155 KB of LDS on every CU, 2 x 208 VGPRs per SIMD lane) or only after it?  Times K search steps alone, K gathers alone
and both enqueued together."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import isehr_amd  # noqa: F401
from isehr_amd import _lib
n, d, nq, k = 1005994, 2048, 1024, 100
dev = torch.device("cuda", 0)
s = torch.cuda.current_stream().cuda_stream
raw = torch.empty((n, d), dtype=torch.float32, device=dev)
_lib.synth_fill_device(raw.data_ptr(), 1234, 0, n, d, s); torch.cuda.synchronize()
g = _lib.Gallery.from_device_ptr(raw.data_ptr(), n, d)
q = torch.empty((nq, d), dtype=torch.float32, device=dev); _lib.synth_fill_device(q.data_ptr(), 99, 0, nq, d, s)
idx = torch.empty((nq, k), dtype=torch.int64, device=dev); sc = torch.empty((nq, k), dtype=torch.float32, device=dev)
side = torch.cuda.Stream()
rows = torch.randint(0, n, (131072,), device=dev)          # 131072 random rows x 8 KiB = 1.07 GB gathered
out = torch.empty((131072, d), dtype=torch.float32, device=dev)
def search(reps):
    for _ in range(reps):
        g.search_device(q.data_ptr(), nq, k, idx.data_ptr(), sc.data_ptr(), None, s)
def gather(reps):
    with torch.cuda.stream(side):
        for _ in range(reps):
            torch.index_select(raw, 0, rows, out=out)
def timed(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
search(3); gather(3)
K = 10
a = timed(lambda: search(K)) / K
b = timed(lambda: gather(K)) / K
c = timed(lambda: (search(K), gather(K))) / K
print("search alone %.3f ms/step, 1.07 GB row gather alone %.3f ms, both enqueued together %.3f ms per pair (sum %.3f)" % (a, b, c, a + b))
