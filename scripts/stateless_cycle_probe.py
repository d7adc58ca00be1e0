"""What a stateless caller pays per call on device-resident rows: create (ingest) -> first search of the handle -> destroy, phase by
phase (wall clock, device-synchronised), for 1024 and 70 queries.  Usage on the GPU box: python scripts/stateless_cycle_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import isehr_amd  # noqa: F401
from isehr_amd import _lib

n, d, k = 1005994, 2048, 100
dev = torch.device("cuda", 0)
s = torch.cuda.current_stream().cuda_stream
raw = torch.empty((n, d), dtype=torch.float32, device=dev)
_lib.synth_fill_device(raw.data_ptr(), 1234, 0, n, d, s)
for nq in (1024, 70):
    q = torch.empty((nq, d), dtype=torch.float32, device=dev)
    _lib.synth_fill_device(q.data_ptr(), 99, 0, nq, d, s)
    idx = torch.empty((nq, k), dtype=torch.int64, device=dev)
    sc = torch.empty((nq, k), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    rows = []
    for rep in range(6):
        t0 = time.perf_counter()
        g = _lib.Gallery.from_device_ptr(raw.data_ptr(), n, d)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        g.search_device(q.data_ptr(), nq, k, idx.data_ptr(), sc.data_ptr(), None, s)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        g.search_device(q.data_ptr(), nq, k, idx.data_ptr(), sc.data_ptr(), None, s)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        g.close()
        t4 = time.perf_counter()
        rows.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3))
    for name, col in (("create", 0), ("first search", 1), ("second search", 2), ("destroy", 3)):
        print("%4d queries  %-14s %s ms" % (nq, name, " ".join("%7.3f" % (r[col] * 1e3) for r in rows)))
