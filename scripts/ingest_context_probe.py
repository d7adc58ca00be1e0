"""Where the ingest kernel's launch-to-launch spread comes from (3.6-3.8 ms back to back in ingestbench, 3.7-5.1 ms inside
bench.py): the same 1 005 994 x 2048 rows appended (HIP events on the launch stream) to
  a) a gallery created just before (fresh 12 GB allocation), first launch after the allocation;
  b) one gallery of 4x the capacity, four appends back to back (steady state; destinations differ);
  c) as a) after a 200 ms idle gap;   d) as a) with a 20 ms busy kernel in front (clocks up, fresh allocation).
Run on the GPU box: python scripts/ingest_context_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import isehr_amd
from isehr_amd import _lib

n, d = 1005994, 2048
dev = torch.device("cuda", 0)
raw = torch.empty((n, d), dtype=torch.float32, device=dev)
s = torch.cuda.current_stream().cuda_stream
_lib.synth_fill_device(raw.data_ptr(), 1234, 0, n, d, s)
torch.cuda.synchronize()
busy = torch.empty((8192, 8192), dtype=torch.float32, device=dev)


def timed_append(g, m=n):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    g.append_device(raw.data_ptr(), m, s)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b)


def fresh(pre=None):
    t0 = time.time()
    g = _lib.Gallery.empty(n, d, norm_mode=_lib.NORM_L2, device=0)
    torch.cuda.synchronize()
    alloc = time.time() - t0
    if pre:
        pre()
    ms = timed_append(g)
    g.close()
    return ms, alloc


print("a) fresh allocation, then append:", ["%.3f ms (alloc %.1f ms)" % (ms, al * 1e3) for ms, al in (fresh() for _ in range(6))])
m = n // 4 * 4                                  # appends that start at multiples of 4 rows (the cooperative image stores)
g = _lib.Gallery.empty(8 * n, d, norm_mode=_lib.NORM_L2, device=0)
torch.cuda.synchronize()
print("b) eight appends back to back into one gallery:", ["%.3f" % timed_append(g, m) for _ in range(8)])
g.close()
g = _lib.Gallery.empty(8 * n, d, norm_mode=_lib.NORM_L2, device=0)
torch.cuda.synchronize()
out = []
for _ in range(8):
    time.sleep(0.003)
    out.append("%.3f" % timed_append(g, m))
print("b') the same with 3 ms of idle GPU before each:", out)
g.close()
print("c) fresh allocation after 200 ms idle:", ["%.3f" % fresh(lambda: time.sleep(0.2))[0] for _ in range(3)])
print("d) fresh allocation, 20 ms of copies in front:",
      ["%.3f" % fresh(lambda: [busy.copy_(busy.roll(1, 0)) for _ in range(20)])[0] for _ in range(3)])
g = _lib.Gallery.empty(n, d, norm_mode=_lib.NORM_L2, device=0)
torch.cuda.synchronize()
time.sleep(0.2)
print("e) allocation made 200 ms earlier, GPU idle since:", "%.3f" % timed_append(g))
g.close()
