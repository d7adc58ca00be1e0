"""How long an idle gap on the device makes the NEXT scoring launches slow (DVFS transient): a steady stream of searches,
then synchronize + sleep(gap), then 14 launches; prints the first launches after every gap.
Usage: python scripts/gap_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import isehr_amd  # noqa: F401
from isehr_amd import _lib

N, D, Q, K = 1005994, 2048, 1024, 100
dev = torch.device("cuda", 0)
s = torch.cuda.current_stream().cuda_stream
raw = torch.empty((N, D), dtype=torch.float32, device=dev)
_lib.synth_fill_device(raw.data_ptr(), 1234, 0, N, D, s)
torch.cuda.synchronize()
gal = _lib.Gallery.from_device_ptr(raw.data_ptr(), N, D, norm_mode=_lib.NORM_L2, device=0)
del raw
pool = []
for i in range(4):
    qb = torch.empty((Q, D), dtype=torch.float32, device=dev)
    _lib.synth_fill_device(qb.data_ptr(), 1235 + i, 0, Q, D, s)
    pool.append(qb)
idx = torch.empty((Q, K), dtype=torch.int64, device=dev)
sc = torch.empty((Q, K), dtype=torch.float32, device=dev)
torch.cuda.synchronize()
gal.status(reset=True)
gal.profile(True)


def run(n):
    for i in range(n):
        gal.search_device(pool[i % 4].data_ptr(), Q, K, idx.data_ptr(), sc.data_ptr(), stream=s)


run(60)
torch.cuda.synchronize()
base = gal.launch_ms()
print("steady state (last 20 of 60): %.3f ms" % float(np.mean(base[-20:])))
done = len(base)
for gap in (0.0, 0.0, 50e-6, 200e-6, 1e-3, 5e-3, 20e-3, 100e-3, 1.0):
    run(30)
    torch.cuda.synchronize()                      # the gap starts here
    if gap > 0:
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < gap:
            pass
    run(14)
    torch.cuda.synchronize()
    ms = gal.launch_ms()
    new = ms[done + 30:]
    done = len(ms)
    print("gap sync + %8.0f us: " % (gap * 1e6) + " ".join("%.3f" % v for v in new[:12]) + "   mean of 14: %.3f" % float(np.mean(new)))
