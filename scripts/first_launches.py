"""Per-launch durations of the scoring launch in a FRESH process, from the first search after the gallery ingest
(what do the first ~25 launches pay for).  Usage: python scripts/first_launches.py [n_launches] [opt=val ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import isehr_amd  # noqa: F401
from isehr_amd import _lib

n_launch = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 60
opts = [a for a in sys.argv[1:] if "=" in a]
N, D, Q, K = 1005994, 2048, 1024, 100
dev = torch.device("cuda", 0)
s = torch.cuda.current_stream().cuda_stream
raw = torch.empty((N, D), dtype=torch.float32, device=dev)
_lib.synth_fill_device(raw.data_ptr(), 1234, 0, N, D, s)
torch.cuda.synchronize()
gal = _lib.Gallery.from_device_ptr(raw.data_ptr(), N, D, norm_mode=_lib.NORM_L2, device=0)
del raw
for o in opts:
    k_, v_ = o.split("=")
    gal.set_option(k_, float(v_))
pool = []
for i in range(4):
    qb = torch.empty((Q, D), dtype=torch.float32, device=dev)
    _lib.synth_fill_device(qb.data_ptr(), 1235 + i, 0, Q, D, s)
    pool.append(qb)
idx = torch.empty((Q, K), dtype=torch.int64, device=dev)
sc = torch.empty((Q, K), dtype=torch.float32, device=dev)
torch.cuda.synchronize()


def burst(n, label):
    gal.status(reset=True)
    gal.profile(True)
    t0 = time.perf_counter()
    for i in range(n):
        gal.search_device(pool[i % 4].data_ptr(), Q, K, idx.data_ptr(), sc.data_ptr(), stream=s)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ms = gal.launch_ms()
    st = gal.status(reset=True)
    gal.profile(False)
    print("%s: %d launches, %.3f ms per step, clock of the last launch %.0f MHz" % (label, len(ms), dt / n * 1e3, st["kernel_clock_mhz"]))
    for i in range(0, len(ms), 10):
        print("  launches %3d..%3d: " % (i, min(len(ms), i + 10) - 1) + " ".join("%.3f" % v for v in ms[i:i + 10]))
    return ms


a = burst(n_launch, "burst 1 (first searches after ingest)")
b = burst(n_launch, "burst 2 (right behind)")
time.sleep(2.0)
c = burst(n_launch, "burst 3 (after 2 s idle)")
print("mean of launches 5..24 / mean of the last 20 of burst 2: %.4f" % (float(np.mean(a[5:25])) / float(np.mean(b[-20:]))))
