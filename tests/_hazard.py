"""The use-after-recycle scenario of commit 87f05b2 (VERDICT r05 Weak #7), shared by tests/test_gpu_buffer_reuse.py and
scripts/recycle_hazard_ab.sh (which runs it against a build WITHOUT the fix: it must fail there).

A search of handle A is in flight on a caller's NON-BLOCKING stream (1024 queries x 1 M rows: ~3 ms).  A's workspace is then
re-built (a search with a larger K): the old arena is parked in the process's spare slot.  Handle B -- same row width, so the
same workspace size -- starts its first search: it takes the parked arena and clears it.  Without the device synchronisation
in ws_free that happens while A's first search still reads and writes the arena: thresholds, counters and candidate lists of
a running batch are zeroed under it."""
import numpy as np


def recycle_scenario(rows=1005994, d=2048, nq=1024, k=100, rounds=3):
    """Returns (mismatching rounds, rounds).  0 mismatches = the in-flight search was not disturbed."""
    import torch
    from isehr_amd import _lib
    dev = torch.device("cuda", 0)
    _lib.set_global_option("keep_buffers", 1)
    raw = torch.empty((rows, d), dtype=torch.float32, device=dev)
    _lib.synth_fill_device(raw.data_ptr(), 4242, 0, rows, d, None)
    small = torch.empty((4096, d), dtype=torch.float32, device=dev)
    _lib.synth_fill_device(small.data_ptr(), 4243, 0, 4096, d, None)
    q = torch.empty((nq, d), dtype=torch.float32, device=dev)
    _lib.synth_fill_device(q.data_ptr(), 4244, 0, nq, d, None)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()                       # non-blocking: the null stream's work does not wait for it
    s = side.cuda_stream
    bad = 0
    try:
        for _ in range(rounds):
            _lib.set_global_option("release_spares", 1)
            A = _lib.Gallery.from_device_ptr(raw.data_ptr(), rows, d)
            B = _lib.Gallery.from_device_ptr(small.data_ptr(), 4096, d)          # no workspace yet
            try:
                ref_i = torch.empty((nq, k), dtype=torch.int64, device=dev)
                ref_s = torch.empty((nq, k), dtype=torch.float32, device=dev)
                A.search_device(q.data_ptr(), nq, k, ref_i.data_ptr(), ref_s.data_ptr(), None, s)     # workspace for K = 100
                torch.cuda.synchronize()
                got_i, got_s = torch.full_like(ref_i, -7), torch.zeros_like(ref_s)
                o2 = torch.empty((8, 2 * k), dtype=torch.int64, device=dev)
                ob = torch.empty((8, k), dtype=torch.int64, device=dev)
                A.search_device(q.data_ptr(), nq, k, got_i.data_ptr(), got_s.data_ptr(), None, s)     # in flight (~3 ms) ...
                A.search_device(q.data_ptr(), 8, 2 * k, o2.data_ptr(), None, None, s)                  # ... workspace re-built: the old arena is parked
                B.search_device(q.data_ptr(), 8, k, ob.data_ptr(), None, None, None)                   # same size: takes the parked arena and clears it
                torch.cuda.synchronize()
                if not (torch.equal(got_i, ref_i) and torch.equal(got_s, ref_s)):
                    bad += 1
            finally:
                B.close()
                A.close()
    finally:
        _lib.set_global_option("release_spares", 1)
    return bad, rounds


if __name__ == "__main__":
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import isehr_amd  # noqa: F401
    b, r = recycle_scenario()
    print("in-flight search disturbed in %d of %d rounds" % (b, r))
