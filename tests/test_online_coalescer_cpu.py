"""CPU: the coalescing worker of entry/online.py Searcher (src/online.py:163 runs Flask's threaded server: concurrent request
threads) with the GPU chain replaced by a stand-in -- the queueing logic alone: every caller gets ITS rows back, requests are
batched, a lone caller is never delayed by the straggler wait, oversized and multi-row requests keep their rows together, a
failure inside a chain reaches every caller of that chain, close() drains.
(native=False: the Python worker; the library's own, mi_online_*, needs the device: tests/test_gpu_entry_points.py.)"""
import threading
import time
import types

import numpy as np
import pytest


@pytest.fixture()
def searcher_cls(monkeypatch):
    import torch
    from isehr_amd.entry import online
    # the worker's two touches of the device: its current stream (to wait for callers' events) and set_device
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: None)
    monkeypatch.setattr(torch.cuda, "current_stream", lambda *a: types.SimpleNamespace(cuda_stream=0, wait_event=lambda e: None))

    class Stub(online.Searcher):
        chain_s = 0.002
        fail_on = None

        def _chain_rows(self, desc):                       # stands for search -> qge1 -> re-search -> one D2H copy
            time.sleep(self.chain_s)
            if self.fail_on is not None and bool((desc[:, 0] == self.fail_on).any()):
                raise RuntimeError("chain failed")
            # row i of the answer encodes the descriptor it belongs to
            return np.repeat(desc[:, :1].numpy().astype(np.int64), self.K, axis=1)
    return Stub


def test_concurrent_callers_get_their_own_rows_and_are_batched(searcher_cls):
    import torch
    srv = searcher_cls(None, None, 7, coalesce=True, native=False)
    nthr, per = 32, 12
    got = np.full((nthr * per,), -1, dtype=np.int64)
    errs = []

    def client(t):
        try:
            for i in range(per):
                j = t * per + i
                d = torch.full((16,), float(j))
                out = srv.query_device(d, return_indices=True)
                assert out.shape == (1, 7)
                got[j] = out[0, 3]
        except Exception as e:                                 # noqa: BLE001
            errs.append(e)
    try:
        assert srv.query_device(torch.full((16,), 5.0), return_indices=True)[0, 0] == 5 and srv.batches == 1
        ths = [threading.Thread(target=client, args=(t,)) for t in range(nthr)]
        for th in ths:
            th.start()
        for th in ths:
            th.join(timeout=60)
        assert not errs, errs[:1]
        assert np.array_equal(got, np.arange(nthr * per))
        assert srv.batched_requests == nthr * per + 1 and srv.batches <= nthr * per // 3
        # a [Q, D] request keeps its rows together; one wider than max_batch bypasses the queue
        many = torch.arange(10, dtype=torch.float32)[:, None].repeat(1, 16)
        assert np.array_equal(srv.query_device(many, return_indices=True)[:, 0], np.arange(10))
        b0 = srv.batches
        wide = torch.arange(200, dtype=torch.float32)[:, None].repeat(1, 16)
        assert np.array_equal(srv.query_device(wide, return_indices=True)[:, 0], np.arange(200)) and srv.batches == b0
    finally:
        srv.close()


def test_a_lone_sequential_caller_never_waits_for_stragglers(searcher_cls):
    import torch
    srv = searcher_cls(None, None, 3, coalesce=True, native=False, max_wait_s=0.05)        # a straggler wait that would be visible
    srv.chain_s = 0.0005
    try:
        d = torch.zeros(8)
        srv.query_device(d, return_indices=True)
        t0 = time.perf_counter()
        for _ in range(40):
            srv.query_device(d, return_indices=True)
        per_call = (time.perf_counter() - t0) / 40
        assert per_call < 0.02, per_call                   # 50 ms waits would make this 0.05+
        assert srv.batches == 41 and srv.batched_requests == 41
    finally:
        srv.close()


def test_a_failing_chain_reaches_its_callers_and_the_worker_lives_on(searcher_cls):
    import torch
    srv = searcher_cls(None, None, 3, coalesce=True, native=False)
    srv.fail_on = 13.0
    try:
        with pytest.raises(RuntimeError, match="chain failed"):
            srv.query_device(torch.full((4,), 13.0), return_indices=True)
        assert srv.query_device(torch.full((4,), 2.0), return_indices=True)[0, 0] == 2
        # paths are looked up for the caller when it does not ask for indices
        srv.img_paths = ["p%d" % i for i in range(20)]
        assert srv.query_device(torch.full((4,), 4.0)) == [["p4"] * 3]
    finally:
        srv.close()
    assert srv._worker is None
