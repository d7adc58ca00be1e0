"""Counterpart of the retrieval part of src/online.py:108-152 (the Flask route minus the CNN):
`Searcher.query(qvec)` = dispatch on --matching_method, then qge1 re-ranking, then the top-K paths.
A lock serialises concurrent callers (Flask's threaded server)."""
import argparse
import threading

import numpy as np

from ..nnsearch import matching_HIP
from ..reranking import qge1_hip
from .features import load_database

parser = argparse.ArgumentParser(description="Online retrieval (HIP exhaustive matcher)")
parser.add_argument("--datasets", "-d", default="database")
parser.add_argument("--matching_method", "-mm", default="HIP")
parser.add_argument("--K-nearest-neighbour", "-K", dest="K_nearest_neighbour", type=int, default=30)
parser.add_argument("--ifgenerate", "-gen", dest="ifgenerate", action="store_true")
parser.add_argument("--gpu-id", "-g", default="0")
parser.add_argument("--query-npy", help="file with query descriptors [D] or [D,Q] (stands in for the uploaded image)")


class _Request:
    """One caller's descriptors waiting for the coalescing worker: `done` is a bare lock the caller blocks on (the cheapest
    hand-off CPython has: no condition variable, no Future)."""
    __slots__ = ("desc", "event", "done", "out", "error")

    def __init__(self, desc, event):
        self.desc, self.event, self.out, self.error = desc, event, None, None
        self.done = threading.Lock()
        self.done.acquire()


class Searcher:
    """coalesce: concurrent query_device callers (src/online.py:163 runs Flask's threaded server: every request thread calls
    the route, module-level globals shared) are answered TOGETHER -- one worker thread drains the waiting descriptors (up to
    `max_batch` = 128 rows: the streaming kernel's limit) into one search -> qge1 -> re-search chain and hands every caller
    its rows.  A 70-query launch costs what a single query does (bench.py `q1` / `q70`), so 64 clients cost one chain, not 64.
    The answers are those of sequential calls, bit for bit (the exact re-score defines them, not the batch).  max_wait_s: when
    the recent chains held several requests, the worker waits up to this long for as many to be queued again before the next
    launch (the callers just answered are on their way back); a lone sequential caller never waits.
    native (default): the worker is the LIBRARY's (`mi_online_*`, csrc/api_online.hip): a request thread blocks inside one
    foreign call, outside the interpreter lock, instead of handing its request to a Python thread under it -- with 64 client
    threads the interpreter, not the GPU, bounded the Python worker (bench.py `online_concurrent`).  native=False keeps the
    Python worker (the only one for a chain on sharded galleries of a process group)."""

    def __init__(self, vecs, img_paths, K, matching_method="HIP", ifgenerate=False, device=0, coalesce=True, max_batch=128,
                 max_wait_s=5e-4, native=True):
        self.vecs, self.img_paths, self.K = vecs, img_paths, K
        self.method, self.ifgenerate, self.device = matching_method, ifgenerate, device
        self._lock = threading.Lock()
        self.coalesce, self.max_batch, self.max_wait_s = bool(coalesce), int(max_batch), float(max_wait_s)
        self._cv = threading.Condition()
        self._queue, self._worker, self._stop = [], None, False
        self.native, self._native_chain = bool(native), None
        self.batches, self.batched_requests = 0, 0           # statistics: chains launched / requests they answered

    @classmethod
    def from_galleries(cls, g_l2, g_raw, K, img_paths=None, device=0, **kw):
        """A Searcher on two galleries that are already resident (L2-normalised rows for the search, the rows as stored for
        qge1): what _device_chain builds from `vecs`, for callers that prepared them otherwise (bench.py, an offline step)."""
        from ..sharded import ShardedGallery
        s = cls(None, img_paths, K, device=device, **kw)
        s._chain = (ShardedGallery(g_l2), ShardedGallery(g_raw))
        return s

    def close(self):
        with self._cv:
            self._stop = True
            self._cv.notify_all()
            chain, self._native_chain = self._native_chain, None
        if chain is not None:
            chain.close()
        if self._worker is not None:
            self._worker.join()
            self._worker = None

    @staticmethod
    def _caller_event(desc):
        """An event behind what the caller's stream has enqueued so far (the descriptor's producer), for the worker's stream
        to wait on -- or None when the caller works on the device's default stream, which the worker uses too (stream order
        does it; an event per request is ~10 us of a 30 us budget)."""
        import torch
        if not desc.is_cuda or torch.cuda.current_stream(desc.device).cuda_stream == 0:
            return None
        ev = torch.cuda.Event()
        ev.record()
        return ev

    def _chain_rows(self, desc):
        """search (K) -> qge1 expansion (k = 3, w = 4) from the rows as stored -> re-search; one D2H copy of [Q, K] indices."""
        with self._lock:
            sg1, sg2 = self._device_chain()
            idx, _ = sg1.search(desc, self.K)
            idx2, _, _ = sg2.aqe_search(idx.t(), 3, 4.0, self.K)
            return idx2.cpu().numpy()                                 # the one D2H copy (synchronises)

    def _serve(self):
        import time
        import torch
        torch.cuda.set_device(self.device)
        # the callers known to be around when the last chain was handed out: the ones it answered (on their way back) + the
        # ones already queued behind it (csrc/api_online.hip has the same policy)
        expect, t_done = 1, time.perf_counter()
        while True:
            with self._cv:
                while not self._queue and not self._stop:
                    self._cv.wait()
                if self._stop and not self._queue:
                    return
                crowd = min(expect, self.max_batch)
                if crowd > 1 and self.max_wait_s > 0:
                    # a chain of 64 descriptors costs what a chain of one does, so the chain worth launching holds every caller
                    # that is around: wait -- until max_wait_s after the last hand-out at most -- for as many requests as that
                    # (launching whatever is queued when a chain ends settles into two half crowds taking turns).  A lone
                    # sequential caller (expect 1) never waits, nor does anyone after an idle period
                    deadline = t_done + self.max_wait_s
                    while len(self._queue) < crowd:
                        left = deadline - time.perf_counter()
                        if left <= 0:
                            break
                        self._cv.wait(left)
                take, rows = [], 0
                while self._queue and (not take or rows + self._queue[0].desc.shape[0] <= self.max_batch):
                    r = self._queue.pop(0)
                    take.append(r)
                    rows += r.desc.shape[0]
            try:
                st = torch.cuda.current_stream()
                for r in take:
                    if r.event is not None:
                        st.wait_event(r.event)                        # the caller's stream produced the descriptor
                desc = take[0].desc if len(take) == 1 else torch.cat([r.desc for r in take], dim=0)
                out = self._chain_rows(desc)
                r0 = 0
                for r in take:
                    r.out = out[r0:r0 + r.desc.shape[0]]
                    r0 += r.desc.shape[0]
            except Exception as e:                                    # every waiting caller sees the failure
                for r in take:
                    r.error = e
            self.batches += 1
            self.batched_requests += len(take)
            with self._cv:
                expect = len(take) + len(self._queue)
            for r in take:
                r.done.release()
            t_done = time.perf_counter()

    def _device_chain(self):
        """The two prepared galleries of the chain -- L2-normalised rows for the search (matching_L2's ranking), the rows as
        stored for qge1's expansion and re-search (src/online.py:132,148: `vecs` itself) -- wrapped for device tensors."""
        if getattr(self, "_chain", None) is None:
            from ..nnsearch import get_gallery, NORM_L2, NORM_NONE
            from ..sharded import ShardedGallery
            g1 = get_gallery(self.vecs.T, "database", self.ifgenerate, NORM_L2, self.device)
            g2 = get_gallery(self.vecs.T, "database_raw", self.ifgenerate, NORM_NONE, self.device)
            self.ifgenerate = False
            self._chain = (ShardedGallery(g1), ShardedGallery(g2))
        return self._chain

    def _native(self):
        """The library's coalescing front on the two galleries of the chain, or None (sharded chain / native=False)."""
        if self._native_chain is None:
            with self._lock:
                if self._native_chain is None:
                    from .. import _lib
                    sg1, sg2 = self._device_chain()
                    if sg1._protocol or sg2._protocol:
                        self.native = False
                        return None
                    self._native_chain = _lib.OnlineChain(sg1.g, sg2.g, self.K, 3, 4.0, 1e-6, self.max_batch,
                                                          int(round(self.max_wait_s * 1e6)))
        return self._native_chain

    @property
    def chain_stats(self):
        """(chains launched, requests they answered) of whichever worker serves this Searcher."""
        if self._native_chain is not None:
            st = self._native_chain.stats()
            return st["chains"], st["requests"]
        return self.batches, self.batched_requests

    def query_device(self, desc, return_indices=False):
        """The online chain WITHOUT a host round trip (src/online.py:121-152 copies the descriptor to the CPU,
        searches there and re-ranks there): `desc` is the extractor tail's output on the device (float32 cuda tensor [D] or
        [Q, D]: isehr_amd.extractor.DescriptorTail / extract_ms_device -> mi_desc_tail_device), the search (K nearest by cosine),
        the qge1 expansion from the top-3 rows (k = 3, w = 4, float64 sum, eps-normalised) and the re-search of the expanded
        query all run on the device through the asynchronous entry points; ONE device-to-host copy of Q x K indices ends it."""
        import torch
        nq = 1 if desc.dim() == 1 else desc.shape[0]
        chain = None
        if self.coalesce and self.native and nq <= self.max_batch:
            chain = self._native_chain or self._native()
        if chain is not None:
            # the request thread's whole share of interpreter time: no view, no copy of a descriptor that is float32 and
            # contiguous already, one foreign call (which releases the interpreter lock while the chain runs)
            if desc.dtype is not torch.float32 or not desc.is_contiguous():
                desc = desc.contiguous().float()
            if desc.is_cuda:
                # the producer's stream as a raw handle (what torch.cuda.current_stream(dev).cuda_stream returns, without
                # building the Stream object: ~3 us of the ~15 a request spends under the interpreter lock)
                raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
                st = raw(desc.device.index) if raw is not None else torch.cuda.current_stream(desc.device).cuda_stream
                out = chain.query(desc.data_ptr(), nq, 1, True, st or None)
            else:
                out = chain.query(desc.data_ptr(), nq, 0)
            if return_indices:
                return out
            return [[self.img_paths[i] for i in row] for row in out]
        if desc.dim() == 1:
            desc = desc[None, :]
        desc = desc.contiguous().float()
        if self.coalesce and desc.shape[0] <= self.max_batch:
            req = _Request(desc, self._caller_event(desc))
            with self._cv:
                if self._worker is None:
                    self._stop = False
                    self._worker = threading.Thread(target=self._serve, name="isehr-online-coalescer", daemon=True)
                    self._worker.start()
                self._queue.append(req)
                self._cv.notify()
            req.done.acquire()                                        # released by the worker
            if req.error is not None:
                raise req.error
            out = req.out
        else:
            out = self._chain_rows(desc)
        if return_indices:
            return out
        return [[self.img_paths[i] for i in row] for row in out]

    def query(self, qvec):
        qvec = np.asarray(qvec)
        if qvec.ndim == 1:
            qvec = qvec[:, None]                                  # src/online.py:123
        with self._lock:
            if self.method == "HIP":
                match_idx, _ = matching_HIP(self.K, self.vecs.T, qvec.T, dataset="database",
                                            ifgenerate=self.ifgenerate, device=self.device)
            else:
                raise ValueError("Invalid method")               # the reference prints and then fails on a NameError
            self.ifgenerate = False
            ranks = match_idx.T
            ranks2 = qge1_hip(ranks, qvec, self.vecs, self.K, dataset="database_raw", device=self.device)
        idx2 = ranks2.T
        return [[self.img_paths[i] for i in row[:self.K]] for row in idx2]


def main(argv=None):
    args = parser.parse_args(argv)
    vecs, paths = load_database(args.datasets.split(","))
    s = Searcher(vecs, paths, args.K_nearest_neighbour, args.matching_method, args.ifgenerate, int(args.gpu_id))
    if args.query_npy:
        for row in s.query(np.load(args.query_npy)):
            print("\n".join(map(str, row)))
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
