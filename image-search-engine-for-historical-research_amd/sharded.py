"""Row-sharded gallery over up to 8 GPUs of one node: one process per GPU, RCCL over xGMI.

New functionality (the reference is single-process).  Every rank owns a contiguous
row range as one `Gallery` handle.  A query batch (replicated on every rank) is answered in two
phases with two small all-gathers, so that the exact re-scoring work is split across the shards:

  phase 1  local bf16 MFMA scoring + filtering          -> the shard's K largest approximate scores
  gather 1 all-gather [G][Q][K] f32 (K*4 B per query per rank; 400 KiB at Q=1024, K=100)
           L[q] = K-th largest of the union = K-th largest approximate score of the whole gallery
  phase 2  exact f64 re-score of local rows within the error margin of L -> local exact top-K
  gather 2 ONE all-gather of the packed [G][2][Q][K] (f64 score, i64 idx) lists (1.6 MiB per rank at Q=1024, K=100)
  merge    (score desc, idx asc) -> identical to the single-GPU result, bit for bit

Both collectives are latency-bound on xGMI; they run on the caller's stream order (no host sync).
"""
import numpy as np

from . import _lib


def shard_bounds(n, world, rank):
    """Contiguous row range [lo, hi) of `rank`: ceil(n / world) rows per shard."""
    per = -(-n // world)
    lo = min(n, rank * per)
    return lo, min(n, lo + per)


def job_layout(world, rank, nq, layout="auto"):
    """A strong-scaling job on `world` GPUs as gq query groups x gs row shards (gq * gs = world): every batch of nq queries
    is cut into gq slices, the gs ranks of a group shard the gallery rows among themselves (two-phase protocol with
    all-gathers INSIDE the group), and groups never exchange anything -- a query's answer lives with the group that
    computed it.  Fewer ranks per collective and per-query kernels of nq / gq queries per rank; each rank streams a gallery
    shard gq times larger, which 288 GB per GPU has room for.  'auto' = 2 x world/2 for an even world and batches of >= 512
    queries (measured per-rank steps of every layout of 2, 4, 8 GPUs: scripts/layout_model.sh), else 1 x world.
    Returns (gq, gs, query group of `rank`, row shard of `rank` inside its group)."""
    if layout == "auto":
        gq = 2 if (world % 2 == 0 and nq >= 512 and nq % 2 == 0) else 1
    else:
        a, b = (int(v) for v in str(layout).lower().split("x"))
        if a * b != world or nq % a:
            raise ValueError("layout %s does not fit %d ranks / %d queries" % (layout, world, nq))
        gq = a
    gs = world // gq
    return gq, gs, rank // gs, rank % gs


def layout_groups(gq, gs):
    """The process groups of a layout: group g = ranks [g * gs, (g + 1) * gs).  Every rank must call this (collective
    creation); returns the list of all gq groups."""
    import torch.distributed as dist
    return [dist.new_group(list(range(g * gs, (g + 1) * gs))) for g in range(gq)]


def all_gather_stacked(t, group=None):
    """[...] tensor -> [world, ...] tensor, same on every rank (RCCL on GPU, gloo on CPU)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    t = t.contiguous()
    # concatenated along dim 0 (the form both RCCL and gloo accept), viewed as [world, ...]
    out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t, group=group)
    return out.view((world,) + tuple(t.shape))


class ShardedGallery:
    """`gallery` is this rank's shard (created with row_offset = shard_bounds(...)[0])."""

    def __init__(self, gallery, group=None, force_protocol=False):
        """force_protocol: run the two-phase protocol and its collectives even when the group has ONE rank (a process group
        must be initialised) -- the way to exercise the RCCL path itself on a one-GPU box."""
        import torch.distributed as dist
        self.g = gallery
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self._buf = {}
        self._turn = {}                    # (nq, k, tag) -> ring position of the result buffers handed out last
        self._tag = "search"
        self._join = True
        self._protocol = self.world > 1 or (force_protocol and dist.is_available() and dist.is_initialized())
        if self._protocol:
            self._agree()
            # a shard of a G-way gallery expects ~127 / G candidates per query: size the re-score launch for that (a grid
            # for 128 per query is mostly workgroups that exit at once, 23 us of dispatch at G = 8); larger counts loop
            self.g.set_option("rescore_grid_x", max(8, min(64, round(96 / self.world))))

    def _agree(self):
        """One error margin for all shards: the certificate's eps must cover the rows of EVERY shard (a row at the
        K-th approximate score L may sit on a shard with larger rounding norms than mine), so the norm maxima are
        all-reduced (MAX) and the image element type is the coarsest any shard chose (MIN: a raw shard with large rows
        falls back to bf16 on its own, csrc/api_gallery.hip mi_gallery_create)."""
        import torch
        import torch.distributed as dist
        dev = "cuda" if dist.get_backend(self.group) == "nccl" else "cpu"
        f16 = torch.tensor([int(self.g.get_option("image_dtype"))], dtype=torch.int32, device=dev)
        dist.all_reduce(f16, op=dist.ReduceOp.MIN, group=self.group)
        if int(f16.item()) != int(self.g.get_option("image_dtype")):
            self.g.set_image_dtype(int(f16.item()))
        b = torch.tensor(self.g.norm_bounds(), dtype=torch.float32, device=dev)
        dist.all_reduce(b, op=dist.ReduceOp.MAX, group=self.group)
        self.g.norm_bounds(raise_to=[float(v) for v in b.tolist()])

    def _ring_buffers(self, nq, k, device):
        """Result buffers of a synchronous entry point: a ring of TWO per (shape, entry point), so that what search() returned
        survives the next search() of the same shape, and a search() result survives any aqe_search() (whose internal
        re-search has a ring of its own): `ranks = search(..); aqe_search(ranks.t(), ..); use(ranks)` -- the order of
        src/online.py:132-152 -- reads what it was given."""
        key = (nq, k, self._tag)
        turn = self._turn[key] = self._turn.get(key, 1) ^ 1
        return self._buffers(nq, k, device, slot=(self._tag, turn))

    def _buffers(self, nq, k, device, slot=0):
        import torch
        key = (nq, k, slot)
        if key not in self._buf:
            self._buf[key] = dict(
                approx=torch.empty((nq, k), dtype=torch.float32, device=device),
                L=torch.empty((nq,), dtype=torch.float32, device=device),
                sc=torch.empty((nq, k), dtype=torch.float32, device=device),
                # exact scores (f64, bit pattern kept in int64) and global indices of phase 2, packed so that ONE
                # all-gather moves both
                pack=torch.empty((2, nq, k), dtype=torch.int64, device=device),
                oidx=torch.empty((nq, k), dtype=torch.int64, device=device),
                osc=torch.empty((nq, k), dtype=torch.float32, device=device))
        return self._buf[key]

    def search(self, q, k, query_norm_none=False, verify=False, join=True):
        """q: [Q, D] float32 cuda tensor (same on every rank), Q <= 1024.
        Returns (idx int64 [Q,k], score float32 [Q,k]) cuda tensors, identical on every rank.  They are this object's
        buffers: a ring of two per (Q, k), so a result stays intact through the NEXT search of the same shape and through
        any aqe_search (which keeps a ring of its own); the second-next search of the same shape overwrites it (clone what
        must live longer).
        query_norm_none: use the queries as they are (expanded queries of alpha-QE).

        The device entry points run asynchronously and report buffer overflows / a failed speculative threshold
        through sticky flags (include/mi355_retrieval.h).  verify=True synchronises, reads the flags of EVERY shard
        (all-reduce) and, if one is raised anywhere, answers the batch again on all shards with the fallbacks the
        host entry point applies on its own: the rigorous chunk schedule, then the f32 scorer.  The result is exact
        either way; a caller that batches many searches can instead call `any_flag()` once at the end.

        join: only matters when the shard's handle runs in the asynchronous-tail mode (`set_option("async_tail", 1)`,
        single shard): join=False leaves the exact re-score + sort of this batch running on the handle's own stream
        beside the scoring launch of the next search; the returned tensors are complete after `self.g.join(stream)`."""
        if query_norm_none:
            self.g.set_option("query_norm_override", _lib.NORM_NONE)
        self._join = join or verify
        try:
            out = self._search(q, k)
            if verify and self.any_flag():
                # the fallbacks of the host entry point, in its order; the caller's option values are restored afterwards
                for name, val in (("speculative", 0), ("force_exact", 1)):
                    prev = self.g.get_option(name)
                    self.g.set_option(name, val)
                    try:
                        out = self._search(q, k)
                        bad = self.any_flag()
                    finally:
                        self.g.set_option(name, prev)
                    if not bad:
                        break
                else:
                    if self._protocol:
                        raise RuntimeError("sharded search: candidate buffers overflow even with the f32 scorer "
                                           "(massive ties: use the dense path)")
                    # one shard: the last resort of the host entry point -- every score in float64, exact top-k of the dense
                    # rows (mi_knn_dense64_search); any data, slow
                    qh = q.detach().cpu().numpy()
                    if query_norm_none:
                        raise RuntimeError("sharded search: candidate buffers overflow even with the f32 scorer (expanded "
                                           "queries: call Gallery.dense64_search on a MI_NORM_NONE gallery)")
                    di, ds, _, _ = self.g.dense64_search(qh, k)
                    import torch
                    out[0].copy_(torch.from_numpy(di).to(out[0].device))
                    out[1].copy_(torch.from_numpy(ds).to(out[1].device))
            return out
        finally:
            if query_norm_none:
                self.g.set_option("query_norm_override", -1)

    def search_stream(self, batches, k, query_norm_none=False):
        """Pipelined search over an iterable of query batches (each [Q, D] float32 cuda, Q <= 1024): a generator of
        (idx, score) in the order of the batches.  Both all-gathers of a batch are asynchronous and are waited for only
        after phase 1 of the NEXT batch (and phase 2 of the one between) has been enqueued, so their latency hides behind
        compute instead of idling the GPU twice per batch.  Three batches are in flight; the handle keeps two
        workspaces (option "workspace_slot").  The yielded tensors are this object's buffers of the batch's slot: they stay
        valid until two further batches have been yielded (clone what must live longer).  Same answers as search()."""
        import torch
        import torch.distributed as dist
        if not self._protocol:
            for q in batches:
                yield self.search(q, k, query_norm_none=query_norm_none)
            return
        stream = torch.cuda.current_stream().cuda_stream
        G = self.world

        def gather_async(t):
            t = t.contiguous()
            out = torch.empty((G * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
            return out, dist.all_gather_into_tensor(out, t, group=self.group, async_op=True)

        def finish1(rec):                       # first collective done -> L -> phase 2 -> second collective (async)
            slot, b, nq, g1, w1 = rec
            w1.wait()
            self.g.set_option("workspace_slot", slot)
            _lib.kth_of_gathered_device(g1.data_ptr(), G, nq, k, b["L"].data_ptr(), stream)
            pack = b["pack"]
            self.g.phase2_device(nq, k, b["L"].data_ptr(), pack[1].data_ptr(), b["sc"].data_ptr(), pack[0].data_ptr(), stream)
            g2, w2 = gather_async(pack)
            return slot, b, nq, g2, w2

        def finish2(rec):                       # second collective done -> merge
            slot, b, nq, g2, w2 = rec
            w2.wait()
            g2 = g2.view((G,) + tuple(b["pack"].shape))
            _lib.topk_merge_strided_device(g2[0, 0].data_ptr(), g2[0, 1].data_ptr(), 2 * nq * k, G, nq, k,
                                           b["oidx"].data_ptr(), b["osc"].data_ptr(), stream)
            return b["oidx"], b["osc"]

        if query_norm_none:
            self.g.set_option("query_norm_override", _lib.NORM_NONE)
        try:
            stage1 = stage2 = None
            for i, q in enumerate(batches):
                slot, nq = i & 1, q.shape[0]
                self.g.set_option("workspace_slot", slot)
                b = self._buffers(nq, k, q.device, slot)
                self.g.phase1_device(q.data_ptr(), nq, k, b["approx"].data_ptr(), stream)
                g1, w1 = gather_async(b["approx"])
                done = None
                if stage1 is not None:
                    nxt = finish1(stage1)
                    if stage2 is not None:
                        done = finish2(stage2)
                    stage2 = nxt
                stage1 = (slot, b, nq, g1, w1)
                if done is not None:
                    yield done
            if stage1 is not None:
                nxt = finish1(stage1)
                if stage2 is not None:
                    yield finish2(stage2)
                yield finish2(nxt)
            elif stage2 is not None:
                yield finish2(stage2)
        finally:
            self.g.set_option("workspace_slot", 0)
            if query_norm_none:
                self.g.set_option("query_norm_override", -1)

    def any_flag(self):
        """True on every rank iff any shard raised a sticky flag since the last call (synchronises; the handle's
        statistics accumulators are left alone)."""
        bad = 1 if self.g.flags() else 0
        if self._protocol:
            import torch
            import torch.distributed as dist
            dev = "cuda" if dist.get_backend(self.group) == "nccl" else "cpu"
            t = torch.tensor([bad], dtype=torch.int32, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            bad = int(t.item())
        return bad > 0

    def _search(self, q, k):
        import torch
        nq = q.shape[0]
        b = self._ring_buffers(nq, k, q.device)
        stream = torch.cuda.current_stream().cuda_stream
        if not self._protocol:
            self.g.search_device(q.data_ptr(), nq, k, b["oidx"].data_ptr(), b["osc"].data_ptr(), None, stream)
            if self._join:
                self.g.join(stream)
            return b["oidx"], b["osc"]
        self.g.phase1_device(q.data_ptr(), nq, k, b["approx"].data_ptr(), stream)
        gathered = all_gather_stacked(b["approx"], self.group)
        _lib.kth_of_gathered_device(gathered.data_ptr(), self.world, nq, k, b["L"].data_ptr(), stream)
        pack = b["pack"]
        self.g.phase2_device(nq, k, b["L"].data_ptr(), pack[1].data_ptr(), b["sc"].data_ptr(), pack[0].data_ptr(),
                             stream)
        gathered2 = all_gather_stacked(pack, self.group)                 # [world, 2, nq, k]
        _lib.topk_merge_strided_device(gathered2[0, 0].data_ptr(), gathered2[0, 1].data_ptr(), 2 * nq * k, self.world,
                                       nq, k, b["oidx"].data_ptr(), b["osc"].data_ptr(), stream)
        return b["oidx"], b["osc"]

    def search_timed(self, q, k):
        """One synchronous search of the two-phase protocol with an event behind every stage on the caller's stream:
        returns ((idx, score), {"phase1", "allgather1", "kth", "phase2", "allgather2", "merge", "total"} in ms).  A collective
        of torch.distributed runs on the communicator's own stream and the caller's stream waits for it, so the interval up
        to the next event is what the collective costs THIS rank, waiting for its slowest peer included.  Synchronises.
        Diagnostic (bench.py reports the maxima over the ranks): what the RCCL all-gathers cost beside the kernels."""
        import torch
        if not self._protocol:
            raise RuntimeError("search_timed: the two-phase protocol is not active (one rank, no force_protocol)")
        nq = q.shape[0]
        b = self._buffers(nq, k, q.device)
        st = torch.cuda.current_stream()
        stream = st.cuda_stream
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(7)]
        ev[0].record(st)
        self.g.phase1_device(q.data_ptr(), nq, k, b["approx"].data_ptr(), stream)
        ev[1].record(st)
        gathered = all_gather_stacked(b["approx"], self.group)
        ev[2].record(st)
        _lib.kth_of_gathered_device(gathered.data_ptr(), self.world, nq, k, b["L"].data_ptr(), stream)
        ev[3].record(st)
        pack = b["pack"]
        self.g.phase2_device(nq, k, b["L"].data_ptr(), pack[1].data_ptr(), b["sc"].data_ptr(), pack[0].data_ptr(), stream)
        ev[4].record(st)
        gathered2 = all_gather_stacked(pack, self.group)
        ev[5].record(st)
        _lib.topk_merge_strided_device(gathered2[0, 0].data_ptr(), gathered2[0, 1].data_ptr(), 2 * nq * k, self.world,
                                       nq, k, b["oidx"].data_ptr(), b["osc"].data_ptr(), stream)
        ev[6].record(st)
        ev[6].synchronize()
        names = ("phase1", "allgather1", "kth", "phase2", "allgather2", "merge")
        ms = {n: ev[i].elapsed_time(ev[i + 1]) for i, n in enumerate(names)}
        ms["total"] = ev[0].elapsed_time(ev[6])
        return (b["oidx"], b["osc"]), ms

    def aqe_search(self, ranks, k_qe, w, k, eps=1e-6, join=True, verify=False):
        """alpha-QE across shards (src/utils/Reranking.py:195-208).  The k_qe rows of a query may live on any shard, and the
        expanded query must not depend on where: every rank writes the f32 rows IT owns of the k_qe x Q requested ones into a
        [k_qe, Q, D] block (zeros elsewhere, `mi_aqe_rows_device`), the blocks are summed over the group -- each element has
        exactly one non-zero contributor, so the all-reduce is exact in whatever order the collective adds; 24 MiB at
        k_qe = 3, Q = 1024, D = 2048 -- and every rank adds the rows in j = 0 .. k_qe-1 order in float64 with the single-shard
        kernel's own step (`mi_aqe_combine_device`): the sum, and with it q', is the single-GPU one bit for bit.  (Round 3
        all-gathered world x Q x D float64 partial sums, 128 MiB at 8 ranks, and was only equal up to the order of the
        additions across shard boundaries.)  Every rank normalises redundantly (`mi_aqe_finish_device`) and the expanded
        queries go through the sharded search as they are (no second normalisation).
        ranks: int64 cuda tensor [K_in, Q] of GLOBAL row ids (any strides).  Returns (idx [Q,k], score [Q,k], q_exp f32 [Q,D]);
        idx / score are buffers of this object's alpha-QE ring (two per shape): intact through the next aqe_search and every
        search(), overwritten by the second-next aqe_search of the same shape."""
        import torch
        import torch.distributed as dist
        nq = ranks.shape[1]
        d = self.g.d
        stream = torch.cuda.current_stream().cuda_stream
        part = torch.empty((nq, d), dtype=torch.float64, device=ranks.device)
        if self._protocol:
            rows = torch.empty((k_qe, nq, d), dtype=torch.float32, device=ranks.device)
            self.g.aqe_rows_device(ranks.data_ptr(), ranks.stride(0), ranks.stride(1), nq, k_qe, rows.data_ptr(), stream)
            if dist.get_backend(self.group) == "nccl":
                dist.all_reduce(rows, op=dist.ReduceOp.SUM, group=self.group)
            else:                                   # gloo (CPU rehearsals and tests): reduce on the host
                h = rows.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
                rows.copy_(h)
            _lib.aqe_combine_device(rows.data_ptr(), nq, d, k_qe, w, part.data_ptr(), stream)
        else:
            self.g.aqe_partial_device(ranks.data_ptr(), ranks.stride(0), ranks.stride(1), nq, k_qe, w, part.data_ptr(),
                                      stream)
        qx = torch.empty((nq, d), dtype=torch.float32, device=ranks.device)
        _lib.aqe_finish_device(part.data_ptr(), nq, d, eps, qx.data_ptr(), None, stream)
        self._tag = "aqe"                  # the re-search writes into the alpha-QE ring: `ranks` (a search() result) stays intact
        try:
            idx, sc = self.search(qx, k, query_norm_none=True, join=join, verify=verify)
        finally:
            self._tag = "search"
        return idx, sc, qx


class MultiDeviceGallery:
    """ONE process, several GPUs of the node: the reference's drivers are single processes (src/offline.py, src/online.py,
    src/test_rOP1m.py), so a drop-in `matching_HIP(..., devices=[0, 1, ...])` cannot rely on a launcher.  The gallery rows are
    split contiguously over `devices` (shard_bounds), one handle per device, and a query batch runs the same two-phase
    protocol as ShardedGallery -- phase 1 on every device, the K-th largest approximate score of the union, phase 2 on
    every device, merge -- with device-to-device copies (peer transfers over xGMI, stream-ordered by torch) where the
    one-process-per-GPU form has its RCCL all-gathers.  Answers are those of one single-device gallery, bit for bit
    (float32 queries; float64 queries are rounded to float32 first: the phase entry points take float32 device rows).
    The benchmark and the scaling runs use the process-per-GPU form (bench.py); this class is the convenience for callers
    that cannot be started N times."""

    def __init__(self, shards, devices):
        self.shards, self.devices = list(shards), [int(d) for d in devices]
        self.n, self.d = sum(s.n for s in self.shards), self.shards[0].d
        self.norm_mode = self.shards[0].norm_mode
        # one error margin and one image element type for all shards (ShardedGallery._agree, DESIGN section 4 "Shards")
        f16 = min(int(s.get_option("image_dtype")) for s in self.shards)
        for s in self.shards:
            if int(s.get_option("image_dtype")) != f16:
                s.set_image_dtype(f16)
        bounds = [max(v) for v in zip(*[s.norm_bounds() for s in self.shards])]
        for s in self.shards:
            s.norm_bounds(raise_to=bounds)
            s.set_option("rescore_grid_x", max(8, min(64, round(96 / len(self.shards)))))

    @classmethod
    def from_host(cls, rows, devices, norm_mode=_lib.NORM_L2, k_max=1):
        """rows: [N, D] float32 / float64 host array, any strides (`vecs.T` of the reference's [D, N]).
        k_max: the largest K the gallery will be searched with.  The shards answer phase 1 with their own K best rows, so a
        shard must hold at least K rows: trailing devices are left out (with a note) until every shard does."""
        n = rows.shape[0]
        devices = list(devices)
        fit = max(1, min(len(devices), n // max(1, int(k_max))))
        if fit < len(devices):
            print(">> MultiDeviceGallery: %d rows / K = %d: using %d of the %d listed devices (every shard must hold >= K rows)"
                  % (n, k_max, fit, len(devices)))
            devices = devices[:fit]
        shards = []
        try:
            for r, dev in enumerate(devices):
                lo, hi = shard_bounds(n, len(devices), r)
                if hi <= lo:
                    raise ValueError("more devices than gallery rows")
                shards.append(_lib.Gallery.from_host(rows[lo:hi], norm_mode=norm_mode, device=int(dev), row_offset=lo))
        except Exception:
            for s in shards:
                s.close()
            raise
        return cls(shards, devices)

    def close(self):
        for s in self.shards:
            s.close()
        self.shards = []

    def _batch(self, qb, k):
        import torch
        G, nq = len(self.shards), qb.shape[0]
        dev0 = torch.device("cuda", self.devices[0])
        approx, packs = [], []
        for s, dv in zip(self.shards, self.devices):
            dev = torch.device("cuda", dv)
            with torch.cuda.device(dev):
                q = torch.from_numpy(qb).to(dev)
                a = torch.empty((nq, k), dtype=torch.float32, device=dev)
                s.phase1_device(q.data_ptr(), nq, k, a.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
                approx.append(a)
        with torch.cuda.device(dev0):
            gathered = torch.stack([a.to(dev0) for a in approx]).contiguous()            # [G, nq, k]: the "all-gather"
            L0 = torch.empty((nq,), dtype=torch.float32, device=dev0)
            _lib.kth_of_gathered_device(gathered.data_ptr(), G, nq, k, L0.data_ptr(),
                                        torch.cuda.current_stream(dev0).cuda_stream)
        for s, dv in zip(self.shards, self.devices):
            dev = torch.device("cuda", dv)
            with torch.cuda.device(dev):
                L = L0.to(dev)
                pack = torch.empty((2, nq, k), dtype=torch.int64, device=dev)            # f64 score bits, global row ids
                sc = torch.empty((nq, k), dtype=torch.float32, device=dev)
                s.phase2_device(nq, k, L.data_ptr(), pack[1].data_ptr(), sc.data_ptr(), pack[0].data_ptr(),
                                torch.cuda.current_stream(dev).cuda_stream)
                packs.append(pack)
        with torch.cuda.device(dev0):
            allp = torch.stack([p.to(dev0) for p in packs]).contiguous()                 # [G, 2, nq, k]
            oidx = torch.empty((nq, k), dtype=torch.int64, device=dev0)
            osc = torch.empty((nq, k), dtype=torch.float32, device=dev0)
            _lib.topk_merge_strided_device(allp[0, 0].data_ptr(), allp[0, 1].data_ptr(), 2 * nq * k, G, nq, k,
                                           oidx.data_ptr(), osc.data_ptr(), torch.cuda.current_stream(dev0).cuda_stream)
            torch.cuda.synchronize(dev0)
        return oidx.cpu().numpy(), osc.cpu().numpy()

    def aqe_search(self, ranks, k_qe, w, k, eps=1e-6):
        """alpha-QE + re-search on the shards of one process (src/utils/Reranking.py:195-208): ranks int64 host array
        [K_in, Q] of global row ids (any strides).  Every device writes the rows it owns (zeros elsewhere), device 0 adds the
        blocks -- one non-zero contributor per element: exact -- and sums them in j order with the single-shard kernel's step,
        so q' and the re-search are those of one handle, bit for bit.  Returns (idx [Q, k], score [Q, k], q_exp f32 [Q, D])."""
        import torch
        ranks = np.ascontiguousarray(np.asarray(ranks)[:k_qe], dtype=np.int64)
        nq = ranks.shape[1]
        out_i = np.empty((nq, k), dtype=np.int64)
        out_s = np.empty((nq, k), dtype=np.float32)
        out_q = np.empty((nq, self.d), dtype=np.float32)
        dev0 = torch.device("cuda", self.devices[0])
        for q0 in range(0, nq, 1024):
            rb = np.ascontiguousarray(ranks[:, q0:q0 + 1024])
            b = rb.shape[1]
            total = None
            for s, dv in zip(self.shards, self.devices):
                dev = torch.device("cuda", dv)
                with torch.cuda.device(dev):
                    r = torch.from_numpy(rb).to(dev)
                    rows = torch.empty((k_qe, b, self.d), dtype=torch.float32, device=dev)
                    s.aqe_rows_device(r.data_ptr(), r.stride(0), r.stride(1), b, k_qe, rows.data_ptr(),
                                      torch.cuda.current_stream(dev).cuda_stream)
                    torch.cuda.synchronize(dev)
                with torch.cuda.device(dev0):
                    rows0 = rows.to(dev0)
                    total = rows0 if total is None else total.add_(rows0)
            with torch.cuda.device(dev0):
                st0 = torch.cuda.current_stream(dev0).cuda_stream
                part = torch.empty((b, self.d), dtype=torch.float64, device=dev0)
                qx = torch.empty((b, self.d), dtype=torch.float32, device=dev0)
                _lib.aqe_combine_device(total.data_ptr(), b, self.d, k_qe, w, part.data_ptr(), st0)
                _lib.aqe_finish_device(part.data_ptr(), b, self.d, eps, qx.data_ptr(), None, st0)
                torch.cuda.synchronize(dev0)
                out_q[q0:q0 + b] = qx.cpu().numpy()
        for s in self.shards:
            s.set_option("query_norm_override", _lib.NORM_NONE)
        try:
            out_i, out_s = self.search(out_q, k)
        finally:
            for s in self.shards:
                s.set_option("query_norm_override", -1)
        return out_i, out_s, out_q

    def search(self, queries, k):
        """queries: [Q, D] host array -> (idx int64 [Q, k] of GLOBAL row ids, score float32 [Q, k]); any Q (batches of 1024).
        A batch that raised a sticky flag on any shard (failed speculative threshold, candidate-buffer overflow) is answered
        again with the fallbacks of the host entry point, like ShardedGallery.search(verify=True)."""
        q = np.ascontiguousarray(np.asarray(queries), dtype=np.float32)
        if q.ndim != 2 or q.shape[1] != self.d:
            raise ValueError("queries must be [Q, %d]" % self.d)
        if k > self.n:
            raise RuntimeError("mi355_retrieval error 1: k > number of gallery rows")
        if k > min(s.n for s in self.shards):
            raise RuntimeError("multi-device search: k = %d exceeds the rows of the smallest shard (%d); create the gallery "
                               "with from_host(..., k_max=k) so that it uses fewer devices" % (k, min(s.n for s in self.shards)))
        out_i = np.empty((q.shape[0], k), dtype=np.int64)
        out_s = np.empty((q.shape[0], k), dtype=np.float32)
        for q0 in range(0, q.shape[0], 1024):
            qb = q[q0:q0 + 1024]
            idx, sc = self._batch(qb, k)
            if any([s.flags() for s in self.shards]):          # a list: every shard's flags are read (and cleared)
                for name, val in (("speculative", 0), ("force_exact", 1)):
                    prev = [s.get_option(name) for s in self.shards]
                    for s in self.shards:
                        s.set_option(name, val)
                    try:
                        idx, sc = self._batch(qb, k)
                        bad = any([s.flags() for s in self.shards])
                    finally:
                        for s, p in zip(self.shards, prev):
                            s.set_option(name, p)
                    if not bad:
                        break
                else:
                    raise RuntimeError("multi-device search: candidate buffers overflow even with the f32 scorer")
            out_i[q0:q0 + qb.shape[0]], out_s[q0:q0 + qb.shape[0]] = idx, sc
        return out_i, out_s
