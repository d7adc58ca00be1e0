"""CPU, world_size 2 over gloo: the host orchestration of the sharded path -- row partition, the two
all-gathers and the globalised indices -- with the oracle standing in for the per-shard GPU phases
(the HIP phases themselves are covered by tests/test_gpu_rerank_and_shards.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from isehr_amd.sharded import shard_bounds, all_gather_stacked
from isehr_amd.synth import synth_rows


def test_shard_bounds_cover_rows_exactly():
    for n in (1, 7, 100, 1005994):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c and a <= b
            assert sum(b - a for a, b in spans) == n


def _worker(rank, world, port, n, d, nq, k, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = synth_rows(5, 0, n, d)
        q = synth_rows(6, 0, nq, d)
        lo, hi = shard_bounds(n, world, rank)
        # phase 1 stand-in: the shard's k largest scores
        s = oracle.exact_scores_f64(g[lo:hi], q)
        kl = min(k, hi - lo)
        top = -np.sort(-s, axis=1)[:, :kl]
        approx = np.full((nq, k), -np.inf, dtype=np.float32)
        approx[:, :kl] = top
        gathered = all_gather_stacked(torch.from_numpy(approx))
        assert gathered.shape == (world, nq, k)
        L = np.sort(gathered.numpy().transpose(1, 0, 2).reshape(nq, -1), axis=1)[:, -k]
        # phase 2 stand-in: local exact top-k with GLOBAL ids
        li, ls = oracle.exact_topk_f64(g[lo:hi], q, kl)
        idx = np.full((nq, k), -1, dtype=np.int64)
        sc = np.full((nq, k), -np.inf)
        idx[:, :kl] = li + lo
        sc[:, :kl] = ls
        # protocol invariant: L is the K-th best score of the whole gallery, so every shard's local list holds
        # every row of the global top-k that lives on it, i.e. at least one shard's best is >= L for each query
        best = all_gather_stacked(torch.from_numpy(np.ascontiguousarray(sc[:, :1]))).numpy().max(axis=0)[:, 0]
        assert (best >= L - 1e-6).all()
        ref_s_all = oracle.exact_topk_f64(g, q, k)[1]
        assert np.abs(L - ref_s_all[:, k - 1]).max() < 1e-6
        g_sc = all_gather_stacked(torch.from_numpy(sc)).numpy()
        g_idx = all_gather_stacked(torch.from_numpy(idx)).numpy()
        ms, mi = oracle.merge_topk(list(g_sc), list(g_idx), k)
        ref_i, ref_s = oracle.exact_topk_f64(g, q, k)
        ok = np.array_equal(mi, ref_i) and np.allclose(ms, ref_s)
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_exchange_matches_global_topk():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    ret = mgr.dict()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 301, 32, 5, 20, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(100)
        assert p.exitcode == 0
    assert ret.get(0) is True and ret.get(1) is True


def test_job_layout_covers_every_query_and_row_once():
    """isehr_amd.sharded.job_layout: for every world size and layout the (query group, row shard) pairs of the ranks tile
    the job exactly once; 'auto' is 2 x N/2 for an even N and batches of >= 512 queries."""
    from isehr_amd.sharded import job_layout, shard_bounds
    for world in (1, 2, 3, 4, 6, 8):
        for layout in ["auto", "1x%d" % world] + (["2x%d" % (world // 2)] if world % 2 == 0 else []):
            nq, n = 1024, 1005994
            cells = [job_layout(world, r, nq, layout) for r in range(world)]
            gq, gs = cells[0][0], cells[0][1]
            assert gq * gs == world and all(c[:2] == (gq, gs) for c in cells)
            assert sorted((c[2], c[3]) for c in cells) == [(a, b) for a in range(gq) for b in range(gs)]
            if layout == "auto":
                assert gq == (2 if world % 2 == 0 else 1)
            covered = sum(shard_bounds(n, gs, b)[1] - shard_bounds(n, gs, b)[0] for b in range(gs))
            assert covered == n and nq % gq == 0
    assert job_layout(8, 5, 100, "auto")[:2] == (1, 8)          # small batches: row shards only
    import pytest
    with pytest.raises(ValueError):
        job_layout(4, 0, 1024, "3x1")


def _worker_layout(rank, world, port, layout, n, d, nq, k, ret):
    """One rank of a gq x gs job: the rank's query group answers ITS slice of the batch against the gallery sharded over
    the group's ranks, with the collectives of ShardedGallery (all_gather_stacked on the group) and the oracle standing in
    for the per-shard GPU phases; plus the fixed-order sum of the sharded alpha-QE."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from isehr_amd.sharded import job_layout, layout_groups
        gq, gs, qgroup, shard = job_layout(world, rank, nq, layout)
        groups = layout_groups(gq, gs)                       # collective: every rank creates every group
        group = groups[qgroup]
        assert dist.get_world_size(group) == gs and dist.get_rank(group) == shard
        g = synth_rows(5, 0, n, d)
        q_all = synth_rows(6, 0, nq, d)
        per = nq // gq
        q = q_all[qgroup * per:(qgroup + 1) * per]
        lo, hi = shard_bounds(n, gs, shard)
        li, ls = oracle.exact_topk_f64(g[lo:hi], q, min(k, hi - lo))
        idx = np.full((per, k), -1, dtype=np.int64)
        sc = np.full((per, k), -np.inf)
        idx[:, :li.shape[1]] = li + lo
        sc[:, :ls.shape[1]] = ls
        g_sc = all_gather_stacked(torch.from_numpy(sc), group).numpy()
        g_idx = all_gather_stacked(torch.from_numpy(idx), group).numpy()
        assert g_sc.shape == (gs, per, k)
        ms, mi = oracle.merge_topk(list(g_sc), list(g_idx), k)
        ref_i, ref_s = oracle.exact_topk_f64(g, q, k)
        ok = np.array_equal(mi, ref_i) and np.allclose(ms, ref_s)
        # sharded alpha-QE: float64 partial sums of the rows each shard owns, all-gathered and added in rank order
        w = (np.arange(3, 0, -1) / 3.0) ** 4.0
        part = np.zeros((per, d))
        for j in range(3):
            rows = ref_i[:, j]
            mine = (rows >= lo) & (rows < hi)
            part[mine] += g[rows[mine]].astype(np.float64) * w[j]
        parts = all_gather_stacked(torch.from_numpy(part), group).numpy()
        total = parts[0].copy()
        for r in range(1, gs):
            total += parts[r]
        full = sum(g[ref_i[:, j]].astype(np.float64) * w[j] for j in range(3))
        ok = ok and np.abs(total - full).max() < 1e-12
        # every rank of the group holds the same bits (the point of the fixed order)
        chk = all_gather_stacked(torch.from_numpy(total), group).numpy()
        ok = ok and all(np.array_equal(chk[0], chk[r]) for r in range(gs))
        ret[rank] = (bool(ok), gq, gs, qgroup, shard)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world,layout", [(8, "2x4"), (8, "1x8"), (4, "2x2")])
def test_query_groups_times_row_shards_over_gloo(world, layout):
    """The rank count of the target node (8 processes) in the layouts bench.py --gpus 8 uses: 2 query groups x 4 row shards
    for the headline, 1 x 8 for the 10 M-row block -- group creation, collectives inside a group, completeness."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    ret = mgr.dict()
    procs = [ctx.Process(target=_worker_layout, args=(r, world, port, layout, 803, 24, 16, 10, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(250)
        assert p.exitcode == 0
    gq, gs = (int(v) for v in layout.split("x"))
    assert sorted((ret[r][3], ret[r][4]) for r in range(world)) == [(a, b) for a in range(gq) for b in range(gs)]
    assert all(ret[r][0] is True and ret[r][1:3] == (gq, gs) for r in range(world))


def _worker_aqe(rank, world, port, n, d, nq, k_qe, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = synth_rows(15, 0, n, d)
        g[7, :5] = [-0.0, 0.0, 1e-42, -1e-42, np.float32(3.0e38)]         # signed zeros, denormals, a huge value
        rng = np.random.default_rng(9)
        ranks = rng.integers(0, n, size=(k_qe, nq)).astype(np.int64)
        ranks[0, 0] = 7
        ranks[1, 1] = -3                                                   # nobody's row: stays zero
        lo, hi = shard_bounds(n, world, rank)
        # the shard's block (mi_aqe_rows_device): its own rows, zeros for everybody else's
        rows = np.zeros((k_qe, nq, d), dtype=np.float32)
        mine = (ranks >= lo) & (ranks < hi)
        rows[mine] = g[ranks[mine]]
        t = torch.from_numpy(rows)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)                           # ShardedGallery.aqe_search's exchange
        want = np.where(((ranks >= 0) & (ranks < n))[..., None], g[np.clip(ranks, 0, n - 1)], np.float32(0.0))
        # exact in any order: every element has one non-zero contributor (x + 0 == x); the sign of a zero is the one thing
        # that may differ, and the weighted f64 sum downstream treats both zeros alike
        ok = np.array_equal(t.numpy(), want)
        nz = want != 0
        ok = ok and np.array_equal(t.numpy().view(np.uint32)[nz], want.view(np.uint32)[nz])
        # the j-ordered f64 sum of the exchanged rows = the single-gallery sum (oracle: feature_enhancement's weights)
        wts = (np.arange(k_qe, 0, -1) / k_qe) ** 4.0
        got = np.zeros((nq, d))
        ref = np.zeros((nq, d))
        for j in range(k_qe):
            got += t.numpy()[j].astype(np.float64) * wts[j]
            ref += want[j].astype(np.float64) * wts[j]
        ret[rank] = bool(ok and np.array_equal(got, ref))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
@pytest.mark.parametrize("world", [2, 3])
def test_aqe_row_exchange_is_exact_over_gloo(world):
    """Sharded alpha-QE (round 4): the shards exchange ROWS by an all-reduce in which every element has exactly one non-zero
    contributor -- exact whatever order the collective adds in -- so the j-ordered float64 sum is that of ONE gallery."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ret = mp.Manager().dict()
    mp.spawn(_worker_aqe, args=(world, port, 1000, 48, 33, 3, ret), nprocs=world, join=True)
    assert all(ret[r] for r in range(world)) and len(ret) == world
