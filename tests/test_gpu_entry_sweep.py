"""GPU: the other entry points of the path on data with structure -- alpha-QE on clustered galleries, full-length rankings
with ties / zero rows, galleries grown in ragged blocks and reloaded, non-finite inputs.  Every answer against the oracle
(float64 where an order is judged, the reference's own dtype where values are compared).
"""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu

TAU = 1e-6


@pytest.fixture(scope="module")
def lib():
    from isehr_amd import _lib
    _lib.load()
    return _lib


def _clustered(rng, n, d, ncl, noise):
    c = rng.standard_normal((ncl, d))
    lab = rng.integers(0, ncl, n)
    g = c[lab] + noise * rng.standard_normal((n, d))
    return g.astype(np.float32), c, lab


@pytest.mark.parametrize("nq", [30, 200])
def test_alpha_qe_on_clustered_gallery(lib, nq):
    """feature_enhancement + re-search (src/utils/Reranking.py:195-208) where the k_qe neighbours are near-duplicates of one
    another: the expanded query must equal the oracle's (float64 sum of the gathered rows), and the re-search is judged
    on the float64 scores of THAT query."""
    from isehr_amd._lib import Gallery
    rng = np.random.default_rng(3)
    n, d, k = 50000, 160, 100
    g, c, _ = _clustered(rng, n, d, 40, 0.15)
    q = (c[rng.integers(0, 40, nq)] + 0.1 * rng.standard_normal((nq, d))).astype(np.float32)
    G = Gallery.from_host(g)
    try:
        idx, _, _ = G.search(q, k)
        idx2, sc2, qx = G.aqe_search(np.ascontiguousarray(idx.T), 3, 4.0, k, return_qexp=True)[:3]
        stored = G.get_rows(0, n)
    finally:
        G.close()
    assert oracle.check_topk_parity(idx, oracle.exact_scores_f64(g, q), k, TAU) == []
    # expanded query: weighted sum of the STORED (f32-normalised) rows, in float64, / (norm + 1e-6)
    qx_ref, _ = oracle.feature_enhancement(3, idx.T, stored.T.astype(np.float64), 4.0)
    assert np.abs(qx - qx_ref.T).max() < 2e-7
    s2 = stored.astype(np.float64) @ qx.astype(np.float64).T                 # [N, Q] scores of the expanded queries as used
    assert oracle.check_topk_parity(idx2, s2.T, k, TAU) == []
    assert np.abs(np.take_along_axis(s2.T, idx2, 1) - sc2).max() < 3e-7


def test_full_ranking_with_zero_rows_and_block_ties(lib):
    """np.argsort(-scores) over ALL rows (src/main_retrieve.py:176) when the gallery holds zero rows (cosine undefined: NaN
    in the reference, ranked last there) and blocks of identical rows (stable order = index order)."""
    from isehr_amd._lib import Gallery
    rng = np.random.default_rng(8)
    n, d, nq = 6000, 72, 5
    g = rng.standard_normal((n, d)).astype(np.float32)
    g[1000:1300] = g[999]                                     # 301 identical rows
    zero = np.array([5, 2500, 5999])
    g[zero] = 0.0
    q = rng.standard_normal((nq, d)).astype(np.float32)
    G = Gallery.from_host(g)
    try:
        ranks, sc, _ = G.rank_all(q, return_scores=True)
        pos = G.rank_positions(q, np.tile(np.array([999, 1000, 1299, 5, 17]), (nq, 1)))
        pre = G.rank_prefix(q, 700)[0]
    finally:
        G.close()
    assert ranks.shape == (nq, n)
    for j in range(nq):
        assert np.array_equal(np.sort(ranks[j]), np.arange(n))
        assert set(ranks[j, -3:]) == set(zero)                # undefined cosines go last
        blk = np.flatnonzero(np.isin(ranks[j], np.arange(999, 1300)))
        assert blk.max() - blk.min() == 300 and np.array_equal(ranks[j, blk], np.arange(999, 1300))
        assert np.array_equal(ranks[j, pos[j]], [999, 1000, 1299, 5, 17])
    assert np.array_equal(pre, ranks[:, :700])
    keep = np.setdiff1d(np.arange(n), zero)
    s = oracle.exact_scores_f64(g[keep], q)
    got = np.take_along_axis(s, np.searchsorted(keep, ranks[:, :n - 3]), 1)
    assert (np.diff(got, axis=1) <= 2e-6).all()               # sorted up to the f32 scorer's near-ties


def test_gallery_grown_in_ragged_blocks_saved_and_reloaded(lib, tmp_path):
    """extract -> append -> save -> load (src/utils/nnsearch.py:503-525 convention): blocks of 1, 255, 256, 257 ... rows,
    float64 and strided blocks, then the same answers from the reloaded file."""
    from isehr_amd._lib import Gallery
    rng = np.random.default_rng(12)
    d, k = 136, 60
    sizes = [1, 255, 256, 257, 1000, 3, 4097, 64, 9000]
    n = sum(sizes)
    g = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal((150, d)).astype(np.float32)
    G = Gallery.empty(n + 100, d)
    try:
        lo = 0
        for i, m in enumerate(sizes):
            blk = g[lo:lo + m]
            if i % 3 == 1:
                blk = blk.astype(np.float64)
            elif i % 3 == 2:
                blk = np.ascontiguousarray(blk.T).T           # the reference's [D, N] layout seen as rows
            G.append(blk)
            lo += m
        i1, s1, _ = G.search(q, k)
        path = str(tmp_path / "grown.bin")
        G.save(path)
    finally:
        G.close()
    W = Gallery.from_host(g)
    i0, s0, _ = W.search(q, k)
    W.close()
    assert np.array_equal(i1, i0) and np.array_equal(s1, s0)
    L = Gallery.load(path)
    try:
        i2, s2, _ = L.search(q, k)
        assert L.n == n
    finally:
        L.close()
    assert np.array_equal(i2, i0) and np.array_equal(s2, s0)
    assert oracle.check_topk_parity(i0, oracle.exact_scores_f64(g, q), k, TAU) == []


def test_non_finite_and_huge_inputs(lib):
    """Rows with NaN / Inf (a broken extractor run) must never enter a top-K nor disturb the other rows; float64 rows far
    outside the float32 range are normalised before they are stored, like `train / np.linalg.norm(train)` in float64."""
    from isehr_amd._lib import Gallery
    rng = np.random.default_rng(21)
    n, d, nq, k = 20000, 64, 33, 40
    g = rng.standard_normal((n, d))
    g[10:20] *= 1e100                                        # fine after the float64 normalisation
    g[30:35] *= 1e-100
    bad = np.array([100, 7000, 19999])
    g[bad[0], 3] = np.nan
    g[bad[1], 0] = np.inf
    g[bad[2]] = np.nan
    q = rng.standard_normal((nq, d))
    G = Gallery.from_host(g)
    try:
        idx, sc, _ = G.search(q, k)
    finally:
        G.close()
    assert not np.isin(idx, bad).any()
    assert np.isfinite(sc).all()
    ok = np.setdiff1d(np.arange(n), bad)
    s = oracle.exact_scores_f64(g[ok], q)
    assert oracle.check_topk_parity(np.searchsorted(ok, idx), s, k, TAU) == []


def test_concurrent_searches_from_threads(lib):
    """src/online.py:163 serves queries from a threaded Flask app: several threads search the same gallery (serialised per
    handle by the wrapper's lock) while others search a second gallery on the same device.  Every answer must equal the
    single-threaded one."""
    import threading
    from isehr_amd._lib import Gallery
    rng = np.random.default_rng(17)
    ga = rng.standard_normal((60000, 96)).astype(np.float32)
    gb = rng.standard_normal((45000, 160)).astype(np.float32)
    qa = [rng.standard_normal((n, 96)).astype(np.float32) for n in (1, 7, 130, 300)]
    qb = [rng.standard_normal((n, 160)).astype(np.float32) for n in (3, 64, 129, 257)]
    A, B = Gallery.from_host(ga), Gallery.from_host(gb)
    try:
        ref_a = [A.search(q, 50)[:2] for q in qa]
        ref_b = [B.search(q, 20)[:2] for q in qb]
        errors = []

        def worker(G, queries, refs, k, seed):
            try:
                order = np.random.default_rng(seed).permutation(len(queries) * 6) % len(queries)
                for j in order:
                    idx, sc, _ = G.search(queries[j], k)
                    if not (np.array_equal(idx, refs[j][0]) and np.array_equal(sc, refs[j][1])):
                        errors.append((seed, int(j)))
            except Exception as e:                           # noqa: BLE001
                errors.append((seed, repr(e)))

        threads = [threading.Thread(target=worker, args=(A, qa, ref_a, 50, s)) for s in range(3)]
        threads += [threading.Thread(target=worker, args=(B, qb, ref_b, 20, 10 + s)) for s in range(3)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=300)
        assert not any(t.is_alive() for t in threads)
        assert errors == []
    finally:
        A.close()
        B.close()
    assert oracle.check_topk_parity(ref_a[2][0], oracle.exact_scores_f64(ga, qa[2]), 50, TAU) == []


def test_handles_release_their_device_memory(lib):
    """A server that rebuilds its gallery (--ifgenerate) must not leak: create / search (every path: filtered, exact, dense
    fallback, full ranking, alpha-QE, diffusion) / close in a loop and compare the free device memory."""
    import torch
    from isehr_amd._lib import Gallery, NORM_NONE
    rng = np.random.default_rng(2)
    g = rng.standard_normal((40000, 128)).astype(np.float32)
    q = rng.standard_normal((200, 128)).astype(np.float32)
    small = rng.standard_normal((1500, 32)).astype(np.float32)

    def cycle():
        G = Gallery.from_host(g)
        idx, _, _ = G.search(q, 50)
        G.aqe_search(np.ascontiguousarray(idx.T), 3, 4.0, 50)
        G.rank_prefix(q[:4], 300)
        G.dense_search(q[:8], 400)
        G.set_option("force_exact", 1)
        G.search(q[:16], 10)
        G.close()
        S = Gallery.from_host(small, norm_mode=NORM_NONE)
        S.diffusion_offline(128, 16)
        S.diffusion_online(small[:3], 3, 3, 100)
        S.close()

    cycle()                                                  # first use: code objects, torch's own pools
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(12):
        cycle()
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < 32 * 2**20, (free0, free1)
