"""GPU: dense exact kNN and the truncated graph diffusion against the oracle's restatement of
src/utils/diffusion.py / src/utils/Reranking.py:230-253 and against the solutions of the reference's own
get_offline_result (tests/golden/diffusion_solve.npz).  Unpinned there: only the faiss search (absent); the oracle uses
exact numpy inner-product top-k in its place, see oracle/__init__.py."""
import numpy as np
import pytest

import oracle
from isehr_amd.synth import synth_rows, planted_dataset

pytestmark = pytest.mark.gpu


def _features(seed, n, d):
    f = synth_rows(seed, 0, n, d).astype(np.float64)
    # clustered data: a graph with real mutual neighbours
    centers = synth_rows(seed + 1, 0, 12, d).astype(np.float64)
    f = 0.9 * f + 2.0 * centers[np.arange(n) % 12]
    f /= np.linalg.norm(f, axis=1, keepdims=True)
    return f.astype(np.float32)


def test_dense_search_large_k():
    from isehr_amd._lib import Gallery, NORM_NONE
    f = _features(3, 1500, 96)
    g = Gallery.from_host(f, norm_mode=NORM_NONE)
    idx, sc, _ = g.dense_search(f[:40], 1000)
    g.close()
    s64 = f[:40].astype(np.float64) @ f.astype(np.float64).T
    assert oracle.check_topk_parity(idx, s64, 1000, 2e-6) == []
    assert (np.diff(sc, axis=1) <= 0).all()
    rs, ri = oracle.knn_flat_ip(f, f[:40], 1000)
    assert np.abs(sc - rs).max() < 2e-6
    assert (idx == ri).mean() > 0.98
    # massive ties (duplicated rows) are broken by the lower index, exactly k results, no duplicates
    f2 = np.repeat(f[:50], 30, axis=0)
    g = Gallery.from_host(f2, norm_mode=NORM_NONE)
    idx, sc, _ = g.dense_search(f2[:3], 100)
    g.close()
    assert all(len(set(r)) == 100 for r in idx)
    assert (idx[0, :30] == np.arange(30)).all()


@pytest.mark.parametrize("n,trunc,kd", [(700, 300, 50), (1200, 500, 80)])
def test_diffusion_offline_matches_oracle(n, trunc, kd):
    from isehr_amd.diffusion import Diffusion
    f = _features(5, n, 64)
    d = Diffusion(f)
    off = d.get_offline_results(trunc, kd)
    ids, vals, sims = d.gallery.diffusion_offline(trunc, kd, return_sims=True)
    d.close()
    ref_off, ref_sims, ref_ids, lap, ref_scores = oracle.diffusion_offline(f, trunc, kd, return_parts=True)
    # kNN graph: same neighbour sets up to f32 near-ties
    s64 = f.astype(np.float64) @ f.astype(np.float64).T
    assert oracle.check_topk_parity(ids, s64, trunc, 2e-6) == []
    assert np.abs(sims - ref_sims).max() < 2e-6
    # offline matrix: compare as dense (a near-tie at the truncation boundary may swap one column)
    a = np.asarray(off.todense(), dtype=np.float64)
    b = np.asarray(ref_off.todense(), dtype=np.float64)
    same_support = (ids == ref_ids).all(axis=1)
    assert same_support.mean() > 0.95
    err = np.abs(a[same_support] - b[same_support]).max()
    assert err < 5e-5, err
    assert np.abs(b).max() > 0.5        # the diagonal-ish entries are O(1): the tolerance is meaningful
    assert off.dtype == np.float32 and off.shape == (n, n) and off.nnz <= n * trunc


def test_diffusion_online_and_qge_small():
    from isehr_amd.reranking import QGE_hip
    from isehr_amd import evaluate
    vecs, qv, gnd = planted_dataset(51, 900, 64, 10)
    base = oracle.matching_l2(100, vecs.T, qv.T).T
    out = QGE_hip(base, qv, vecs, "roxford5k-synthetic", gnd, AQE=True, K=100, quiet=True)
    trunc = 899
    qx, ranks_aqe, ranks_dfs = oracle.qge_small(base, qv, vecs, True, truncation_number=trunc, k_gallery=200, k_query=3)
    assert np.abs(out["qvecs_qe"] - qx).max() < 1e-7
    assert (out["ranks_aqe"] == ranks_aqe[:100]).mean() > 0.99
    got = out["ranks_dfs"]
    assert got.shape == (trunc, 10)
    # diffusion scores are small positive numbers with many exact zeros (ties): compare the clearly ordered head
    ref_map = oracle.compute_map_revisited(ranks_dfs, gnd)
    got_map = evaluate.compute_map_revisited(got, gnd)
    assert np.allclose(got_map, ref_map, atol=2e-3), (got_map, ref_map)
    head = 20
    agree = np.mean([len(set(got[:head, q]) & set(ranks_dfs[:head, q])) / head for q in range(10)])
    assert agree > 0.95, agree


def test_diffusion_cache_roundtrip(tmp_path):
    from isehr_amd.diffusion import Diffusion
    f = _features(7, 600, 32)
    d = Diffusion(f, str(tmp_path))
    off1 = d.get_offline_results(200, 40)
    r1, s1 = d.search_online(f[:5], 3, 200)
    d.close()
    d2 = Diffusion(f, str(tmp_path))
    off2 = d2.get_offline_results(200, 40)          # from offline.jbl
    r2, s2 = d2.search_online(f[:5], 3, 200)
    d2.close()
    assert (off1 != off2).nnz == 0
    assert np.array_equal(r1, r2) and np.array_equal(s1, s2)


def test_diffusion_offline_at_the_reference_size():
    """The only size the reference ever diffuses at: rOxford5k, N = 4993, truncation 2000, k_gallery 200
    (src/utils/Reranking.py:230-236; N >= 120000 skips diffusion, :212).  About 20 s of scipy on the host."""
    import time
    from isehr_amd.diffusion import Diffusion
    n, d, T, kd = 4993, 128, 2000, 200
    f = synth_rows(5, 0, n, d).astype(np.float64)
    c = synth_rows(6, 0, 40, d).astype(np.float64)
    f = 0.8 * f + 1.5 * c[np.arange(n) % 40]
    f /= np.linalg.norm(f, axis=1, keepdims=True)
    f = f.astype(np.float32)
    dd = Diffusion(f)
    t0 = time.time()
    ids, vals, sims = dd.gallery.diffusion_offline(T, kd, return_sims=True)
    t_gpu = time.time() - t0
    dd.close()
    t0 = time.time()
    ref_off, ref_sims, ref_ids, lap, ref_scores = oracle.diffusion_offline(f, T, kd, return_parts=True)
    t_cpu = time.time() - t0
    s64 = f.astype(np.float64) @ f.astype(np.float64).T
    assert oracle.check_topk_parity(ids, s64, T, 2e-6) == []
    assert np.abs(sims - ref_sims).max() < 2e-6
    same = (ids == ref_ids).all(axis=1)
    assert same.mean() > 0.99, same.mean()
    err = np.abs(vals[same].astype(np.float64) - ref_scores[same]).max()
    assert err < 5e-6, err
    assert np.abs(ref_scores).max() > 0.5
    print("diffusion offline N=%d T=%d kd=%d: GPU %.2f s, oracle %.1f s, identical supports %.4f, max |d| %.2e"
          % (n, T, kd, t_gpu, t_cpu, same.mean(), err))


def test_offline_diffusion_by_node_ranges():
    """mi_diffusion_offline_nodes: any partition of the nodes gives the rows of the full result, bit for bit (the CG solves
    are independent, src/utils/diffusion.py:15-19); an empty range is legal."""
    from isehr_amd._lib import Gallery, NORM_NONE
    from isehr_amd.synth import synth_rows
    n, d, T, kd = 1203, 40, 256, 32
    f = synth_rows(15, 0, n, d) * 0.7 + 1.1 * synth_rows(16, 0, 25, d)[np.arange(n) % 25]
    f = (f / np.linalg.norm(f, axis=1, keepdims=True)).astype(np.float32)
    G = Gallery.from_host(f, norm_mode=NORM_NONE)
    try:
        ids, vals = G.diffusion_offline(T, kd)
        parts = []
        for lo, hi in ((0, 1), (1, 400), (400, 400), (400, 1203)):
            pid, pv = G.diffusion_offline_nodes(T, kd, lo, hi)
            assert np.array_equal(pid, ids) and pv.shape == (hi - lo, T)
            parts.append(pv)
        assert np.array_equal(np.concatenate(parts), vals)
        with pytest.raises(RuntimeError):
            G.diffusion_offline_nodes(T, kd, 5, n + 1)
    finally:
        G.close()


def test_diffusion_offline_vs_reference_solve_golden(golden_dir):
    """The truncated CG solutions the reference's own get_offline_result (src/utils/diffusion.py:15-19) produced for 50
    nodes of a 300-row clustered feature set (oracle/make_golden.py; n_trunc = 200, kd = 40, Laplacian from the reference's
    get_laplacian) against mi_diffusion_offline: same truncation supports, values within float32 rounding of the f64
    solution (the reference stores float32 too, :80-84)."""
    import os
    from isehr_amd.diffusion import Diffusion
    z = np.load(os.path.join(golden_dir, "diffusion_solve.npz"))
    T, kd, nodes = int(z["n_trunc"]), int(z["kd"]), z["nodes"]
    vd = synth_rows(61, 0, 300, 24).astype(np.float64)
    cd = synth_rows(62, 0, 12, 24).astype(np.float64)
    vd = 0.8 * vd + 1.1 * cd[np.arange(300) % 12]
    vd /= np.linalg.norm(vd, axis=1, keepdims=True)
    f = vd.astype(np.float32)
    d = Diffusion(f)
    try:
        ids, vals = d.gallery.diffusion_offline(T, kd)
    finally:
        d.close()
    same = (ids[nodes] == z["ids"]).all(axis=1)
    assert same.mean() >= 0.9, same.mean()        # a float32 near-tie between two neighbours may swap two columns of a row
    ref = z["scores"].astype(np.float32)
    err = np.abs(vals[nodes][same].astype(np.float64) - z["scores"][same])
    assert err.max() < 2e-6, err.max()
    assert np.abs(ref).max() > 1.0
    # rows with a swapped near-tie: equal as sparse rows (values matched by neighbour id)
    for r in np.flatnonzero(~same):
        i = nodes[r]
        got = dict(zip(ids[i].tolist(), vals[i].tolist()))
        want = dict(zip(z["ids"][r].tolist(), z["scores"][r].tolist()))
        common = set(got) & set(want)
        assert len(common) >= T - 2
        assert max(abs(got[c] - want[c]) for c in common) < 5e-5
