"""GPU: alpha-QE, raw inner-product ranking / KNN wrapper, eps-normalisation, persistence and the sharded
two-phase protocol (simulated with several shard handles on one GPU), all through the C ABI."""
import os

import numpy as np
import pytest

import oracle
from isehr_amd.synth import synth_rows, planted_dataset

pytestmark = pytest.mark.gpu
TAU = 1e-6


def _setup(seed, n, d, nq):
    vecs = np.ascontiguousarray(synth_rows(seed, 0, n, d).T)
    qv = np.ascontiguousarray(synth_rows(seed + 1000, 0, nq, d).T)
    return vecs, qv


def _differ_only_at_near_ties(got, ref, s64, tol):
    """Rankings `got`, `ref` [N, Q] over the same rows: wherever a position holds different rows, the two rows' exact
    (float64) scores `s64` [N, Q] are within `tol` -- i.e. only an order the float32 arithmetic of the reference cannot
    resolve itself may differ.  Returns the worst score gap at a differing position."""
    diff = got != ref
    if not diff.any():
        return 0.0
    a = np.take_along_axis(s64, got, 0)[diff]
    b = np.take_along_axis(s64, ref, 0)[diff]
    gap = float(np.abs(a - b).max())
    assert gap <= tol, (gap, tol)
    return gap


def test_ip_ranker_golden(golden_dir):
    """a2: top rows of argsort(-(vecs.T @ qvecs)) -- raw inner product, no normalisation."""
    from isehr_amd.nnsearch import ip_topk_hip
    z = np.load(os.path.join(golden_dir, "ip_rank.npz"))
    seed, n, d, nq = (int(v) for v in z["meta"])
    vecs, qv = _setup(seed, n, d, nq)
    ranks, scores = ip_topk_hip(vecs, qv, 200)
    s64 = (vecs.astype(np.float64).T @ qv.astype(np.float64)).T          # [Q,N]
    scale = float(np.abs(s64).max())
    assert oracle.check_topk_parity(ranks.T, s64, 200, TAU * scale) == []
    assert (ranks == z["ranks_top"]).mean() > 0.99
    assert np.allclose(scores, z["scores_top"], rtol=0, atol=2e-5 * scale)


def test_knn_wrapper_contract():
    """a6: KNN(database,'cosine').search -> (sims f32 desc, ids i64), exact inner-product top-k."""
    from isehr_amd.knn import KNN
    g = synth_rows(5, 0, 4000, 128)
    g /= np.linalg.norm(g, axis=1, keepdims=True)
    q = g[:7] + 0.05 * synth_rows(6, 0, 7, 128)
    knn = KNN(g, "cosine")
    sims, ids = knn.search(q, 50)
    knn.close()
    rs, ri = oracle.knn_flat_ip(g, q, 50)
    assert sims.dtype == np.float32 and ids.dtype == np.int64
    s64 = q.astype(np.float64) @ g.astype(np.float64).T
    assert oracle.check_topk_parity(ids, s64, 50, TAU) == []
    assert (ids[:, 0] == np.arange(7)).all()
    assert np.allclose(sims, rs, atol=2e-6)
    assert (ids == ri).mean() > 0.99


@pytest.mark.parametrize("k_qe", [3, 10])
def test_alpha_qe_vs_reference_golden(golden_dir, k_qe):
    """a3: expanded queries and re-ranked lists against the reference's own feature_enhancement."""
    from isehr_amd.reranking import feature_enhancement_hip, qge1_hip
    z = np.load(os.path.join(golden_dir, "qge.npz"))
    seed, n, d, nq = (int(v) for v in z["meta"])
    vecs, qv = _setup(seed, n, d, nq)
    vecs = vecs / np.linalg.norm(vecs, axis=0, keepdims=True)
    base = z["base"]
    qx, ranks = feature_enhancement_hip(k_qe, base, vecs, 4.0, 200)
    ref_qx, ref_ranks = z[f"qx{k_qe}"], z[f"ranks{k_qe}_top"]
    assert qx.dtype == np.float64 and qx.shape == ref_qx.shape
    assert np.abs(qx - ref_qx).max() < 1e-7            # f32-stored rows vs the reference's f32 inputs
    s64 = (vecs.astype(np.float64).T @ ref_qx).T
    assert oracle.check_topk_parity(ranks.T, s64, 200, TAU) == []
    assert (ranks == ref_ranks).mean() > 0.99
    if k_qe == 3:
        assert np.array_equal(qge1_hip(base, qv, vecs, 200), ranks)


def test_alpha_qe_strided_ranks_and_bad_ids():
    from isehr_amd._lib import Gallery, NORM_NONE
    vecs, qv = _setup(31, 1500, 64, 5)
    g = Gallery.from_host(vecs.T, norm_mode=NORM_NONE)
    idx0, _, _ = g.search(qv.T, 20)
    ranks = idx0.T                                       # [20,Q] view with strides (8, 160)
    i1, s1, q1, _ = g.aqe_search(ranks, 3, 4.0, 20, return_qexp=True)
    i2, s2, q2, _ = g.aqe_search(np.ascontiguousarray(ranks), 3, 4.0, 20, return_qexp=True)
    assert np.array_equal(i1, i2) and np.array_equal(q1, q2)
    ref_q, _ = oracle.feature_enhancement(3, ranks, vecs, 4.0)
    assert np.abs(q1.T - ref_q).max() < 1e-7
    bad = ranks.copy()
    bad[0, 0] = 1500
    with pytest.raises(RuntimeError):
        g.aqe_search(bad, 3, 4.0, 20)
    g.close()


def test_eps_normalisation_matches_l2n(golden_dir):
    """a7: x / (||x|| + 1e-6) (l2n) as an ingest mode; rows read back from the device."""
    from isehr_amd._lib import Gallery, NORM_L2_EPS, NORM_L2
    x = synth_rows(31, 0, 5, 2048)
    z = np.load(os.path.join(golden_dir, "normalise.npz"))
    g = Gallery.from_host(x, norm_mode=NORM_L2_EPS)
    rows = g.get_rows(0, 5)
    g.close()
    assert np.abs(rows - z["l2n"]).max() < 1e-7
    g = Gallery.from_host(x, norm_mode=NORM_L2)
    rows = g.get_rows(0, 5)
    g.close()
    assert np.abs(np.linalg.norm(rows.astype(np.float64), axis=1) - 1).max() < 1e-7


def test_save_load_roundtrip(tmp_path):
    from isehr_amd._lib import Gallery
    g = synth_rows(9, 0, 3000, 96)
    q = synth_rows(10, 0, 6, 96)
    G = Gallery.from_host(g, row_offset=100)
    i1, s1, _ = G.search(q, 30)
    path = str(tmp_path / "gal.bin")
    G.save(path)
    G.close()
    H = Gallery.load(path)
    assert (H.n, H.d, H.row_offset) == (3000, 96, 100)
    i2, s2, _ = H.search(q, 30)
    H.close()
    assert np.array_equal(i1, i2) and np.array_equal(s1, s2)
    assert i1.min() >= 100
    size = os.path.getsize(path)
    blob = open(path, "rb").read()

    def corrupt(offset, label):
        b = bytearray(blob)
        b[offset] ^= 0x10
        open(path, "wb").write(bytes(b))
        with pytest.raises(RuntimeError, match=label):
            Gallery.load(path)

    # one flipped bit anywhere must be caught: header, f32 rows, 16-bit image, row norms (sections in file order)
    rows_bytes, img_bytes = 3000 * 128 * 4, 3072 * 128 * 2
    hdr = size - rows_bytes - img_bytes - 3072 * 12
    assert 64 <= hdr <= 256
    corrupt(20, "header")
    corrupt(hdr + 12345, "f32 rows")
    corrupt(hdr + rows_bytes + 54321, "16-bit image")
    corrupt(size - 7, "row norms")
    open(path, "wb").write(blob + b"x")
    with pytest.raises(RuntimeError, match="trailing"):
        Gallery.load(path)
    open(path, "wb").write(blob)
    Gallery.load(path).close()                                  # the intact file still loads
    with open(path, "r+b") as f:
        f.truncate(size - 1000)
    with pytest.raises(RuntimeError, match="truncated"):
        Gallery.load(path)


def test_xcd_shares_travel_with_the_file_and_within_the_process(tmp_path):
    """VERDICT r04 #7b: the XCD shares the tile kernel learns (common.h XccBalance) are written with the prepared-gallery file and
    remembered per device inside the process: `load -> first search` and a second gallery start from them instead of from an
    even split that the first ~4 large launches would have to correct (mi_gallery_calibrate stays available, no longer needed)."""
    from isehr_amd._lib import Gallery
    n, d = 140000, 64                                      # 547 gallery tiles: launches large enough to measure shares
    rows = synth_rows(501, 0, n, d)
    q = synth_rows(502, 0, 512, d)
    G = Gallery.from_host(rows)
    path = str(tmp_path / "g.mi355gal")
    try:
        w0, l0 = G.xcc_shares()
        assert l0 == -1                                     # no workspace yet
        G.calibrate(6)
        w1, l1 = G.xcc_shares()
        assert l1 >= 5 and abs(float(w1.sum()) - 1.0) < 1e-5 and (w1 > 0.05).all()
        ref, _, _ = G.search(q, 10)
        G.save(path)
    finally:
        G.close()
    L = Gallery.load(path)
    try:
        w2, l2 = L.xcc_shares()
        assert l2 == -1 and not np.allclose(w2, 0.125) and abs(float(w2.sum()) - 1.0) < 1e-5
        got, _, _ = L.search(q[:4], 10)                     # a small batch: creates the workspace, measures nothing
        w3, l3 = L.xcc_shares()
        assert l3 == 0 and np.allclose(w3, w2, atol=1e-6) and not np.allclose(w3, 0.125)
        assert np.array_equal(L.search(q, 10)[0], ref)      # and the answers do not depend on any of it
    finally:
        L.close()
    # a gallery created afterwards in this process starts from the device's last shares too
    H = Gallery.from_host(rows[:4000])
    try:
        H.search(q[:4], 10)
        w4, l4 = H.xcc_shares()
        assert l4 == 0 and not np.allclose(w4, 0.125) and abs(float(w4.sum()) - 1.0) < 1e-5
    finally:
        H.close()
    # a file with anything else behind its last section is refused
    with open(path, "ab") as f:
        f.write(b"garbage")
    with pytest.raises(RuntimeError, match="trailing"):
        Gallery.load(path)


def test_matching_hip_dataset_cache(tmp_path, monkeypatch):
    """The stateful form follows the ANN methods' convention: outputs/<dataset>/ + ifgenerate."""
    from isehr_amd import nnsearch
    monkeypatch.chdir(tmp_path)
    g = synth_rows(2, 0, 2000, 64)
    q = synth_rows(3, 0, 4, 64)
    i1, _ = nnsearch.matching_HIP(10, g, q, dataset="unit/test", ifgenerate=True)
    assert nnsearch.last_timing["source"] == "built"
    nnsearch.wait_for_saves()                               # the file is written behind the call (round 5)
    assert os.path.exists(os.path.join("outputs", "unit_test", "mi355_gallery_l2.bin"))
    assert not [f for f in os.listdir(os.path.join("outputs", "unit_test")) if ".tmp." in f]
    i2, _ = nnsearch.matching_HIP(10, g, q, dataset="unit/test", ifgenerate=False)      # in-process cache
    nnsearch.drop_cached_galleries()
    i3, _ = nnsearch.matching_HIP(10, g, q, dataset="unit/test", ifgenerate=False)      # from disk
    assert nnsearch.last_timing["source"] == "file"
    nnsearch.drop_cached_galleries()
    assert np.array_equal(i1, i2) and np.array_equal(i1, i3)


@pytest.mark.parametrize("nshards", [2, 3, 8])
def test_sharded_two_phase_protocol_equals_single_shard(nshards):
    """e: shards on one GPU, the exact exchange the RCCL path performs (stack == all-gather)."""
    import torch
    from isehr_amd import _lib
    from isehr_amd.sharded import shard_bounds
    n, d, nq, k = 30000, 256, 70, 100
    g = synth_rows(77, 0, n, d)
    qh = synth_rows(78, 0, nq, d)
    single = _lib.Gallery.from_host(g)
    ref_idx, ref_sc, _ = single.search(qh, k)
    single.close()
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    q = torch.from_numpy(qh).to(dev)
    shards = []
    for r in range(nshards):
        lo, hi = shard_bounds(n, nshards, r)
        shards.append(_lib.Gallery.from_host(g[lo:hi], row_offset=lo))
    approx = torch.empty((nshards, nq, k), dtype=torch.float32, device=dev)
    for r, sh in enumerate(shards):
        sh.phase1_device(q.data_ptr(), nq, k, approx[r].data_ptr(), stream)
    L = torch.empty((nq,), dtype=torch.float32, device=dev)
    _lib.kth_of_gathered_device(approx.data_ptr(), nshards, nq, k, L.data_ptr(), stream)
    idx = torch.empty((nshards, nq, k), dtype=torch.int64, device=dev)
    sc = torch.empty((nshards, nq, k), dtype=torch.float32, device=dev)
    sc64 = torch.empty((nshards, nq, k), dtype=torch.float64, device=dev)
    for r, sh in enumerate(shards):
        sh.phase2_device(nq, k, L.data_ptr(), idx[r].data_ptr(), sc[r].data_ptr(), sc64[r].data_ptr(), stream)
    oi = torch.empty((nq, k), dtype=torch.int64, device=dev)
    os_ = torch.empty((nq, k), dtype=torch.float32, device=dev)
    _lib.topk_merge_device(sc64.data_ptr(), idx.data_ptr(), nshards, nq, k, oi.data_ptr(), os_.data_ptr(), stream)
    torch.cuda.synchronize()
    for sh in shards:
        assert sh.status()["overflow_batches"] == 0
        sh.close()
    assert np.array_equal(oi.cpu().numpy(), ref_idx)
    assert np.array_equal(os_.cpu().numpy(), ref_sc)
    # the packed form the RCCL path gathers with ONE collective: [G][2][nq][k] (scores as bit patterns, then indices)
    packed = torch.stack([sc64.view(torch.int64), idx], dim=1).contiguous()
    oi2 = torch.empty_like(oi)
    os2 = torch.empty_like(os_)
    _lib.topk_merge_strided_device(packed[0, 0].data_ptr(), packed[0, 1].data_ptr(), 2 * nq * k, nshards, nq, k,
                                   oi2.data_ptr(), os2.data_ptr(), stream)
    torch.cuda.synchronize()
    assert torch.equal(oi2, oi) and torch.equal(os2, os_)
    # the merge kernel against the oracle's merge
    ms, mi = oracle.merge_topk([s for s in sc64.cpu().numpy()], [i for i in idx.cpu().numpy()], k)
    assert np.array_equal(mi, ref_idx)


def test_sharded_search_verify_falls_back_on_sticky_flags():
    """The asynchronous device path reports a failed speculative threshold only through sticky flags;
    ShardedGallery.search(verify=True) must notice and answer the batch again with the rigorous schedule.  90
    near-duplicates of query 0 planted inside the bootstrap sample defeat both speculative thresholds."""
    import torch
    from isehr_amd import _lib
    from isehr_amd.sharded import ShardedGallery
    n, d, nq, k = 200000, 64, 6, 100
    g = synth_rows(71, 0, n, d)
    qh = synth_rows(72, 0, nq, d)
    rows = np.random.default_rng(3).choice(_lib.sample_source_rows(n), size=90, replace=False)
    for j, r in enumerate(rows):
        g[r] = qh[0] * (1.0 + 0.01 * j) + 0.02 * synth_rows(73, j, 1, d)[0]
    G = _lib.Gallery.from_host(g)
    G.set_option("chunk0_tiles", 32)               # the 8192-row sample the planted rows were drawn from
    sg = ShardedGallery(G)
    q = torch.from_numpy(qh).to("cuda:0")
    s = oracle.exact_scores_f64(g, qh)
    idx, _ = sg.search(q, k)                       # unverified: the flag is raised, the answer may be incomplete
    torch.cuda.synchronize()
    assert sg.any_flag()
    idx, sc = sg.search(q, k, verify=True)
    torch.cuda.synchronize()
    assert not sg.any_flag()
    assert oracle.check_topk_parity(idx.cpu().numpy(), s, k, 1e-6) == []
    assert set(rows) <= set(idx.cpu().numpy()[0])
    G.close()


def test_planted_dataset_map_end_to_end():
    """Search -> alpha-QE -> mAP on a planted dataset equals the CPU restatement's mAP."""
    from isehr_amd.nnsearch import matching_HIP
    from isehr_amd.reranking import QGE_hip
    from isehr_amd import evaluate
    vecs, qv, gnd = planted_dataset(41, 1200, 64, 12)
    idx, _ = matching_HIP(300, vecs.T, qv.T)
    ranks = idx.T
    ref = oracle.matching_l2(300, vecs.T, qv.T).T
    assert np.allclose(evaluate.compute_map_revisited(ranks, gnd), oracle.compute_map_revisited(ref, gnd), atol=1e-6)


def test_whitenapply_golden(golden_dir):
    """a8: P[:dims] @ (X - m) with the eps-normalisation, float64, against the reference's own output."""
    from isehr_amd.whiten import whitenapply_hip
    z = np.load(os.path.join(golden_dir, "normalise.npz"))
    X = synth_rows(32, 0, 40, 24, np.float64).T.copy()
    m = X.mean(axis=1, keepdims=True)
    P = synth_rows(33, 0, 24, 24, np.float64)
    assert np.abs(whitenapply_hip(X, m, P) - z["whiten"]).max() < 1e-12
    assert np.abs(whitenapply_hip(X, m, P, 16) - z["whiten16"]).max() < 1e-12
    # a bigger float32 case against the oracle
    Xb = synth_rows(34, 0, 700, 300).T.copy()
    mb = Xb.mean(axis=1, keepdims=True).astype(np.float64)
    Pb = synth_rows(35, 0, 300, 300, np.float64) / 17.0
    got = whitenapply_hip(Xb, mb, Pb, 200)
    ref = oracle.whitenapply(Xb.astype(np.float64), mb, Pb, 200)
    assert got.shape == (200, 700) and np.abs(got - ref).max() < 1e-12


def test_full_length_ranking_vs_reference_golden(golden_dir):
    """f-3: the complete argsort(-(vecs.T @ qvecs), axis=0) -- every one of the N positions."""
    from isehr_amd.nnsearch import ip_rank_hip, matching_HIP
    z = np.load(os.path.join(golden_dir, "ip_rank.npz"))
    seed, n, d, nq = (int(v) for v in z["meta"])
    vecs, qv = _setup(seed, n, d, nq)
    ranks, scores = ip_rank_hip(vecs, qv, return_scores=True)
    assert ranks.shape == (n, nq) and ranks.dtype == np.int64
    assert all(np.array_equal(np.sort(ranks[:, j]), np.arange(n)) for j in range(nq))      # a permutation
    s64 = vecs.astype(np.float64).T @ qv.astype(np.float64)                                 # [N,Q]
    got = np.take_along_axis(s64, ranks, 0)
    tol = 2e-6 * float(np.abs(s64).max())
    assert (np.diff(got, axis=0) <= tol).all()                                              # sorted up to f32 near-ties
    assert (ranks[:200] == z["ranks_top"]).mean() > 0.99
    ref_ranks, _ = oracle.ip_rank(vecs, qv)
    assert (ranks == ref_ranks).mean() > 0.98
    _differ_only_at_near_ties(ranks, ref_ranks, s64, tol)           # every differing position is a float32 near-tie
    # matching_HIP with K = N (the --mode mAP case) goes through the same path
    idx, _ = matching_HIP(n, vecs.T, qv.T)
    sc = oracle.exact_scores_f64(vecs.T, qv.T)
    assert oracle.check_topk_parity(idx, sc, n, TAU) == []


def test_full_length_ranking_ties_and_qge1_full():
    from isehr_amd._lib import Gallery, NORM_NONE
    from isehr_amd.reranking import qge1_hip
    g = synth_rows(91, 0, 3000, 48)
    g[100:140] = g[7]                      # 41 exact ties
    q = g[[7, 2500]].copy()
    G = Gallery.from_host(g, norm_mode=NORM_NONE)
    idx, sc, _ = G.rank_all(q, return_scores=True)
    G.close()
    tie_block = np.concatenate([[7], np.arange(100, 140)])
    pos = np.flatnonzero(np.isin(idx[0], tie_block))
    assert np.array_equal(idx[0, pos], tie_block) and pos.max() - pos.min() == 40          # contiguous, index order
    assert (np.diff(sc, axis=1) <= 0).all()
    vecs = np.ascontiguousarray((g / np.linalg.norm(g, axis=1, keepdims=True)).T)
    qv = vecs[:, [5, 9]]
    base = np.argsort(-(vecs.T @ qv), axis=0)[:10]
    full = qge1_hip(base, qv, vecs, 10, full=True)
    ref = oracle.qge1(base, qv, vecs, 10)
    assert full.shape == ref.shape == (3000, 2)
    assert (full == ref).mean() > 0.97
    qx = oracle.feature_enhancement(3, base, vecs, 4.0)[0].astype(np.float64)               # expanded queries [D, Q]
    s64 = vecs.astype(np.float64).T @ qx
    _differ_only_at_near_ties(full, ref, s64, 2e-6 * float(np.abs(s64).max()))
    assert np.array_equal(full[:10], qge1_hip(base, qv, vecs, 10))


def test_alpha_qe_across_shards_equals_single_shard():
    """e: per-shard partial sums of the gathered rows (f64) add up to the single-shard expanded query, and the
    re-search of the expanded queries over the shards equals the single-shard alpha-QE result."""
    import torch
    from isehr_amd import _lib
    from isehr_amd.sharded import shard_bounds, ShardedGallery
    n, d, nq, k = 20000, 128, 17, 50
    g = synth_rows(55, 0, n, d)
    g /= np.linalg.norm(g, axis=1, keepdims=True)
    qh = synth_rows(56, 0, nq, d)
    single = _lib.Gallery.from_host(g, norm_mode=_lib.NORM_NONE)
    base, _, _ = single.search(qh / np.linalg.norm(qh, axis=1, keepdims=True), 10)
    ranks_h = np.ascontiguousarray(base.T)                                   # [10, Q]
    ref_idx, ref_sc, ref_qx, _ = single.aqe_search(ranks_h, 3, 4.0, k, return_qexp=True)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    ranks = torch.from_numpy(ranks_h).to(dev)
    nsh = 3
    shards = [_lib.Gallery.from_host(g[lo:hi], norm_mode=_lib.NORM_NONE, row_offset=lo)
              for lo, hi in (shard_bounds(n, nsh, r) for r in range(nsh))]
    total = torch.zeros((nq, d), dtype=torch.float64, device=dev)
    for sh in shards:
        part = torch.empty((nq, d), dtype=torch.float64, device=dev)
        sh.aqe_partial_device(ranks.data_ptr(), ranks.stride(0), ranks.stride(1), nq, 3, 4.0, part.data_ptr(), stream)
        total += part                                                      # == all-reduce(SUM)
    qx = torch.empty((nq, d), dtype=torch.float32, device=dev)
    qx64 = torch.empty((nq, d), dtype=torch.float64, device=dev)
    _lib.aqe_finish_device(total.data_ptr(), nq, d, 1e-6, qx.data_ptr(), qx64.data_ptr(), stream)
    torch.cuda.synchronize()
    assert np.abs(qx64.cpu().numpy() - ref_qx).max() < 1e-15
    # round 4, what the multi-rank aqe_search does: the shards' ROW blocks (zeros for rows of other shards) summed -- one
    # non-zero contributor per element, exact in any order -- then added in j order by the single-shard kernel's own step:
    # the single-shard expanded query bit for bit, wherever the shard boundaries fall
    for k_qe in (3, 10):
        ref_i, ref_s, ref_q, _ = single.aqe_search(ranks_h, k_qe, 4.0, k, return_qexp=True)
        rows_sum = torch.zeros((k_qe, nq, d), dtype=torch.float32, device=dev)
        for sh in reversed(shards):                                        # any order
            rows = torch.empty((k_qe, nq, d), dtype=torch.float32, device=dev)
            sh.aqe_rows_device(ranks.data_ptr(), ranks.stride(0), ranks.stride(1), nq, k_qe, rows.data_ptr(), stream)
            rows_sum += rows
        tot = torch.empty((nq, d), dtype=torch.float64, device=dev)
        _lib.aqe_combine_device(rows_sum.data_ptr(), nq, d, k_qe, 4.0, tot.data_ptr(), stream)
        _lib.aqe_finish_device(tot.data_ptr(), nq, d, 1e-6, qx.data_ptr(), qx64.data_ptr(), stream)
        torch.cuda.synchronize()
        assert np.array_equal(qx64.cpu().numpy(), ref_q)
    # world-size-1 ShardedGallery.aqe_search on the single shard: same code path as the multi-rank one minus the collective
    sg = ShardedGallery(single)
    idx, sc, _ = sg.aqe_search(ranks, 3, 4.0, k)
    torch.cuda.synchronize()
    assert np.array_equal(idx.cpu().numpy(), ref_idx) and np.array_equal(sc.cpu().numpy(), ref_sc)
    for sh in shards:
        sh.close()
    # the row shards of ONE process (MultiDeviceGallery.aqe_search): same answers, same expanded queries
    from isehr_amd.sharded import MultiDeviceGallery
    mg = MultiDeviceGallery.from_host(g, [0, 0, 0], norm_mode=_lib.NORM_NONE)
    try:
        mi, ms, mq = mg.aqe_search(ranks_h, 3, 4.0, k)
        assert np.array_equal(mi, ref_idx) and np.array_equal(ms, ref_sc)
        assert np.array_equal(mq, ref_qx.astype(np.float32))
    finally:
        mg.close()
    single.close()


def test_search_result_survives_the_alpha_qe_that_follows_it():
    """src/online.py:132-152 searches, keeps `ranks`, calls qge1 on them and uses both.  Through ShardedGallery the first
    result used to be the very buffer the re-search of aqe_search wrote into (VERDICT r05 Weak #6: it bit bench.py's own
    checker).  Results now come from a ring of two per (shape, entry point): intact through any aqe_search and through the
    next search of the same shape."""
    import torch
    from isehr_amd import _lib
    from isehr_amd.sharded import ShardedGallery
    n, d, nq, k = 20000, 128, 33, 50
    g = synth_rows(61, 0, n, d)
    g /= np.linalg.norm(g, axis=1, keepdims=True)
    q = torch.from_numpy(synth_rows(62, 0, 2 * nq, d)).cuda()
    gal = _lib.Gallery.from_host(g, norm_mode=_lib.NORM_NONE)
    try:
        sg = ShardedGallery(gal)
        idx1, sc1 = sg.search(q[:nq], k)
        keep_i, keep_s = idx1.clone(), sc1.clone()
        idx2, sc2, _ = sg.aqe_search(idx1.t(), 3, 4.0, k)                 # same (Q, k): used to overwrite idx1 / sc1
        torch.cuda.synchronize()
        assert torch.equal(idx1, keep_i) and torch.equal(sc1, keep_s)
        assert idx2.data_ptr() != idx1.data_ptr() and not torch.equal(idx2, idx1)
        keep2 = idx2.clone()
        idx3, _ = sg.search(q[nq:], k)                                    # the NEXT search of the shape: both earlier results stay
        torch.cuda.synchronize()
        assert torch.equal(idx1, keep_i) and torch.equal(idx2, keep2) and idx3.data_ptr() != idx1.data_ptr()
        sg.search(q[:nq], k)                                              # the second-next one takes idx1's buffer again
        torch.cuda.synchronize()
        assert torch.equal(idx1, keep_i)                                  # (same queries: same content, same buffer)
        assert sg.search(q[nq:], k)[0].data_ptr() == idx3.data_ptr()
    finally:
        gal.close()


def test_whitening_on_the_device_and_into_a_gallery(golden_dir):
    """mi_whiten_apply_device (f64 MFMA GEMM, centring on load, one-pass normalisation) against the reference's own
    whitenapply outputs (tests/golden/normalise.npz, 1e-12) for row-major and [D, N] sources of both dtypes, and
    mi_gallery_append_whitened_device: the whitened rows ingested straight into a searchable MI_NORM_L2_EPS gallery equal
    float32(whitenapply) and are found by the search."""
    import torch
    from isehr_amd import _lib
    z = np.load(os.path.join(golden_dir, "normalise.npz"))
    X = synth_rows(32, 0, 40, 24, np.float64).T.copy()   # [D, N] like the reference (the inputs of test_whitenapply_golden)
    m = X.mean(axis=1, keepdims=True)
    P = synth_rows(33, 0, 24, 24, np.float64)
    D, N = X.shape
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    md = torch.from_numpy(np.ascontiguousarray(m.reshape(-1))).to(dev)
    Pd = torch.from_numpy(np.ascontiguousarray(P)).to(dev)
    for dims, ref in ((D, z["whiten"]), (16, z["whiten16"])):
        for dt in (np.float64, np.float32):
            Xc = X.astype(dt)
            want = ref if dt == np.float64 else oracle.whitenapply(Xc.astype(np.float64), m, P, dims)
            for layout in ("dn", "rows"):
                src = torch.from_numpy(np.ascontiguousarray(Xc if layout == "dn" else Xc.T)).to(dev)
                out = torch.empty((N, dims), dtype=torch.float64, device=dev)
                rs, cs = (1, N) if layout == "dn" else (D, 1)
                _lib.whiten_apply_device(src.data_ptr(), N, D, md.data_ptr(), Pd.data_ptr(), dims, out.data_ptr(),
                                         dtype=_lib.MI_F64 if dt == np.float64 else _lib.MI_F32, row_stride=rs, col_stride=cs,
                                         stream=st)
                torch.cuda.synchronize()
                assert np.abs(out.cpu().numpy().T - want).max() < 1e-12, (dims, dt, layout)
    # a bigger, ragged case straight into a gallery: 5000 x 200 descriptors -> 72 dims
    rng = np.random.default_rng(3)
    n, d, dims = 5000, 200, 72
    Xb = rng.standard_normal((n, d)).astype(np.float32)
    mb = Xb.mean(axis=0).astype(np.float64)
    Pb = rng.standard_normal((d, d)) / np.sqrt(d)
    want = oracle.whitenapply(Xb.T.astype(np.float64), mb.reshape(-1, 1), Pb, dims).T          # [n, dims]
    xd = torch.from_numpy(Xb).to(dev)
    gal = _lib.Gallery.empty(n, dims, norm_mode=_lib.NORM_L2_EPS)
    try:
        gal.append_whitened_device(xd.data_ptr(), 3000, d, torch.from_numpy(mb).to(dev).data_ptr(),
                                   torch.from_numpy(Pb).to(dev).data_ptr(), stream=st)
        gal.append_whitened_device(xd[3000:].data_ptr(), 2000, d, torch.from_numpy(mb).to(dev).data_ptr(),
                                   torch.from_numpy(Pb).to(dev).data_ptr(), stream=st)
        assert gal.n == n
        got = gal.get_rows(0, n)
        assert np.abs(got - want).max() < 1e-7
        idx, sc, _ = gal.search(want[:7].astype(np.float32), 5)
        assert np.array_equal(idx[:, 0], np.arange(7)) and np.abs(sc[:, 0] - 1.0).max() < 1e-5
    finally:
        gal.close()


def test_aqe_and_dba_vs_reference_golden(golden_dir):
    """f-4: average_query_expansion / database_augmentation against ranks captured from the reference functions."""
    from isehr_amd.reranking import average_query_expansion_hip, database_augmentation_hip
    z = np.load(os.path.join(golden_dir, "aqe_dba.npz"))
    va = synth_rows(95, 0, 700, 40).astype(np.float64)
    ca = synth_rows(96, 0, 9, 40).astype(np.float64)
    va = 0.7 * va + 1.2 * ca[np.arange(700) % 9] + 0.3
    va /= np.linalg.norm(va, axis=1, keepdims=True)
    qa = va[:11] + 0.1 * synth_rows(97, 0, 11, 40)
    qa /= np.linalg.norm(qa, axis=1, keepdims=True)
    qv, vecs = qa.T.copy(), va.T.copy()
    for fn, key in ((average_query_expansion_hip, "ranks_aqe"), (database_augmentation_hip, "ranks_dba")):
        got = fn(qv, vecs, 50)
        assert got.shape == (50, 11) and (got == z[key]).mean() > 0.97, (key, (got == z[key]).mean())
        assert (got[0] == z[key][0]).all()


def test_rank_positions_and_prefix_equal_the_full_ranking():
    """mi_rank_positions / mi_rank_prefix against mi_rank_all on a gallery with exact ties."""
    from isehr_amd._lib import Gallery, NORM_NONE
    g = synth_rows(93, 0, 5000, 40)
    g[200:230] = g[11]                                          # 31 exact ties
    q = np.concatenate([g[[11, 4000]], synth_rows(94, 0, 3, 40)])
    G = Gallery.from_host(g, norm_mode=NORM_NONE)
    try:
        full, _ = G.rank_all(q)
        inv = np.empty_like(full)
        for i in range(len(q)):
            inv[i, full[i]] = np.arange(5000)
        ids = np.stack([np.concatenate([[11, 215, 229, 200, 4999, 0], np.arange(300, 340), [-1, -1]]) for _ in range(len(q))])
        pos = G.rank_positions(q, ids)
        assert (pos[:, -2:] == -1).all()
        assert np.array_equal(pos[:, :-2], np.take_along_axis(inv, ids[:, :-2], 1))
        pre, sc, _ = G.rank_prefix(q, 3000, return_scores=True)
        assert np.array_equal(pre, full[:, :3000]) and (np.diff(sc, axis=1) <= 0).all()
    finally:
        G.close()


def test_qge_large_branch_map_is_the_full_ranking_map():
    """QGE for N >= 120000 (src/utils/Reranking.py:273-284): the printed mAP is the mAP of the COMPLETE ranking
    (:206-207, :280-283).  Positives are planted far beyond rank 2048 (weak matches among 130k rows), so a top-K
    evaluation would report less."""
    from isehr_amd.reranking import QGE_hip
    from isehr_amd.nnsearch import ip_topk_hip
    from isehr_amd import evaluate
    n, d, nq = 130000, 48, 8
    vecs, qv, gnd = planted_dataset(43, n, d, nq, n_pos=(30, 40), sigmas=(0.35, 4.0, 6.0))
    base, _ = ip_topk_hip(vecs, qv, 10)
    out = QGE_hip(base, qv, vecs, "roxford5k", gnd, quiet=True)
    qx_ref, ranks_ref = oracle.feature_enhancement(3, base, vecs, 4.0)
    ref = oracle.compute_map_revisited(ranks_ref, gnd)
    assert np.allclose(out["map_aqe"], ref, rtol=0, atol=1e-6), (out["map_aqe"], ref)
    # the labelled images really do sit deeper than the top-K path reaches, and a K = 1000 cut scores lower
    deepest = max(max(p.values()) for p in out["positions"])
    assert deepest > 2048
    cut = evaluate.compute_map_revisited(ranks_ref[:1000], gnd)
    assert cut[1] < ref[1] - 1e-3
    # positions equal the oracle's (f32 near-ties aside: compare through the scores)
    inv = np.empty_like(ranks_ref)
    for i in range(nq):
        inv[ranks_ref[:, i], i] = np.arange(n)
    for i in range(nq):
        ids = np.array(sorted(out["positions"][i]))
        got = np.array([out["positions"][i][int(v)] for v in ids])
        assert np.abs(got - inv[ids, i]).max() <= 2


def test_distractor_tensor_and_block_ingest(tmp_path):
    """f-1: the reference keeps the 1M distractors as a torch tensor [D, N] float32 (src/extract_1m.py:97-98) and
    concatenates [rOxford | distractors] on the host (src/test_rOP1m.py:136-139).  Here the file is memory-mapped and the
    gallery is built block by block from column chunks (2-D copies, strided ingest): the result must equal the gallery
    built from the concatenated array, bit for bit, for normalised and raw rows."""
    import torch
    from isehr_amd._lib import Gallery, NORM_L2, NORM_NONE
    from isehr_amd.entry.features import load_torch_vecs
    from isehr_amd.nnsearch import matching_HIP, ColumnBlocks
    d, n1, n2 = 96, 700, 50021
    small = np.ascontiguousarray(synth_rows(15, 0, n1, d).T)                  # [D, n1] like the feature pickles
    big = torch.from_numpy(np.ascontiguousarray(synth_rows(16, 0, n2, d).T))   # [D, n2]
    path = str(tmp_path / "net_vecs_revisitop1m.pt")
    torch.save(big, path)
    v1m = load_torch_vecs(path)
    assert v1m.shape == (d, n2) and v1m.dtype == np.float32 and np.array_equal(v1m, big.numpy())
    q = synth_rows(17, 0, 9, d)
    cat = np.concatenate([small, big.numpy()], axis=1)
    for norm in (NORM_L2, NORM_NONE):
        ref = Gallery.from_host(cat.T, norm_mode=norm)
        got = Gallery.from_blocks([small, v1m], norm_mode=norm, chunk_rows=8192)
        try:
            assert got.n == n1 + n2
            assert np.array_equal(got.get_rows(0, got.n), ref.get_rows(0, ref.n))
            assert got.get_option("image_dtype") == ref.get_option("image_dtype")
            i1, s1, _ = ref.search(q, 40)
            i2, s2, _ = got.search(q, 40)
            assert np.array_equal(i1, i2) and np.array_equal(s1, s2)
        finally:
            ref.close()
            got.close()
    # through the matcher surface: ColumnBlocks stands for the concatenation
    a, _ = matching_HIP(25, cat.T, q)
    b, _ = matching_HIP(25, ColumnBlocks([small, v1m]), q)
    assert np.array_equal(a, b)


def test_qge_large_branch_accepts_column_blocks():
    """QGE on [rOxford | 1M distractors] (src/test_rOP1m.py:136-139,168) without the host concatenation: the column blocks
    go to the device one by one; rankings and mAP equal those of the concatenated array."""
    from isehr_amd.nnsearch import matching_HIP, ColumnBlocks
    from isehr_amd.reranking import QGE_hip
    from isehr_amd.synth import planted_dataset
    vecs, qvecs, gnd = planted_dataset(5, 130000, 32, 12)
    ranks = matching_HIP(100, vecs.T, qvecs.T)[0].T
    a = QGE_hip(ranks, qvecs, vecs, "roxford5k", gnd, K=100, quiet=True)
    b = QGE_hip(ranks, qvecs, ColumnBlocks([vecs[:, :4993], vecs[:, 4993:]]), "roxford5k", gnd, K=100, quiet=True)
    assert np.array_equal(a["ranks_aqe"], b["ranks_aqe"])
    assert a["map_aqe"] == b["map_aqe"]
    assert np.array_equal(a["qvecs_qe"], b["qvecs_qe"])


@pytest.mark.parametrize("ndev", [2, 3])
def test_multi_device_gallery_in_one_process_equals_single_device(ndev):
    """sharded.MultiDeviceGallery / matching_HIP(devices=[...]): the row shards of ONE process (the reference's drivers are
    single processes).  The test box has one GPU, so every "device" is GPU 0 -- the code path (per-device handles, phase
    API, device-to-device copies, merge) is the one several GPUs take; answers equal the single-handle search bit for bit,
    through a strided `vecs.T` view and over a batch boundary (1100 queries), and a raised flag triggers the fallbacks."""
    from isehr_amd import _lib, nnsearch
    from isehr_amd.sharded import MultiDeviceGallery
    n, d, nq, k = 41003, 192, 1100, 100
    vecs = np.ascontiguousarray(synth_rows(301, 0, n, d).T)                  # the reference's [D, N] layout
    qv = synth_rows(302, 0, nq, d)
    single = _lib.Gallery.from_host(vecs.T)
    ref_idx, ref_sc, _ = single.search(qv, k)
    single.close()
    idx, t, sc = nnsearch.matching_HIP(k, vecs.T, qv, devices=[0] * ndev, return_scores=True)
    assert np.array_equal(idx, ref_idx) and np.array_equal(sc, ref_sc) and t > 0
    mg = MultiDeviceGallery.from_host(vecs.T, [0] * ndev)
    try:
        assert mg.n == n and len(mg.shards) == ndev
        # a reduced survivor capacity on every shard: whichever path ends up answering (filter, or the fallbacks after a
        # raised overflow flag), the answer is the same
        for s in mg.shards:
            s.set_option("survivor_cap", 4096)
        i2, s2 = mg.search(qv[:70], k)
        assert np.array_equal(i2, ref_idx[:70]) and np.array_equal(s2, ref_sc[:70])
        with pytest.raises(RuntimeError):
            mg.search(qv[:4], n + 1)
    finally:
        mg.close()


def test_multi_device_gallery_with_more_devices_than_k_fits(capsys):
    """3000 rows over 8 listed devices with K = 1000 -- a shard of 375 rows cannot answer phase 1 with 1000
    rows.  matching_HIP(devices=[...]) then uses as many of the listed devices as hold >= K rows each (and says so); a
    gallery that was created for a smaller K refuses the larger one with a message that names the cause."""
    from isehr_amd import _lib, nnsearch
    from isehr_amd.sharded import MultiDeviceGallery
    n, d, k = 3000, 96, 1000
    g = synth_rows(311, 0, n, d)
    qv = synth_rows(312, 0, 5, d)
    single = _lib.Gallery.from_host(g)
    ref_idx, ref_sc, _ = single.search(qv, k)
    single.close()
    idx, _, sc = nnsearch.matching_HIP(k, g, qv, devices=[0] * 8, return_scores=True)
    assert "using 3 of the 8 listed devices" in capsys.readouterr().out
    assert np.array_equal(idx, ref_idx) and np.array_equal(sc, ref_sc)
    mg = MultiDeviceGallery.from_host(g, [0] * 8)                     # made for small K: 8 shards of 375 rows
    try:
        assert len(mg.shards) == 8
        i2, s2 = mg.search(qv, 300)
        assert np.array_equal(i2, ref_idx[:, :300]) and np.array_equal(s2, ref_sc[:, :300])
        with pytest.raises(RuntimeError, match="smallest shard"):
            mg.search(qv, k)
    finally:
        mg.close()


def test_multi_device_gallery_on_two_distinct_devices():
    """The peer copies between REAL devices, the hipSetDevice of every handle call against torch's device context and the
    per-device current streams: only where the box has two GPUs (the one-GPU test box skips; the driver's 8-GPU node runs it)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    from isehr_amd import _lib, nnsearch
    n, d, nq, k = 41003, 192, 1100, 100
    g = synth_rows(301, 0, n, d)
    qv = synth_rows(302, 0, nq, d)
    single = _lib.Gallery.from_host(g, device=1)                      # and a single handle on a device that is not 0
    ref_idx, ref_sc, _ = single.search(qv, k)
    single.close()
    idx, _, sc = nnsearch.matching_HIP(k, g, qv, devices=[0, 1], return_scores=True)
    assert np.array_equal(idx, ref_idx) and np.array_equal(sc, ref_sc)
    idx, _, sc = nnsearch.matching_HIP(k, g, qv, devices=[1, 0], return_scores=True)
    assert np.array_equal(idx, ref_idx) and np.array_equal(sc, ref_sc)
    # alpha-QE over the two devices: rows gathered per device, summed on device 0, re-search on both --
    # the expanded queries and the answers of ONE handle, bit for bit
    from isehr_amd.sharded import MultiDeviceGallery
    gn = g / np.linalg.norm(g, axis=1, keepdims=True)
    single = _lib.Gallery.from_host(gn, norm_mode=_lib.NORM_NONE, device=1)
    base, _, _ = single.search(qv[:300] / np.linalg.norm(qv[:300], axis=1, keepdims=True), 10)
    ranks = np.ascontiguousarray(base.T)
    ref_i, ref_s, ref_q, _ = single.aqe_search(ranks, 3, 4.0, k, return_qexp=True)
    single.close()
    mg = MultiDeviceGallery.from_host(gn, [0, 1], norm_mode=_lib.NORM_NONE)
    try:
        mi, ms, mq = mg.aqe_search(ranks, 3, 4.0, k)
        assert np.array_equal(mi, ref_i) and np.array_equal(ms, ref_s) and np.array_equal(mq, ref_q.astype(np.float32))
    finally:
        mg.close()
