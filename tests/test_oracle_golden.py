"""CPU: the oracle (numpy restatement) against the golden vectors produced by the reference's own
functions (oracle/make_golden.py).  This is what pins the oracle."""
import os

import numpy as np
import pytest

import oracle
from isehr_amd.synth import synth_rows, planted_dataset


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


@pytest.mark.parametrize("seed", [11, 12, 13])
@pytest.mark.parametrize("dt", ["float32", "float64"])
def test_matching_l2_bitexact(golden_dir, seed, dt):
    z = _load(golden_dir, "matching_l2.npz")
    tag = f"s{seed}_{dt}"
    seed_, n, d, nq, k = z[tag + "_meta"]
    g = synth_rows(seed, 0, n, d, np.dtype(dt))
    q = synth_rows(seed + 1000, 0, nq, d, np.dtype(dt))
    idx = oracle.matching_l2(int(k), g, q)
    # same numpy ops in the same order -> identical indices, including near-tie order
    assert np.array_equal(idx, z[tag + "_idx"])


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_reference_is_within_tolerance_of_f64_truth(golden_dir, seed):
    """Documents the reference's own order noise and validates the near-tie tolerant
    comparator on the reference's output."""
    z = _load(golden_dir, "matching_l2.npz")
    for dt, tau in (("float32", 1e-6), ("float64", 1e-12)):
        tag = f"s{seed}_{dt}"
        _, n, d, nq, k = z[tag + "_meta"]
        g = synth_rows(seed, 0, n, d, np.dtype(dt))
        q = synth_rows(seed + 1000, 0, nq, d, np.dtype(dt))
        s = oracle.exact_scores_f64(g, q)
        assert oracle.check_topk_parity(z[tag + "_idx"], s, int(k), tau) == []
        ex_idx, _ = oracle.exact_topk_f64(g, q, int(k))
        if dt == "float64":
            assert np.array_equal(ex_idx, z[tag + "_idx"])


def test_matching_l2_edge_cases(golden_dir):
    z = _load(golden_dir, "matching_l2_edge.npz")
    assert np.array_equal(oracle.matching_l2(8, z["g"], z["q"]), z["idx"])
    with np.errstate(all="ignore"):
        got = oracle.matching_l2(64, z["gz"], z["q"])
    assert np.array_equal(got, z["idxz"])
    # a zero gallery row normalises to NaN and sorts last in the reference
    assert (z["idxz"][:, -1] == 20).all()


def test_ip_rank(golden_dir):
    z = _load(golden_dir, "ip_rank.npz")
    seed, n, d, nq = z["meta"]
    vecs = np.ascontiguousarray(synth_rows(seed, 0, n, d).T)
    qv = np.ascontiguousarray(synth_rows(seed + 1000, 0, nq, d).T)
    ranks, scores = oracle.ip_rank(vecs, qv)
    assert np.array_equal(ranks[:200], z["ranks_top"])
    assert np.array_equal(np.take_along_axis(scores, ranks[:200], 0), z["scores_top"])


def test_feature_enhancement_and_qge1(golden_dir):
    z = _load(golden_dir, "qge.npz")
    seed, n, d, nq = z["meta"]
    vecs = np.ascontiguousarray(synth_rows(seed, 0, n, d).T)
    qv = np.ascontiguousarray(synth_rows(seed + 1000, 0, nq, d).T)
    vecs = vecs / np.linalg.norm(vecs, axis=0, keepdims=True)
    qv = qv / np.linalg.norm(qv, axis=0, keepdims=True)
    base = z["base"]
    qx3, r3 = oracle.feature_enhancement(3, base, vecs, 4.0)
    qx10, r10 = oracle.feature_enhancement(10, base, vecs, 4.0)
    assert np.array_equal(qx3, z["qx3"]) and np.array_equal(qx10, z["qx10"])
    assert np.array_equal(r3[:200], z["ranks3_top"])
    assert np.array_equal(r10[:200], z["ranks10_top"])
    assert np.array_equal(oracle.qge1(base, qv, vecs, 100)[:200], z["ranks3_top"])
    assert qx3.dtype == np.float64          # the f64 weight array promotes


def test_l2n_and_whitenapply(golden_dir):
    z = _load(golden_dir, "normalise.npz")
    x = synth_rows(31, 0, 5, 2048)
    assert np.allclose(oracle.l2n(x), z["l2n"], rtol=0, atol=1e-7)
    X = synth_rows(32, 0, 40, 24, np.float64).T.copy()
    m = X.mean(axis=1, keepdims=True)
    P = synth_rows(33, 0, 24, 24, np.float64)
    assert np.array_equal(oracle.whitenapply(X, m, P), z["whiten"])
    assert np.array_equal(oracle.whitenapply(X, m, P, 16), z["whiten16"])


def test_compute_map(golden_dir):
    z = _load(golden_dir, "map.npz")
    vecs, qv, gnd = planted_dataset(41, 1200, 64, 12)
    rk = np.argsort(-(vecs.T @ qv), axis=0)
    for name, r in (("full", rk), ("top100", rk[:100])):
        got = oracle.compute_map_revisited(r, gnd)
        assert np.allclose(got, z[f"{name}_map_EMH"], rtol=0, atol=1e-12)
    assert z["full_map_EMH"][0] > 0.5        # planted positives are actually retrievable


def test_synth_generator_is_stable():
    v = synth_rows(1234, 5, 2, 8)
    assert v.dtype == np.float32 and v.shape == (2, 8)
    # frozen known-answer values: the device generator must reproduce exactly these
    ref = synth_rows(1234, 0, 7, 8)[5:7]
    assert np.array_equal(v, ref)
    assert abs(float(synth_rows(7, 0, 512, 512).std()) - 1.1547) < 0.01


def _aqe_dba_inputs():
    va = synth_rows(95, 0, 700, 40).astype(np.float64)
    ca = synth_rows(96, 0, 9, 40).astype(np.float64)
    va = 0.7 * va + 1.2 * ca[np.arange(700) % 9] + 0.3
    va /= np.linalg.norm(va, axis=1, keepdims=True)
    qa = va[:11] + 0.1 * synth_rows(97, 0, 11, 40)
    qa /= np.linalg.norm(qa, axis=1, keepdims=True)
    return qa.T.copy(), va.T.copy()


def test_aqe_and_dba(golden_dir):
    z = _load(golden_dir, "aqe_dba.npz")
    qv, vecs = _aqe_dba_inputs()
    assert np.array_equal(oracle.average_query_expansion(qv, vecs, 50), z["ranks_aqe"])
    assert np.array_equal(oracle.database_augmentation(qv, vecs, 50), z["ranks_dba"])


def test_blas_style_flat_ip_equals_the_plain_restatement():
    """knn_flat_ip_blas (sgemm + partial selection: the CPU baseline bench.py times at full BLAS width) returns
    exactly what knn_flat_ip (full stable argsort) returns, ties at the k-th score included."""
    rng = np.random.default_rng(5)
    db = rng.standard_normal((3000, 48)).astype(np.float32)
    db[100] = db[7]
    db[2000] = db[7]                                  # exact ties
    q = np.vstack([db[7:8] * 2.0, rng.standard_normal((9, 48)).astype(np.float32)])
    for k in (1, 2, 3, 50, 3000):
        s0, i0 = oracle.knn_flat_ip(db, q, k)
        s1, i1 = oracle.knn_flat_ip_blas(db, q, k)
        assert np.array_equal(i0, i1) and np.array_equal(s0, s1)


@pytest.mark.parametrize("tag", ["a", "b"])
@pytest.mark.parametrize("dt", ["float32", "float64"])
def test_matching_fractional_dis_bitexact(golden_dir, tag, dt):
    """src/utils/nnsearch.py:709-731, including its slicing of the query axis by K."""
    z = _load(golden_dir, "fractional.npz")
    seed, n, d, nq, k = (int(v) for v in z[f"{tag}_{dt}_meta"])
    g = synth_rows(seed, 0, n, d, np.dtype(dt))
    q = synth_rows(seed + 1000, 0, nq, d, np.dtype(dt))
    idx = oracle.matching_fractional_dis(k, g, q)
    assert idx.shape == (min(nq, k), k)
    assert np.array_equal(idx, z[f"{tag}_{dt}_idx"])


def test_kr_reranking_bitexact(golden_dir):
    """src/utils/Reranking.py:447-624 (kr_reranking), captured by oracle/make_golden.py from the reference function itself
    (Tensor.cuda made the identity for that call: no GPU in the build container).  The numpy restatement reproduces the
    reference's `indices` on every one of the 7 x 400 positions."""
    z = _load(golden_dir, "kr_rerank.npz")["indices"]
    vk = synth_rows(98, 0, 400, 32).astype(np.float64)
    ck = synth_rows(99, 0, 25, 32).astype(np.float64)
    vk = 0.6 * vk + 1.3 * ck[np.arange(400) % 25]
    vk /= np.linalg.norm(vk, axis=1, keepdims=True)
    qk = vk[::57][:7] + 0.15 * synth_rows(100, 0, 7, 32)
    qk /= np.linalg.norm(qk, axis=1, keepdims=True)
    idx = oracle.kr_reranking(qk.T.astype(np.float32), vk.T.astype(np.float32))
    assert idx.shape == z.shape == (7, 400)
    assert np.array_equal(idx, z)
    # the re-ranking is not the plain cosine order (the Jaccard term moves images)
    plain = np.argsort(-(qk @ vk.T), axis=1, kind="stable")
    assert (idx != plain).mean() > 0.05


def test_diffusion_graph_vs_reference_golden(golden_dir):
    """a5, the part of src/utils/diffusion.py that runs under today's scipy: Diffusion.get_affinity / get_laplacian
    (:87-116) called on the reference's own class (oracle/make_golden.py) -- the oracle's restatement must build the same
    sparse matrices entry for entry."""
    z = np.load(os.path.join(golden_dir, "diffusion_graph.npz"))
    sims, ids = z["sims"], z["ids"]
    aff = oracle.get_affinity(sims.copy(), ids).tocsr()
    aff.sort_indices()
    assert np.array_equal(aff.indptr, z["aff_indptr"]) and np.array_equal(aff.indices, z["aff_indices"])
    assert aff.data.dtype == z["aff_data"].dtype and np.array_equal(aff.data, z["aff_data"])
    assert (sims[::7, -3:] < 0).all() and aff.data.min() >= 0          # the negative similarities were clipped, not cubed
    lap = oracle.get_laplacian(sims[:, :15].copy(), ids[:, :15]).tocsr()
    lap.sort_indices()
    assert np.array_equal(lap.indptr, z["lap_indptr"]) and np.array_equal(lap.indices, z["lap_indices"])
    assert lap.data.dtype == z["lap_data"].dtype and np.array_equal(lap.data, z["lap_data"])


def _diffusion_solve_features():
    """The feature set of oracle/make_golden.py's diffusion fixtures (seeds 61 / 62: 300 clustered unit rows, 24-d)."""
    vd = synth_rows(61, 0, 300, 24).astype(np.float64)
    cd = synth_rows(62, 0, 12, 24).astype(np.float64)
    vd = 0.8 * vd + 1.1 * cd[np.arange(300) % 12]
    vd /= np.linalg.norm(vd, axis=1, keepdims=True)
    return vd.astype(np.float32)


def test_diffusion_solve_vs_reference_golden(golden_dir):
    """a5, the per-node solve: the reference's own get_offline_result (src/utils/diffusion.py:15-19) was run by
    oracle/make_golden.py on the Laplacian its own get_laplacian built (its `cg(tol=)` keyword forwarded as `rtol`, atol = 0:
    the same stopping rule since ||e0|| = 1).  The oracle's diffusion_offline must reproduce the truncated CG solutions of
    those 50 nodes bit for bit, on the k-NN lists it finds itself."""
    z = np.load(os.path.join(golden_dir, "diffusion_solve.npz"))
    T, kd, nodes = int(z["n_trunc"]), int(z["kd"]), z["nodes"]
    f = _diffusion_solve_features()
    off, sims, ids, lap, allsc = oracle.diffusion_offline(f, T, kd, return_parts=True)
    assert len(nodes) == 50 and z["scores"].shape == (50, T) and z["scores"].dtype == np.float64
    assert np.array_equal(ids[nodes], z["ids"])
    assert np.array_equal(allsc[nodes], z["scores"])
    # the solutions are not trivial: the node itself carries the largest weight, its neighbours positive mass, and 20
    # iterations at tol 1e-6 leave the iteration truncated for some nodes and converged for others
    assert (allsc[nodes].argmax(axis=1) == 0).all() and (allsc[nodes, 0] > 1.0).all()
    assert (np.abs(allsc[nodes][:, 1:kd]).max(axis=1) > 1e-3).all()
    # and they land in the float32 CSR matrix the way src/utils/diffusion.py:80-84 assembles it
    dense = np.asarray(off[nodes].todense())
    for r, i in enumerate(nodes):
        assert np.array_equal(dense[r, ids[i]], allsc[i].astype(np.float32))
