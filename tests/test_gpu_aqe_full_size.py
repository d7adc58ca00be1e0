"""GPU: BASELINE.json configs[4] at its own size -- rParis6k + 1M distractors, N = 1,007,323 x 2048, alpha-QE (k = 3,
w = 4, N >= 120 000 branch of QGE: src/utils/Reranking.py:195-208, 273-283) then re-search, K = 100 -- against the
oracle itself for 8 queries (every stored row scored in float64 on the host) and through size-independent properties: the
expanded queries against a float64 re-computation from the stored rows, the re-search against the f32-scored path, the
returned scores against float64."""
import numpy as np
import pytest

from _fullsize import assert_oracle_parity, host_f64_scores_and_topk

pytestmark = pytest.mark.gpu
N, D, K, NQ = 1007323, 2048, 100, 70          # 70 queries: rParis6k's query set


def test_aqe_at_rparis_plus_1m():
    import torch
    from isehr_amd import _lib
    dev = torch.device("cuda", 0)
    s = torch.cuda.current_stream().cuda_stream
    raw = torch.empty((N, D), dtype=torch.float32, device=dev)
    _lib.synth_fill_device(raw.data_ptr(), 99, 0, N, D, s)
    q = torch.empty((NQ, D), dtype=torch.float32, device=dev)
    _lib.synth_fill_device(q.data_ptr(), 100, 0, NQ, D, s)
    # plant a cluster per query (the query + noise at three levels) so that the top-3 rows the expansion sums are
    # meaningful neighbours, spread over the whole row range
    gen = torch.Generator(device="cpu").manual_seed(5)
    for qi in range(NQ):
        for j, sigma in enumerate((0.3, 0.5, 0.8, 1.2)):
            row = (qi * 14389 + j * 251003 + 17) % N
            raw[row] = q[qi] + sigma * torch.randn(D, generator=gen).to(dev) * q[qi].norm() / D ** 0.5
    torch.cuda.synchronize()
    g = _lib.Gallery.from_device_ptr(raw.data_ptr(), N, D)
    del raw
    torch.cuda.empty_cache()
    try:
        idx, sc, _ = g.search(q.cpu().numpy(), K)
        ranks = np.ascontiguousarray(idx.T)                          # ranks[K, Q] like `ranks = match_idx.T`
        aidx, asc, qx, _ = g.aqe_search(ranks, 3, 4.0, K, return_qexp=True)
        assert g.flags() == 0
        # expanded queries: weights ((3 - j) / 3)^4, sum of the stored (normalised) rows in float64, / (norm + 1e-6)
        w = (np.arange(3, 0, -1) / 3.0) ** 4.0
        for qi in range(NQ):
            rows = np.stack([g.get_rows(int(ranks[j, qi]), 1)[0] for j in range(3)]).astype(np.float64)
            e = (rows * w[:, None]).sum(0)
            e /= np.linalg.norm(e) + 1e-6
            assert np.abs(e - qx[qi]).max() < 1e-12
        # the planted neighbours lead the first search in the order of their noise level, and stay on top after expansion
        for qi in range(NQ):
            want = [(qi * 14389 + j * 251003 + 17) % N for j in range(4)]
            assert list(idx[qi, :4]) == want and set(aidx[qi, :4]) == set(want)
        assert (np.diff(asc, axis=1) <= 0).all() and all(len(set(r)) == K for r in aidx)
        # returned scores = exact inner products of the expanded queries (f32 rounding of q' as the search sees it)
        qx32 = qx.astype(np.float32).astype(np.float64)
        for qi in range(0, NQ, 7):
            rows = np.stack([g.get_rows(int(r), 1)[0] for r in aidx[qi]]).astype(np.float64)
            assert np.abs(rows @ qx32[qi] - asc[qi]).max() < 3e-7
        # the oracle itself at this size: the expanded queries of 8 queries (used as they are: no second normalisation, like
        # `scores = np.dot(vecs.T, qvecs_qe)`, src/utils/Reranking.py:206) against all 1 007 323 stored rows in float64 on the
        # host; the re-search's answer must be that ranking's top-100
        pick = np.arange(0, NQ, 9)
        scores, top_i, top_s = host_f64_scores_and_topk(g, qx32[pick], K, normalize_queries=False)
        assert_oracle_parity(aidx[pick], asc[pick], scores, top_i, top_s, K)
        # re-search vs the f32-scored path
        g.set_option("force_exact", 1)
        try:
            eidx, esc, _, _ = g.aqe_search(ranks, 3, 4.0, K)
        finally:
            g.set_option("force_exact", 0)
        assert np.array_equal(aidx, eidx) and np.array_equal(asc, esc)
    finally:
        g.close()
