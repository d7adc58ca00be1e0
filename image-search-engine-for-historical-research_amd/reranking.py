"""Host-side mirror of the reference's QGE re-ranking (src/utils/Reranking.py:194-306) for the HIP path.

    qge1_hip(ranks[K_in,Q], qvec[D,Q], vecs[D,N], K)            (src/utils/Reranking.py:287-306)
    feature_enhancement_hip(k, ranks, vecs, w, K)               (src/utils/Reranking.py:195-208)
    QGE_hip(ranks, qvecs, vecs, dataset, gnd, cache_dir, gnd_path2, AQE)   (src/utils/Reranking.py:194-285)

alpha query expansion: the expanded query is the ((k-j)/k)^w-weighted sum of the top-k gallery
columns (the original query is not added), divided by (||.|| + 1e-6); the gallery is then re-scored
by raw inner product (`vecs` is used as given, like `np.dot(vecs.T, qvecs_qe)`).  The reference
returns the full argsort [N,Q]; its callers consume the first K rows (src/online.py:148-152) or
feed compute_map (src/utils/Reranking.py:282), so this path returns the exact top-K rows [K,Q]
(full-length ranking is SURVEY.md §8 f-3).  Gather, weighted sum, normalisation and search all
run on the GPU.
"""
import numpy as np

from . import evaluate
from ._lib import NORM_NONE
from .nnsearch import get_gallery


def feature_enhancement_hip(k, ranks, vecs, w, K, dataset=None, ifgenerate=False, device=0, return_scores=False):
    """-> (qvecs_qe float64 [D,Q], ranks_aqe int64 [K,Q]).  `vecs` [D,N] or a prepared Gallery."""
    g = vecs if hasattr(vecs, "aqe_search") else get_gallery(np.asarray(vecs).T, dataset, ifgenerate, NORM_NONE,
                                                             device)
    try:
        idx, sc, qx, _ = g.aqe_search(np.asarray(ranks)[:k], k, w, int(K), eps=1e-6, return_qexp=True)
    finally:
        if g is not vecs and dataset is None:
            g.close()
    if return_scores:
        return qx.T, idx.T, sc.T
    return qx.T, idx.T


def qge1_hip(ranks, qvec, vecs, K, dataset=None, ifgenerate=False, device=0, full=False):
    """Online single-query re-ranking: k = 3, w = 4.0, one pass (src/utils/Reranking.py:302-305).
    Returns ranks_aqe[:K] (int64 [K,Q]); `qvec` is unused, as in the reference.  full=True returns the
    reference's complete [N,Q] ranking (full-length radix-sort path)."""
    if not full:
        return feature_enhancement_hip(3, ranks, vecs, 8.0 / 2, K, dataset, ifgenerate, device)[1]
    g = get_gallery(np.asarray(vecs).T, dataset, ifgenerate, NORM_NONE, device)
    try:
        kq = min(int(K), 16) if K else 16
        _, _, qx, _ = g.aqe_search(np.asarray(ranks)[:3], 3, 8.0 / 2, kq, eps=1e-6, return_qexp=True)
        idx, _ = g.rank_all(qx, query_norm=NORM_NONE)          # the expanded query is used as is
    finally:
        if dataset is None:
            g.close()
    return idx.T


def QGE_hip(ranks, qvecs, vecs, dataset, gnd, cache_dir=None, gnd_path2=None, AQE=True, K=None, device=0,
            quiet=False):
    """QGE driver.  N >= 120000: alpha-QE with k=3, w=4 and mAP print (src/utils/Reranking.py:273-284).
    N < 120000: alpha-QE with k=10, w=4 followed by truncated graph diffusion
    (src/utils/Reranking.py:212-264, implemented in diffusion.py).  Returns a dict with the rankings and
    mAP values (the reference prints and returns None)."""
    vecs = np.asarray(vecs)
    n = vecs.shape[1]
    out = {}
    if n >= 120000:
        Kq = int(K) if K else min(n, 1000)
        qx, ranks_aqe = feature_enhancement_hip(3, ranks, vecs, 8.0 / 2, Kq, None, False, device)
        out.update(qvecs_qe=qx, ranks_aqe=ranks_aqe)
        if gnd is not None:
            if not quiet:
                print("mAP after Enhancement")
            out["map_aqe"] = evaluate.compute_map_and_print(dataset, ranks_aqe, gnd) if not quiet else \
                evaluate.compute_map_revisited(ranks_aqe, gnd)
        return out
    from . import diffusion
    return diffusion.qge_small_hip(ranks, qvecs, vecs, dataset, gnd, AQE=AQE, K=K, device=device, quiet=quiet)
