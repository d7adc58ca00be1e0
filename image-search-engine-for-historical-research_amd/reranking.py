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


# ---------------------------------------------------------------------------------------------------------------
# f-4: average query expansion / database augmentation (src/utils/Reranking.py:314-432).  Compositions of the exact
# kNN path: centring (device column sums + the f64 whitening GEMM with P = I, eps = 0), exact top-k on the centred,
# normalised vectors, float64 weighted gathers of the ORIGINAL vectors, and matching_HIP on the augmented features.
def _centre_normalise_hip(q, r, device=0):
    from . import _lib
    c = (_lib.column_sum(q, device) + _lib.column_sum(r, device)) / (q.shape[0] + r.shape[0])
    eye = np.eye(q.shape[1])
    qn = _lib.whiten_apply(q, c, eye, q.shape[1], eps=0.0, device=device)
    rn = _lib.whiten_apply(r, c, eye, r.shape[1], eps=0.0, device=device)
    if not (np.isfinite(qn).all() and np.isfinite(rn).all()):
        raise RuntimeError("a vector coincides with the centre (zero norm after centring): the reference leaves the "
                           "whole matrix un-normalised in that case (src/utils/Reranking.py:321-324); not supported")
    return qn, rn


def _neighbours_hip(q, r, k, device=0):
    from ._lib import Gallery
    qn, rn = _centre_normalise_hip(q, r, device)
    g = Gallery.from_host(rn, norm_mode=NORM_NONE, device=device)
    try:
        idx, _, _ = g.search(qn, k)
    finally:
        g.close()
    return idx


def average_query_expansion_hip(qvecs, vecs, K, top_k=3, device=0):
    """average_query_expansion (src/utils/Reranking.py:314-365) -> ranks int64 [K,Q] (the reference prints mAP)."""
    from ._lib import Gallery
    from .nnsearch import matching_HIP
    q, r = np.ascontiguousarray(np.asarray(qvecs).T), np.ascontiguousarray(np.asarray(vecs).T)
    orig = Gallery.from_host(r, norm_mode=NORM_NONE, device=device)
    try:
        w = np.full(top_k, 1.0 / top_k)
        q_new = np.concatenate([q, orig.gather_weighted(_neighbours_hip(q, r, top_k, device).T, w)], axis=1)
        nb = _neighbours_hip(r, r, top_k + 1, device)[:, 1:]                     # positions 1..top_k: skip itself
        r_new = np.concatenate([r, orig.gather_weighted(nb.T, w)], axis=1)
    finally:
        orig.close()
    return matching_HIP(K, r_new, q_new, device=device)[0].T


def database_augmentation_hip(qvecs, vecs, K, top_k=3, device=0):
    """database_augmentation (src/utils/Reranking.py:375-432) -> ranks int64 [K,Q]."""
    from ._lib import Gallery
    from .nnsearch import matching_HIP
    q, r = np.ascontiguousarray(np.asarray(qvecs).T), np.ascontiguousarray(np.asarray(vecs).T)
    w = np.logspace(0, -2., top_k + 1)
    orig = Gallery.from_host(r, norm_mode=NORM_NONE, device=device)
    try:
        q_new = w[0] * q + orig.gather_weighted(_neighbours_hip(q, r, top_k, device).T, w[1:])
        r_new = orig.gather_weighted(_neighbours_hip(r, r, top_k + 1, device).T, w)
    finally:
        orig.close()
    return matching_HIP(K, r_new, q_new, device=device)[0].T
