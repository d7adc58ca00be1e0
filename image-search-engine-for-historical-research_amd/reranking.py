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
    from .nnsearch import ColumnBlocks
    blocks = isinstance(vecs, ColumnBlocks)       # [D, N] given as the column blocks the reference concatenates on the host
    if not blocks:
        vecs = np.asarray(vecs)
    n = vecs.shape[0] if blocks else vecs.shape[1]
    out = {}
    if n >= 120000:
        Kq = int(K) if K else min(n, 1000)
        g = get_gallery(vecs if blocks else vecs.T, None, False, NORM_NONE, device)
        try:
            qx, ranks_aqe = feature_enhancement_hip(3, ranks, g, 8.0 / 2, Kq, None, False, device)
            out.update(qvecs_qe=qx, ranks_aqe=ranks_aqe)
            if gnd is not None:
                # The reference evaluates the COMPLETE ranking ranks_aqe [N, Q] (src/utils/Reranking.py:206-207, 280-283), so
                # a positive ranked deeper than any top-K still counts.  compute_map2 only looks up where the labelled
                # images sit (src/utils/evaluate2.py:73-86): their positions in the full ranking are counted on the device
                # (mi_rank_positions) and give the same mAP without an [N, Q] array.
                keys = ("easy", "hard", "junk") if "easy" in gnd[0] else ("ok", "junk")
                ids = [np.unique(np.concatenate([np.asarray(x.get(k, []), dtype=np.int64).ravel() for k in keys])) for x in gnd]
                m = max(1, max(len(v) for v in ids))
                table = np.full((len(gnd), m), -1, dtype=np.int64)
                for i, v in enumerate(ids):
                    table[i, :len(v)] = v
                pos = g.rank_positions(np.ascontiguousarray(qx.T), table, query_norm=NORM_NONE)   # expanded queries as is
                position_of = [dict(zip(ids[i].tolist(), pos[i, :len(ids[i])].tolist())) for i in range(len(gnd))]
                out["positions"] = position_of
                if "easy" in gnd[0]:
                    vals = evaluate.compute_map_revisited_from_positions(gnd, position_of)
                else:
                    ok = [np.array([position_of[i][int(v)] for v in np.asarray(x["ok"]).ravel()]) for i, x in enumerate(gnd)]
                    jk = [np.array([position_of[i][int(v)] for v in np.asarray(x.get("junk", [])).ravel()])
                          for i, x in enumerate(gnd)]
                    vals = evaluate.compute_map_from_positions(ok, jk)[0]
                out["map_aqe"] = vals
                if not quiet:
                    print("mAP after Enhancement")
                    if isinstance(vals, tuple):
                        print(">> {}: mAP E: {}, M: {}, H: {}".format(dataset, *(np.around(v * 100, decimals=2) for v in vals)))
                    else:
                        print(">> {}: mAP {:.2f}".format(dataset, np.around(vals * 100, decimals=2)))
        finally:
            g.close()
        return out
    from . import diffusion
    if blocks:
        vecs = np.concatenate([np.asarray(b) for b in vecs.blocks], axis=1)
    # cache_dir: the reference's Diffusion(vecs.T, cache_dir) keeps offline.jbl there (src/utils/Reranking.py:234-235)
    return diffusion.qge_small_hip(ranks, qvecs, vecs, dataset, gnd, AQE=AQE, K=K, cache_dir=cache_dir, device=device,
                                   quiet=quiet)


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


def kr_reranking_hip(qvecs, vecs, k1=20, k2=6, lambda_value=0.3, device=0, return_dist=False):
    """kr_reranking(qvecs, vecs) (src/utils/Reranking.py:447-624): k-reciprocal re-ranking of all queries against the
    whole database; qvecs [D, Q], vecs [D, N] L2-normalised columns as there.  Returns `indices` int64 [Q, N] like the
    reference (`return indices`, :623: not transposed).  All x all inner products, reciprocal sets, the float16 encodings
    and the Jaccard pass run on the GPU (csrc/kr_rerank.hip)."""
    from . import _lib
    return _lib.kr_rerank(np.asarray(qvecs).T, np.asarray(vecs).T, k1, k2, lambda_value, device, return_dist)
