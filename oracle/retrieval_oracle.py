"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the reference retrieval hot path.

Every function cites the reference file:line it restates (paths relative to /root/reference).
See oracle/__init__.py for which parts are pinned by golden vectors.
"""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

__all__ = [
    "matching_l2", "matching_fractional_dis", "fractional_distance", "ip_rank", "feature_enhancement", "qge1", "qe_weights", "l2n", "whitenapply",
    "extract_ms_tail", "knn_flat_ip", "knn_flat_ip_blas", "compute_ap2", "compute_map2", "compute_map_revisited", "compute_map_kappas", "map_custom",
    "get_affinity", "get_laplacian", "diffusion_offline", "diffusion_online", "qge_small",
    "average_query_expansion", "database_augmentation", "kr_reranking", "exact_scores_f64", "exact_topk_f64", "check_topk_parity", "merge_topk",
]


# --------------------------------------------------------------------------- a1
def matching_l2(K, gallery, queries):
    """a1: src/utils/nnsearch.py:687-706 (matching_L2).

    Rows of both matrices are divided by their L2 norm with NO eps (:693-698), then for every
    query the distances ||q - G|| over gallery rows are fully argsorted ascending and the
    first K kept (:699-703).  Arithmetic stays in the input dtype.  Timing is not restated.
    """
    gallery = np.asarray(gallery)
    queries = np.asarray(queries)
    gn = gallery / np.linalg.norm(gallery, axis=1)[:, None]
    qn = queries / np.linalg.norm(queries, axis=1)[:, None]
    out = np.zeros((qn.shape[0], K), dtype=np.int64)
    for r in range(qn.shape[0]):
        d = np.linalg.norm(qn[r, :] - gn, axis=1)
        out[r, :] = np.argsort(d)[:K]
    return out


def fractional_distance(x, y, p=0.5):
    """src/utils/nnsearch.py:46-56 -- dist[i, j] = (sum_k |x[i,k] - y[j,k]| ** p) ** (1/p) through an [N, M, d] temporary."""
    diff = np.abs(np.expand_dims(x, axis=1) - np.expand_dims(y, axis=0))
    return np.sum(diff ** p, axis=-1) ** (1 / p)


def matching_fractional_dis(K, gallery, queries):
    """src/utils/nnsearch.py:709-731 (matching_fractional_dis), called with p = 2, i.e. the ordering of matching_L2.

    Normalisation as in matching_L2 (no eps, :715-720), one [Q, N] distance matrix (:721), a full argsort along the
    gallery axis (:723) -- and then `[:K]` on the QUERY axis before `[:, :K]` (:723-724): the reference returns the
    rankings of the first K queries only, shape [min(Q, K), K].  Restated as written."""
    gallery = np.asarray(gallery)
    queries = np.asarray(queries)
    gn = gallery / np.expand_dims(np.linalg.norm(gallery, axis=1), axis=1)
    qn = queries / np.expand_dims(np.linalg.norm(queries, axis=1), axis=1)
    dist = fractional_distance(qn, gn, 2)
    idx = np.argsort(dist)[:K]
    return idx[:, :K]


# --------------------------------------------------------------------------- a2
def ip_rank(vecs, qvecs):
    """a2: src/main_retrieve.py:175-176 -- scores = vecs.T @ qvecs ; ranks = argsort(-scores, 0).

    vecs [D,N], qvecs [D,Q] (column per image).  Returns (ranks int64 [N,Q], scores [N,Q]).
    """
    scores = np.dot(vecs.T, qvecs)
    return np.argsort(-scores, axis=0), scores


# --------------------------------------------------------------------------- a3
def qe_weights(k, w):
    """src/utils/Reranking.py:197 -- ((k, k-1, ..., 1)/k) ** w, float64."""
    return (np.arange(k, 0, -1) / k) ** w


def feature_enhancement(k, ranks, vecs, w):
    """a3: src/utils/Reranking.py:195-208 (identical copy at :288-301).

    ranks [K_in,Q] int, vecs [D,N].  The expanded query is the weighted sum of the top-k
    gallery columns (the original query is NOT added, :203 is commented out), divided by
    (||.||_2 + 1e-6) (:204); all gallery columns are re-scored by inner product and fully
    argsorted descending (:206-207).  The reference's it_times loop re-reads the input ranks
    every pass, so any it_times >= 1 gives this single-pass result.
    Returns (q_exp [D,Q] float64, ranks_aqe int64 [N,Q]).
    """
    wts = qe_weights(k, w).reshape(1, k, 1)
    top = vecs[:, ranks[:k, :]]                       # [D,k,Q]
    qx = (top * wts).sum(axis=1)                      # [D,Q], promotes to float64
    qx = qx / (np.linalg.norm(qx, ord=2, axis=0, keepdims=True) + 1e-6)
    scores = np.dot(vecs.T, qx)
    return qx, np.argsort(-scores, axis=0)


def qge1(ranks, qvec, vecs, K):
    """src/utils/Reranking.py:287-306 -- k=3, w=4.0, one pass; K and qvec are unused there."""
    return feature_enhancement(3, ranks, vecs, 4.0)[1]


# --------------------------------------------------------------------------- a7 / a8
def l2n(x, eps=1e-6):
    """a7: src/layers/functional.py:129-130 -- x / (||x||_2 over dim 1 + eps), x [B,D,...]."""
    x = np.asarray(x)
    return x / (np.sqrt((x * x).sum(axis=1, keepdims=True)) + x.dtype.type(eps))


def extract_ms_tail(per_scale, msp=1.0):
    """a7: src/networks/imageretrievalnet.py:473-477 -- mean over scales of v**msp, **(1/msp),
    then v / ||v|| (no eps).  per_scale [S,D]."""
    v = np.zeros(per_scale.shape[1], dtype=per_scale.dtype)
    for s in range(per_scale.shape[0]):
        v = v + per_scale[s] ** msp
    v = v / per_scale.shape[0]
    v = v ** (1.0 / msp)
    return v / np.linalg.norm(v)


def whitenapply(X, m, P, dimensions=None):
    """a8: src/utils/whiten.py:4-12 -- P[:dims] @ (X - m), columns / (||.|| + 1e-6)."""
    if not dimensions:
        dimensions = P.shape[0]
    Y = np.dot(P[:dimensions, :], X - m)
    return Y / (np.linalg.norm(Y, ord=2, axis=0, keepdims=True) + 1e-6)


# --------------------------------------------------------------------------- a6
def knn_flat_ip(database, queries, k):
    """a6: src/utils/knn.py:8-40 -- contract of faiss.IndexFlatIP.search (third-party, absent):
    exact top-k float32 inner products per query, descending.  Tie order unspecified in faiss;
    here ties go to the lower index.  Returns (sims f32 [Q,k], ids int64 [Q,k])."""
    db = np.ascontiguousarray(database, dtype=np.float32)
    q = np.ascontiguousarray(queries, dtype=np.float32)
    s = q @ db.T
    order = np.argsort(-s, axis=1, kind="stable")[:, :k]
    return np.take_along_axis(s, order, axis=1), order.astype(np.int64)


def knn_flat_ip_blas(database, queries, k):
    """Same contract as knn_flat_ip (a6, src/utils/knn.py:8-40; also the scoring of src/main_retrieve.py:175-176) the
    way a tuned CPU library does it: one multi-threaded sgemm, then a partial selection (argpartition) and a sort of
    only k scores per query.  This is SURVEY.md 8d's "strongest fair CPU exhaustive baseline" and the stand-in for
    faiss-CPU IndexFlatIP, which is not installable offline.  Ties at the k-th score go to the lower index."""
    db = np.ascontiguousarray(database, dtype=np.float32)
    q = np.ascontiguousarray(queries, dtype=np.float32)
    s = q @ db.T
    n = s.shape[1]
    ids = np.empty((s.shape[0], k), dtype=np.int64)
    for i in range(s.shape[0]):
        row = s[i]
        if k < n:
            vk = -np.partition(-row, k - 1)[k - 1]                 # the k-th largest score
            above = np.flatnonzero(row > vk)
            ties = np.flatnonzero(row == vk)[: k - above.size]     # lowest indices first
            cand = np.concatenate([above, ties])
        else:
            cand = np.arange(n)
        order = np.lexsort((cand, -row[cand]))                     # score desc, index asc
        ids[i] = cand[order][:k]
    return np.take_along_axis(s, ids, axis=1), ids


# --------------------------------------------------------------------------- a9
def compute_ap2(pos, nres):
    """a9: src/utils/evaluate2.py:4-33 -- trapezoid AP from zero-based positive positions."""
    ap = 0.0
    step = 1.0 / nres
    for j in range(len(pos)):
        r = pos[j]
        p0 = 1.0 if r == 0 else float(j) / r
        p1 = float(j + 1) / (r + 1)
        ap += (p0 + p1) * step / 2.0
    return ap


def compute_map2(ranks, gnd):
    """a9: src/utils/evaluate2.py:36-107.  ranks [K,Q]; gnd[i] has 'ok' and optional 'junk'.
    Junk entries ranked before a positive shift that positive up.  Queries without positives
    are excluded from the mean.  Returns (map, aps)."""
    nq = len(gnd)
    aps = np.zeros(nq)
    total, nempty = 0.0, 0
    for i in range(nq):
        ok = np.array(gnd[i]["ok"])
        if ok.shape[0] == 0:
            aps[i] = float("nan")
            nempty += 1
            continue
        junk_ids = np.array(gnd[i]["junk"]) if "junk" in gnd[i] else np.empty(0)
        col = ranks[:, i]
        pos = np.flatnonzero(np.isin(col, ok))
        junk = np.flatnonzero(np.isin(col, junk_ids))
        if len(junk):
            shift = np.searchsorted(junk, pos)      # junk ranked strictly before each positive
            pos = pos - shift
        ap = compute_ap2(pos, len(ok))
        total += ap
        aps[i] = ap
    return total / (nq - nempty), aps


def compute_map_revisited(ranks, gnd):
    """a9: src/utils/evaluate2.py:118-145 -- Easy / Medium / Hard splits of the revisited protocol.
    Returns (mapE, mapM, mapH)."""
    out = []
    for ok_keys, junk_keys in ((("easy",), ("junk", "hard")),
                               (("easy", "hard"), ("junk",)),
                               (("hard",), ("junk", "easy"))):
        g = [{"ok": np.concatenate([gnd[i][k] for k in ok_keys]),
              "junk": np.concatenate([gnd[i][k] for k in junk_keys])} for i in range(len(gnd))]
        out.append(compute_map2(ranks, g)[0])
    return tuple(out)


def compute_map_kappas(ranks, gnd, kappas=(1, 5, 10)):
    """src/utils/evaluate.py:40-113 (the first-generation evaluator that src/main_retrieve.py:176 and
    src/main_train.py:705 print through): the same AP as compute_map2 plus precision at kappas -- with the junk-corrected
    positions made 1-based (:103), kq = min(max(pos), kappa) (:105) and prs = #(pos <= kq) / kq (:106).  A query whose
    positives are all missing from a truncated ranking raises in the reference (max of an empty array); here too.
    Returns (map, aps, pr, prs)."""
    nq = len(gnd)
    aps = np.zeros(nq)
    prs = np.zeros((nq, len(kappas)))
    total, nempty = 0.0, 0
    pr = np.zeros(len(kappas))
    for i in range(nq):
        ok = np.array(gnd[i]["ok"])
        if ok.shape[0] == 0:
            aps[i] = float("nan")
            prs[i, :] = float("nan")
            nempty += 1
            continue
        junk_ids = np.array(gnd[i]["junk"]) if "junk" in gnd[i] else np.empty(0)
        col = ranks[:, i]
        pos = np.flatnonzero(np.isin(col, ok))
        junk = np.flatnonzero(np.isin(col, junk_ids))
        if len(junk):
            pos = pos - np.searchsorted(junk, pos)
        ap = compute_ap2(pos, len(ok))
        total += ap
        aps[i] = ap
        pos1 = pos + 1
        for j, kappa in enumerate(kappas):
            kq = min(max(pos1), kappa)
            prs[i, j] = (pos1 <= kq).sum() / kq
        pr = pr + prs[i, :]
    return total / (nq - nempty), aps, pr / (nq - nempty), prs


def map_custom(K, matching_idx, paths_q, paths_d):
    """src/utils/evaluate.py:157-174 (`mAP_custom`, what src/test_custom.py:33 prints): the label of an image is the
    name of its directory; AP of query i = sum over the places j < K that hold a same-label image of
    (#same-label images up to j) / (j + 1), divided by min(#same-label images in the database, K)."""
    label_d = [p.split("/")[-2] for p in paths_d]
    total = 0.0
    for i in range(len(paths_q)):
        label_q = paths_q[i].split("/")[-2]
        tp = {j for j, v in enumerate(label_d) if v == label_q}
        count = 0
        matched = np.zeros(K, dtype=np.int64)
        for j in range(K):
            if int(matching_idx[i, j]) in tp:
                count += 1
                matched[j] = count
        total += sum(matched / (np.arange(K) + 1)) / min(len(tp), K)
    return total / len(paths_q)


# --------------------------------------------------------------------------- a5
def get_affinity(sims, ids, gamma=3):
    """a5: src/utils/diffusion.py:101-116.  sims/ids [N,kd].  Negative sims are clipped to 0,
    raised to gamma; (i, ids[i,j]) is kept iff i appears in the kd-list of ids[i,j] (mutual),
    position 0 is always dropped (:108).  CSC float32 [N,N]."""
    num = sims.shape[0]
    s = np.where(sims < 0, 0, sims) ** gamma
    rr, cc, vv = [], [], []
    for i in range(num):
        mutual = np.isin(ids[ids[i]], i).any(axis=1)
        mutual[0] = False
        if mutual.any():
            rr.append(np.full(int(mutual.sum()), i, dtype=int))
            cc.append(ids[i, mutual])
            vv.append(s[i, mutual])
    rr, cc, vv = map(np.concatenate, (rr, cc, vv))
    return sp.csc_matrix((vv, (rr, cc)), shape=(num, num), dtype=np.float32)


def get_laplacian(sims, ids, alpha=0.99):
    """a5: src/utils/diffusion.py:87-98 -- I - alpha * D^-1/2 A D^-1/2 with deg = A@1 + 1e-12."""
    aff = get_affinity(sims, ids)
    num = aff.shape[0]
    deg = aff @ np.ones(num) + 1e-12
    dm = sp.dia_matrix((deg ** (-0.5), [0]), shape=(num, num), dtype=np.float32)
    stoch = dm @ aff @ dm
    eye = sp.dia_matrix((np.ones(num), [0]), shape=(num, num), dtype=np.float32)
    return eye - alpha * stoch


def diffusion_offline(features, n_trunc, kd=50, return_parts=False):
    """a5: src/utils/diffusion.py:52-84 (N < 110000 branch) + :15-19.

    kNN graph by exact inner product (faiss IndexFlatIP stand-in), Laplacian on the first kd
    columns, then for every node a CG solve of lap[ids_i][:, ids_i] x = e0 with
    rtol=1e-6 (legacy tol=1e-6, ||b|| = 1), at most 20 iterations, result ignored on
    non-convergence like the reference (info is dropped).  CSR float32 [N,N]."""
    n = len(features)
    sims, ids = knn_flat_ip(features, features, n_trunc)
    lap = sp.csr_matrix(get_laplacian(sims[:, :kd].copy(), ids[:, :kd]))
    b = np.zeros(n_trunc)
    b[0] = 1
    allsc = np.empty((n, n_trunc))
    for i in range(n):
        sub = lap[ids[i]][:, ids[i]]
        allsc[i], _ = spla.cg(sub, b, rtol=1e-6, atol=0.0, maxiter=20)
    rows = np.repeat(np.arange(n), n_trunc)
    off = sp.csr_matrix((allsc.reshape(-1), (rows, ids.reshape(-1))), shape=(n, n), dtype=np.float32)
    if return_parts:
        return off, sims, ids, lap, allsc
    return off


def diffusion_online(q_for_search, features, offline, k_query=3, truncation_number=2000):
    """a4: src/utils/Reranking.py:238-253 -- top-k_query gallery neighbours of each query, sims**3,
    scores = sims[i] @ offline[idx[i]], top-`truncation_number` by argpartition then argsort.
    q_for_search [Q,D].  Returns (ranks_dfs [trunc,Q] int64, scores [Q,trunc] float32)."""
    sims, idx = knn_flat_ip(features, q_for_search, k_query)
    sims = sims ** 3
    nq = idx.shape[0]
    tr_s = np.empty((nq, truncation_number), dtype=np.float32)
    tr_r = np.empty((nq, truncation_number), dtype=np.int64)
    for i in range(nq):
        sc = np.asarray(sims[i] @ offline[idx[i]]).ravel()
        parts = np.argpartition(-sc, truncation_number)[:truncation_number]
        o = np.argsort(-sc[parts])
        tr_s[i] = sc[parts][o]
        tr_r[i] = parts[o]
    return tr_r.T, tr_s


def qge_small(ranks, qvecs, vecs, AQE=True, truncation_number=2000, k_gallery=200, k_query=3):
    """a4: src/utils/Reranking.py:212-253 (N < 120000): alpha-QE with k=10, w=4 and then the
    diffusion re-search with the expanded (AQE) or original queries.  Returns
    (qvecs_qe, ranks_aqe, ranks_dfs)."""
    qx, ranks_aqe = feature_enhancement(10, ranks, vecs, 4.0)
    feats = np.ascontiguousarray(vecs.T)
    off = diffusion_offline(feats, truncation_number, k_gallery)
    qs = qx.T if AQE else qvecs.T
    ranks_dfs, _ = diffusion_online(qs, feats, off, k_query, truncation_number)
    return qx, ranks_aqe, ranks_dfs


# --------------------------------------------------------------------------- f-4
def _centre_normalise(q, r):
    """src/utils/Reranking.py:315-336 (identical copy :376-402): subtract the mean of the concatenated rows, then divide
    every row by its norm -- unless ANY norm of that matrix is zero, in which case it is returned un-normalised."""
    c = np.mean(np.concatenate([q, r], axis=0), axis=0)

    def l2(v):
        nrm = np.expand_dims(np.linalg.norm(v, axis=1), axis=1)
        return v if np.any(nrm == 0) else v / nrm
    return l2(q - c), l2(r - c)


def _dist_rank(q, r):
    qn, rn = _centre_normalise(q, r)
    return np.argsort(2 - 2 * np.dot(qn, rn.T), axis=1)


def kr_reranking(qvecs, vecs, k1=20, k2=6, lambda_value=0.3, block=6000, return_dist=False):
    """f-4: src/utils/Reranking.py:447-624 (kr_reranking; k-reciprocal encoding, Zhong et al. CVPR 2017), restated in numpy
    float32 with the reference's constants (k1 = 20, k2 = 6, lambda = 0.3, :546-548) and its own blocking (N = 6000).

      feat = [queries; gallery] rows, assumed L2-normalised ("For L2-norm feature", :461-462)
      dist(a, b) = 2 - 2 a.b (:462); every column block is divided by its column maxima (:501), which changes no order
      initial_rank = the k1 + 1 nearest of every row among ALL rows, itself included (:502, :555)
      R[i] = k-reciprocal set of i, expanded by the k1/2-reciprocal sets of its members that overlap it by > 2/3 (:565-575)
      V[i, R[i]] = softmax(-dist(i, R[i]) / max_j dist(i, j)) (:514-525)
      V <- mean of the rows of the k2 nearest, stored as FLOAT16 (:585-589)
      jaccard[i, j] = 1 - m / (2 - m), m = sum_c min(V[i, c], V[j, c]) accumulated in float32 over ascending c (:602-609)
      final = 0.7 * jaccard + 0.3 * dist / colmax (:612-613); queries x gallery block, argsort ascending (:618, :621)

    Returns indices int64 [Q, N] (as the reference: `return indices`, NOT transposed)."""
    probe = np.asarray(qvecs.T, dtype=np.float32)
    gal = np.asarray(vecs.T, dtype=np.float32)
    query_num = probe.shape[0]
    feat = np.concatenate([probe, gal])
    all_num = feat.shape[0]

    def euclid(qf, gf):
        return (2 - 2 * (qf @ gf.T)).astype(np.float32)

    def normalised_block(gf_block):                       # [m, nj] -> / column max -> [nj, m]   (:492-502, :470-480)
        d = np.concatenate([euclid(feat[i:i + block], gf_block) for i in range(0, all_num // block * block + 1, block)
                            if i < all_num], axis=0)
        d = d / d.max(axis=0)
        return d.T

    initial_rank = np.concatenate(
        [np.argsort(normalised_block(feat[j:j + block]), axis=1, kind="stable")[:, :k1 + 1]
         for j in range(0, all_num // block * block + 1, block) if j < all_num], axis=0)

    def k_reciprocal_neigh(i, k):
        fwd = initial_rank[i, :k + 1]
        back = initial_rank[fwd, :k + 1]
        fi = np.where(back == i)[0]
        return fwd[fi]

    R = []
    for i in range(all_num):
        kr = k_reciprocal_neigh(i, k1)
        exp_idx = kr
        for c in kr:
            ckr = k_reciprocal_neigh(c, int(np.around(k1 / 2)))
            if len(np.intersect1d(ckr, kr)) > 2. / 3 * len(ckr):
                exp_idx = np.append(exp_idx, ckr)
        R.append(np.unique(exp_idx))

    V = np.zeros((all_num, all_num), dtype=np.float32)
    for i in range(all_num):
        d = euclid(feat[i:i + 1], feat)
        d = (d / d.max()).reshape(-1)[R[i]]
        w = np.exp(-d)
        V[i, R[i]] = (w / w.sum()).astype(np.float32)
    ir = initial_rank[:, :k2]
    if k2 != 1:
        V_qe = np.zeros_like(V, dtype=np.float16)
        for i in range(all_num):
            V_qe[i, :] = np.mean(V[ir[i], :], axis=0)
        V = V_qe
    inv_index = [np.where(V[:, i] != 0)[0] for i in range(all_num)]
    jaccard = np.zeros((query_num, all_num), dtype=np.float32)
    for i in range(query_num):
        temp_min = np.zeros((1, all_num), dtype=np.float32)
        nz = np.where(V[i, :] != 0)[0]
        for c in nz:
            temp_min[0, inv_index[c]] = temp_min[0, inv_index[c]] + np.minimum(V[i, c], V[inv_index[c], c])
        jaccard[i] = 1 - temp_min / (2. - temp_min)
    original = normalised_block(feat[:query_num])          # batch_euclidean_distance(feat, feat[:query_num])  (:611)
    final = jaccard * (1 - lambda_value) + original * lambda_value
    final = final[:query_num, query_num:]
    idx = np.argsort(final, axis=1, kind="stable")
    return (idx, final) if return_dist else idx


def average_query_expansion(qvecs, vecs, K, top_k=3):
    """f-4: src/utils/Reranking.py:314-365.  Queries and gallery rows are extended by the mean of their top-3
    neighbours (gallery: positions 1..3, skipping itself) found on centred + normalised vectors; the 2D-dim
    concatenations are matched with matching_L2.  Returns ranks [K,Q]."""
    q, r = qvecs.T, vecs.T
    idx = _dist_rank(q, r)
    qn = np.concatenate([q, np.mean(r[idx[:, :top_k], :], axis=1)], axis=1)
    idx = _dist_rank(r, r)
    rn = np.concatenate([r, np.mean(r[idx[:, 1:top_k + 1], :], axis=1)], axis=1)
    return matching_l2(K, rn, qn).T


def database_augmentation(qvecs, vecs, K, top_k=3):
    """f-4: src/utils/Reranking.py:375-432.  weights = logspace(0, -2, 4); query' = w0 q + sum_j w_j r_j (top-3),
    row' = sum_j w_j r_j over its top-4 (itself first); matching_L2 on the augmented vectors."""
    q, r = qvecs.T, vecs.T
    w = np.logspace(0, -2., top_k + 1)
    idx = _dist_rank(q, r)
    qn = np.tensordot(w, np.concatenate([np.expand_dims(q, 1), r[idx[:, :top_k], :]], axis=1), axes=(0, 1))
    idx = _dist_rank(r, r)
    rn = np.tensordot(w, r[idx[:, :top_k + 1], :], axes=(0, 1))
    return matching_l2(K, rn, qn).T


# --------------------------------------------------------------------------- checkers
def exact_scores_f64(gallery, queries, normalize=True):
    """Ground truth for the tolerance checks: float64 cosine (or raw IP) scores [Q,N]."""
    g = np.asarray(gallery, dtype=np.float64)
    q = np.asarray(queries, dtype=np.float64)
    if normalize:
        g = g / np.linalg.norm(g, axis=1)[:, None]
        q = q / np.linalg.norm(q, axis=1)[:, None]
    return q @ g.T


def exact_topk_f64(gallery, queries, K, normalize=True):
    """Exact top-K by float64 score, ties to the lower index.  (idx [Q,K], scores [Q,K])."""
    s = exact_scores_f64(gallery, queries, normalize)
    order = np.argsort(-s, axis=1, kind="stable")[:, :K]
    return order.astype(np.int64), np.take_along_axis(s, order, axis=1)


def check_topk_parity(idx, scores_f64, K, tau):
    """Near-tie tolerant parity.  `idx` [Q,K] from the system under test,
    `scores_f64` [Q,N] ground-truth scores (higher = better).  For every query:
      * no duplicates;
      * every returned item scores >= s_K - tau  (s_K = K-th best ground-truth score);
      * every item scoring > s_K + tau is returned;
      * the returned order is non-increasing in ground-truth score up to tau.
    Returns a list of violation strings (empty = parity)."""
    bad = []
    idx = np.asarray(idx)
    for q in range(idx.shape[0]):
        s = scores_f64[q]
        sk = np.partition(s, -K)[-K]
        got = idx[q, :K]
        if len(np.unique(got)) != K:
            bad.append(f"q{q}: duplicate indices")
            continue
        gs = s[got]
        if (gs < sk - tau).any():
            bad.append(f"q{q}: returned item below K-th score by {float((sk - gs).max()):.3e}")
        must = np.flatnonzero(s > sk + tau)
        miss = np.setdiff1d(must, got)
        if len(miss):
            bad.append(f"q{q}: missing {len(miss)} items clearly inside the top-K")
        if (np.diff(gs) > tau).any():
            bad.append(f"q{q}: order violates ground truth by {float(np.diff(gs).max()):.3e}")
    return bad


def merge_topk(scores_list, idx_list, K):
    """Checker for the multi-shard merge: lists of per-shard (scores [Q,k], global idx [Q,k]) ->
    top-K by (score desc, idx asc)."""
    s = np.concatenate(scores_list, axis=1)
    i = np.concatenate(idx_list, axis=1)
    out_s = np.empty((s.shape[0], K), dtype=s.dtype)
    out_i = np.empty((s.shape[0], K), dtype=np.int64)
    for q in range(s.shape[0]):
        o = np.lexsort((i[q], -s[q].astype(np.float64)))[:K]
        out_s[q], out_i[q] = s[q][o], i[q][o]
    return out_s, out_i
