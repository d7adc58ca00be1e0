"""mAP evaluation (revisited Oxford/Paris protocol) -- host-side counterpart of
src/utils/evaluate2.py:4-155.  Stays on the CPU like the reference: it consumes
the [K,Q] ranks the HIP path returns and is not part of the accelerated arithmetic.

Own vectorised implementation (not the reference's loops): for every query the positions of the
positives in the ranked list are shifted up by the number of junk entries ranked before them, then
the trapezoid AP is evaluated in closed form.
"""
import numpy as np


def average_precision(pos, nres):
    """pos: zero-based (junk-corrected) ranks of the positives found, ascending; nres: #positives.
    AP = sum_j ((j / pos_j  if pos_j > 0 else 1) + (j + 1) / (pos_j + 1)) / (2 * nres)."""
    pos = np.asarray(pos, dtype=np.float64)
    if pos.size == 0:
        return 0.0
    j = np.arange(pos.size, dtype=np.float64)
    p0 = np.where(pos == 0, 1.0, j / np.where(pos == 0, 1.0, pos))
    p1 = (j + 1.0) / (pos + 1.0)
    return float(((p0 + p1) / (2.0 * nres)).sum())


def compute_map(ranks, gnd):
    """ranks [K,Q] (K may be < N: unseen positives simply lower the AP); gnd[i] = {'ok', 'junk'}.
    Returns (map, aps); queries without positives are excluded (aps = nan)."""
    ranks = np.asarray(ranks)
    nq = len(gnd)
    aps = np.full(nq, np.nan)
    for i in range(nq):
        ok = np.asarray(gnd[i]["ok"])
        if ok.size == 0:
            continue
        junk = np.asarray(gnd[i].get("junk", np.empty(0)))
        col = ranks[:, i]
        is_ok = np.isin(col, ok)
        is_junk = np.isin(col, junk)
        pos = np.flatnonzero(is_ok) - np.cumsum(is_junk)[is_ok]
        aps[i] = average_precision(pos, ok.size)
    valid = ~np.isnan(aps)
    return float(aps[valid].mean()), aps


def precision_at(ranks, gnd, kappas=(1, 5, 10)):
    """Mean precision at kappas of the first-generation evaluator (src/utils/evaluate.py:102-108, printed by
    src/main_retrieve.py:176 and src/main_train.py:705): with p = the junk-corrected places of a query's positives, 1-based,
    and kq = min(max(p), kappa): #(p <= kq) / kq.  Returns (pr [len(kappas)], prs [Q, len(kappas)]); queries without
    positives are excluded (nan rows).  A query with positives of which none is in a truncated ranking is an error, as in
    the reference (max of an empty array)."""
    ranks = np.asarray(ranks)
    kap = np.asarray(kappas, dtype=np.int64)
    prs = np.full((len(gnd), kap.size), np.nan)
    for i in range(len(gnd)):
        ok = np.asarray(gnd[i]["ok"])
        if ok.size == 0:
            continue
        col = ranks[:, i]
        is_ok = np.isin(col, ok)
        p = np.flatnonzero(is_ok) - np.cumsum(np.isin(col, np.asarray(gnd[i].get("junk", np.empty(0)))))[is_ok] + 1
        if p.size == 0:
            raise ValueError("query %d: none of its positives is in the ranking" % i)
        kq = np.minimum(p.max(), kap)
        prs[i] = (p[None, :] <= kq[:, None]).sum(axis=1) / kq
    valid = ~np.isnan(prs[:, 0]) if kap.size else np.zeros(len(gnd), dtype=bool)
    return (prs[valid].mean(axis=0) if kap.size else np.zeros(0)), prs


def map_custom(K, matching_idx, paths_q, paths_d):
    """mAP of a directory-labelled collection (src/utils/evaluate.py:157-174 `mAP_custom`, printed by src/test_custom.py:33):
    the label of an image is the name of the directory it lies in; AP = sum over the places j < K holding a same-label
    image of (#same-label images among places 0..j) / (j + 1), divided by min(#same-label database images, K)."""
    label_d = np.array([p.split("/")[-2] for p in paths_d])
    idx = np.asarray(matching_idx)[:, :K]
    total = 0.0
    for i, pq in enumerate(paths_q):
        same = label_d == pq.split("/")[-2]
        hit = same[idx[i]]
        total += float((np.cumsum(hit)[hit] / (np.flatnonzero(hit) + 1.0)).sum()) / min(int(same.sum()), K)
    return total / len(paths_q)


def compute_map_from_positions(pos_ok, pos_junk):
    """The same AP from the POSITIONS (zero-based places in the full ranking) of every positive and junk image of each
    query -- all that the reference's compute_map2 reads out of `ranks` [N, Q] (src/utils/evaluate2.py:73-86: np.in1d
    look-ups of the positive and junk ids).  pos_ok / pos_junk: per query 1-D arrays (entries < 0 are ignored).
    Equals compute_map on the complete ranking; the [N, Q] array never has to exist (Gallery.rank_positions)."""
    nq = len(pos_ok)
    aps = np.full(nq, np.nan)
    for i in range(nq):
        ok = np.sort(np.asarray(pos_ok[i])[np.asarray(pos_ok[i]) >= 0])
        if ok.size == 0:
            continue
        junk = np.sort(np.asarray(pos_junk[i])[np.asarray(pos_junk[i]) >= 0])
        pos = ok - np.searchsorted(junk, ok)                 # junk ranked before a positive does not count
        aps[i] = average_precision(pos, ok.size)
    valid = ~np.isnan(aps)
    return float(aps[valid].mean()), aps


EMH = ((("easy",), ("junk", "hard")), (("easy", "hard"), ("junk",)), (("hard",), ("junk", "easy")))


def compute_map_revisited_from_positions(gnd, position_of):
    """(mapE, mapM, mapH) from a per-query look-up `position_of[i][image id] -> position in the full ranking`."""
    out = []
    for ok_keys, junk_keys in EMH:
        ok = [np.array([position_of[i][int(v)] for k in ok_keys for v in np.asarray(x[k]).ravel()], dtype=np.int64)
              for i, x in enumerate(gnd)]
        junk = [np.array([position_of[i][int(v)] for k in junk_keys for v in np.asarray(x[k]).ravel()], dtype=np.int64)
                for i, x in enumerate(gnd)]
        out.append(compute_map_from_positions(ok, junk)[0])
    return tuple(out)


def compute_map_revisited(ranks, gnd):
    """(mapE, mapM, mapH) with the Easy / Medium / Hard label sets of the revisited protocol."""
    out = []
    for ok_keys, junk_keys in ((("easy",), ("junk", "hard")), (("easy", "hard"), ("junk",)),
                               (("hard",), ("junk", "easy"))):
        g = [{"ok": np.concatenate([np.asarray(x[k]) for k in ok_keys]),
              "junk": np.concatenate([np.asarray(x[k]) for k in junk_keys])} for x in gnd]
        out.append(compute_map(ranks, g)[0])
    return tuple(out)


def compute_map_and_print(dataset, ranks, gnd, kappas=None):
    """Print format of compute_map_and_print2 (src/utils/evaluate2.py:110-155); returns the values.  With `kappas` (the
    first-generation evaluator's default is [1, 5, 10], src/utils/evaluate.py:115) the mP@k line of
    src/utils/evaluate.py:149 follows the mAP line."""
    if dataset.startswith("roxford") or dataset.startswith("rparis"):
        e, m, h = compute_map_revisited(ranks, gnd)
        print(">> {}: mAP E: {}, M: {}, H: {}".format(dataset, np.around(e * 100, decimals=2),
                                                      np.around(m * 100, decimals=2), np.around(h * 100, decimals=2)))
        if kappas:
            prs = [precision_at(ranks, [{"ok": np.concatenate([np.asarray(x[k]) for k in okk]),
                                         "junk": np.concatenate([np.asarray(x[k]) for k in jk])} for x in gnd], kappas)[0]
                   for okk, jk in EMH]
            print(">> {}: mP@k{} E: {}, M: {}, H: {}".format(dataset, list(kappas), *[np.around(p * 100, decimals=2)
                                                                                      for p in prs]))
        return e, m, h
    mp, _ = compute_map(ranks, gnd)
    print(">> {}: mAP {:.2f}".format(dataset, np.around(mp * 100, decimals=2)))
    return mp
