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


def compute_map_and_print(dataset, ranks, gnd):
    """Print format of compute_map_and_print2 (src/utils/evaluate2.py:110-155); returns the values."""
    if dataset.startswith("roxford") or dataset.startswith("rparis"):
        e, m, h = compute_map_revisited(ranks, gnd)
        print(">> {}: mAP E: {}, M: {}, H: {}".format(dataset, np.around(e * 100, decimals=2),
                                                      np.around(m * 100, decimals=2), np.around(h * 100, decimals=2)))
        return e, m, h
    mp, _ = compute_map(ranks, gnd)
    print(">> {}: mAP {:.2f}".format(dataset, np.around(mp * 100, decimals=2)))
    return mp
