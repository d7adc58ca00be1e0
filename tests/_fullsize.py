"""Full-size checks against the oracle ITSELF: every stored row of a device gallery scored against a few
queries in float64 ON THE HOST -- numpy, `oracle.exact_scores_f64` / `oracle.exact_topk_f64` on 64 k-row chunks of
`Gallery.get_rows()`, `oracle.merge_topk` over the chunks -- and the library's answer judged by `oracle.check_topk_parity`.
What the reference computes per query (src/utils/nnsearch.py:699-703: distance to every normalised row, full argsort) is the
same ranking: ||q - g||^2 = 2 - 2 q.g on unit vectors."""
from concurrent.futures import ThreadPoolExecutor

import numpy as np

import oracle

CHUNK = 65536


def host_f64_scores_and_topk(g, q_host, k, normalize_queries=True, workers=8):
    """-> (scores float64 [Q, N] of all stored rows, oracle top-k idx [Q, k], oracle top-k scores [Q, k]).
    g: isehr_amd._lib.Gallery (one shard, row_offset 0); q_host [Q, D] raw queries."""
    q = np.asarray(q_host, dtype=np.float64)
    if normalize_queries:
        q = q / np.linalg.norm(q, axis=1)[:, None]
    n = g.n
    starts = list(range(0, n, CHUNK))

    def one(r0):
        m = min(CHUNK, n - r0)
        rows = g.get_rows(r0, m)                                  # the gallery as stored: f32 rows, normalised at ingest
        # (normalize=False: the stored rows are what the search ranks by; their norms are 1 to a few 1e-8 and are checked
        # against the reference's definition row by row in `stored_rows_are_the_normalised_raw_rows`)
        s = oracle.exact_scores_f64(rows, q, normalize=False)     # [Q, m] float64
        ti, ts = oracle.exact_topk_f64(rows, q, min(k, m), normalize=False)
        return r0, s, ti + r0, ts
    scores = np.empty((q.shape[0], n), dtype=np.float64)
    parts_i, parts_s = [], []
    with ThreadPoolExecutor(max_workers=workers) as ex:
        for r0, s, ti, ts in ex.map(one, starts):
            scores[:, r0:r0 + s.shape[1]] = s
            parts_i.append(ti)
            parts_s.append(ts)
    top_s, top_i = oracle.merge_topk(parts_s, parts_i, k)
    return scores, top_i, top_s


def assert_oracle_parity(idx, sc, scores, top_i, top_s, k, tau=1e-6):
    """idx / sc [Q, k]: the library's answer.  Parity as SURVEY 8c defines it, against float64 ground truth of EVERY row."""
    assert oracle.check_topk_parity(idx, scores, k, tau) == []
    got = np.take_along_axis(scores, idx, axis=1)
    assert np.abs(got - top_s).max() <= tau                       # position by position the same scores as the oracle's ranking
    assert np.abs(got - sc).max() <= 3e-7                         # and the returned scores are those float64 values
    # rows that differ from the oracle's at a position are near-ties; everywhere else the ids agree
    differ = idx != top_i
    if differ.any():
        assert np.abs(np.take_along_axis(scores, idx, 1)[differ] - np.take_along_axis(scores, top_i, 1)[differ]).max() <= tau


def stored_rows_are_the_normalised_raw_rows(g, raw_rows, row0):
    """The reference divides every gallery row by its L2 norm, no eps (src/utils/nnsearch.py:693-698): the stored rows
    [row0, row0 + m) must be that, computed in float64 and rounded to float32 (<= 1 ulp of the row's largest element)."""
    raw = np.asarray(raw_rows, dtype=np.float64)
    want = raw / np.linalg.norm(raw, axis=1)[:, None]
    got = g.get_rows(row0, raw.shape[0]).astype(np.float64)
    assert np.abs(got - want).max() <= 6e-8 * np.abs(want).max() + 1e-12
