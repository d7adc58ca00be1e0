"""GPU: BASELINE.json's full size on STRUCTURED data: the clustered gallery of bench.py's `hard_data` block --
2000 clusters of 500 rows stored cluster by cluster, queries at cosine 0.95 to a centre, five per cluster, ordered like the
gallery, so that the K-th best score lies INSIDE a cluster -- against the oracle itself: every one of the 1 005 994 stored rows
scored in float64 on the host for 8 queries (tests/_fullsize.py), the library's answers judged by oracle.check_topk_parity.
The reference's cost does not depend on the data (src/utils/nnsearch.py:693-703); this path's does, and round 5 answered every
1024-query batch of this gallery through a chain of fallbacks."""
import os
import sys

import numpy as np
import pytest

from _fullsize import assert_oracle_parity, host_f64_scores_and_topk

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, D, K = 1005994, 2048, 100


def _search(g, q, nq):
    import torch
    idx = torch.empty((nq, K), dtype=torch.int64, device=q.device)
    sc = torch.empty((nq, K), dtype=torch.float32, device=q.device)
    g.search_device(q.data_ptr(), nq, K, idx.data_ptr(), sc.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return idx.cpu().numpy(), sc.cpu().numpy()


def test_clustered_gallery_against_the_oracle_itself():
    import torch
    from isehr_amd import _lib
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import bench                                     # the generator of the `hard_data` block (data only)
    dev = torch.device("cuda", 0)
    raw, queries, _ = bench._hard_rows("clustered", N, D, dev, 1234 + 501)
    torch.cuda.synchronize()                         # torch wrote the rows on ITS stream; the ingest runs on the handle's own
    g = _lib.Gallery.from_device_ptr(raw.data_ptr(), N, D)
    del raw
    torch.cuda.empty_cache()
    try:
        q = queries(1024).contiguous()
        pick = np.array([0, 1, 4, 5, 77, 300, 640, 1023])
        scores, top_i, top_s = host_f64_scores_and_topk(g, q.cpu().numpy()[pick], K)
        # the K-th best row of every query sits inside its cluster (500 rows at 0.855 .. 0.94), far above the background
        assert top_s[:, K - 1].min() > 0.8 and (np.sort(scores, axis=1)[:, -501] < 0.3).all()
        g.status(reset=True)
        for nq in (1024, 70):                        # the tile kernel (record segments spill: DESIGN 4.1) and the streaming kernel
            idx, sc = _search(g, q, nq)
            sel = pick[pick < nq]
            assert_oracle_parity(idx[sel], sc[sel], scores[:len(sel)], top_i[:len(sel)], top_s[:len(sel)], K)
        idx1, sc1 = _search(g, q[5:6].contiguous(), 1)
        assert_oracle_parity(idx1, sc1, scores[3:4], top_i[3:4], top_s[3:4], K)
        st = g.status()
        assert g.flags() == 0 and st["overflow_batches"] == 0 and st["spec_retries"] == 0
        # the fallbacks of the host entry point on the same gallery: the rigorous chunk schedule may overflow on a gallery stored
        # cluster by cluster (its thresholds come from the first rows); the f32 scorer -- on the hashed sample since round 6 -- must not
        g.set_option("force_exact", 1)
        idx_e, sc_e = _search(g, q, 1024)
        assert g.flags() == 0
        assert_oracle_parity(idx_e[pick], sc_e[pick], scores, top_i, top_s, K)
        g.set_option("force_exact", 0)
        # and the host entry point (verified loop, whatever it has to fall back to) gives the oracle's answer
        ih, sh, _ = g.search(q.cpu().numpy()[pick], K)
        assert_oracle_parity(ih, sh, scores, top_i, top_s, K)
    finally:
        g.close()
