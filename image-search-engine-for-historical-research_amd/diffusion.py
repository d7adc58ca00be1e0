"""Host-side mirror of the reference's truncated graph diffusion (src/utils/diffusion.py) and of the
small-database branch of QGE (src/utils/Reranking.py:212-264) for the HIP path.

    d = Diffusion(features[N,D], cache_dir)          # src/utils/diffusion.py:42-50
    offline = d.get_offline_results(n_trunc, kd)     # scipy CSR [N,N] float32, cached as offline.jbl
    ranks, scores = d.search_online(q[Q,D], 3, 2000) # src/utils/Reranking.py:238-253

The kNN graph, the mutual-kNN affinity, the normalised Laplacian, the N truncated CG solves and the
online combination + top-`trunc` selection all run on the GPU (csrc/diffusion.hip, csrc/dense.hip);
this module only keeps the reference's object surface and its joblib cache convention
(`@cache('offline.jbl')`, src/utils/diffusion.py:21-40,52; stale-cache hazard included: delete the
file when the database changes, like the reference's README says).
"""
import os
import time

import numpy as np

from . import evaluate
from ._lib import Gallery, NORM_NONE


class Diffusion:
    def __init__(self, features, cache_dir=None, device=0, group=None):
        features = np.asarray(features)
        self.group = group
        self.N = len(features)
        self.cache_dir = cache_dir
        # (for N >= 110000 the reference switches its kNN graph to an approximate IVFPQ index, src/utils/diffusion.py:47-49,
        # 57-60; QGE never diffuses at that size, src/utils/Reranking.py:212, and this build keeps the graph exact throughout)
        self.gallery = Gallery.from_host(features, norm_mode=NORM_NONE, device=device)
        self.n_trunc = None

    # `diffusion.knn.search(q, k)` is used by QGE (src/utils/Reranking.py:239-241)
    @property
    def knn(self):
        return self

    def search(self, queries, k):
        ids, sims, _ = self.gallery.search(np.asarray(queries, dtype=np.float32), int(k))
        return sims, ids

    def get_offline_results(self, n_trunc, kd=50):
        import scipy.sparse as sparse
        path = os.path.join(self.cache_dir, "offline.jbl") if self.cache_dir else None
        t0 = time.time()
        if path and os.path.exists(path):
            import joblib
            offline = joblib.load(path)
            print("Loading cache: {} costs {:.2f}s".format(path, time.time() - t0))
            csr = offline.tocsr()
            csr.sort_indices()
            # the rows carry exactly n_trunc stored entries each (explicit zeros kept)
            ids = csr.indices.reshape(self.N, -1).astype(np.int64)
            self.gallery.diffusion_set_offline(ids, csr.data.reshape(self.N, -1))
            self.n_trunc = ids.shape[1]
            return offline
        ids, vals = self._offline(int(n_trunc), int(kd))
        self.n_trunc = int(n_trunc)
        rows = np.repeat(np.arange(self.N), n_trunc)
        offline = sparse.csr_matrix((vals.reshape(-1), (rows, ids.reshape(-1))), shape=(self.N, self.N),
                                    dtype=np.float32)
        print("Obtaining cache: {} costs {:.2f}s".format(path, time.time() - t0))
        if path and self._rank() == 0:                      # one writer when several ranks share the cache directory
            import joblib
            os.makedirs(self.cache_dir, exist_ok=True)
            joblib.dump(offline, path)
        return offline

    def _rank(self):
        import torch.distributed as dist
        return dist.get_rank(self.group) if dist.is_available() and dist.is_initialized() else 0

    def _offline(self, n_trunc, kd):
        """One process: every node here.  Under torch.distributed (one process per GPU, every rank holding the same
        features -- the reference diffuses on N < 120 000 only): the N CG solves are independent
        (src/utils/diffusion.py:15-19,73-76 runs them as joblib tasks), so rank r solves the nodes of
        shard_bounds(N, world, r), the parts are all-gathered (N * n_trunc * 4 bytes in all) and every rank installs the
        complete result.  Bit-identical to the single-process result."""
        import torch.distributed as dist
        world = dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1
        if world == 1:
            return self.gallery.diffusion_offline(n_trunc, kd)
        import torch
        from .sharded import shard_bounds
        rank = dist.get_rank(self.group)
        lo, hi = shard_bounds(self.N, world, rank)
        ids, part = self.gallery.diffusion_offline_nodes(n_trunc, kd, lo, hi)
        per = -(-self.N // world)
        dev = "cuda" if dist.get_backend(self.group) == "nccl" else "cpu"
        mine = torch.zeros((per, n_trunc), dtype=torch.float32, device=dev)
        mine[:hi - lo] = torch.from_numpy(part).to(dev)
        out = torch.empty((world * per, n_trunc), dtype=torch.float32, device=dev)
        dist.all_gather_into_tensor(out, mine, group=self.group)
        vals = out[:self.N].cpu().numpy()                    # shard r occupies rows [r * per, r * per + its count)
        self.gallery.diffusion_set_offline(ids, vals)
        return ids, vals

    def search_online(self, queries, k_query=3, truncation_number=2000):
        """-> (ranks_dfs int64 [truncation_number, Q], scores float32 [Q, truncation_number])."""
        ranks, scores = self.gallery.diffusion_online(np.asarray(queries), k_query, 3, truncation_number)
        return ranks.T, scores

    def close(self):
        self.gallery.close()


def qge_small_hip(ranks, qvecs, vecs, dataset, gnd, AQE=True, K=None, cache_dir=None, device=0, quiet=False,
                  truncation_number=2000, k_gallery=200, k_query=3):
    """N < 120000 branch of QGE (src/utils/Reranking.py:212-264): alpha-QE with k=10, w=4, then diffusion
    with the expanded (AQE=True) or the original queries.  Returns a dict (the reference prints)."""
    from .reranking import feature_enhancement_hip
    vecs = np.asarray(vecs)
    n = vecs.shape[1]
    Kq = int(K) if K else min(n, 1000)
    qx, ranks_aqe = feature_enhancement_hip(10, ranks, vecs, 8.0 / 2, Kq, None, False, device)
    trunc = min(truncation_number, n - 1)
    kd = min(k_gallery, trunc)
    diffusion = Diffusion(np.ascontiguousarray(vecs.T), cache_dir, device)
    try:
        diffusion.get_offline_results(trunc, kd)
        q_search = qx.T if AQE else np.asarray(qvecs).T
        ranks_dfs, scores = diffusion.search_online(q_search, k_query, trunc)
    finally:
        diffusion.close()
    out = dict(qvecs_qe=qx, ranks_aqe=ranks_aqe, ranks_dfs=ranks_dfs, scores_dfs=scores)
    if gnd is not None:
        if not quiet:
            print("mAP after Enhancement (Random Walk)")
            out["map_dfs"] = evaluate.compute_map_and_print(dataset, ranks_dfs, gnd)
        else:
            out["map_dfs"] = evaluate.compute_map_revisited(ranks_dfs, gnd)
    return out
