"""Exact kNN wrapper with the interface of the reference's faiss wrapper (src/utils/knn.py:8-40):

    KNN(database[N,D], 'cosine').search(queries[Q,D], k) -> (sims float32 [Q,k], ids int64 [Q,k])

`cosine` is faiss.IndexFlatIP there: exact top-k raw inner products, descending; rows are used as
given (the callers L2-normalise beforehand).  Here the index is a device-resident gallery
(NORM_NONE) and the search runs through the HIP path; ties go to the lower id.
"""
import numpy as np

from ._lib import Gallery, NORM_NONE


class KNN:
    def __init__(self, database, method="cosine", device=0):
        if method != "cosine":
            raise NotImplementedError("only 'cosine' (IndexFlatIP) is on the reference's hot path")
        database = np.asarray(database)
        if database.dtype != np.float32:          # src/utils/knn.py:10-11
            database = database.astype(np.float32)
        self.N, self.D = database.shape
        self.gallery = Gallery.from_host(database, norm_mode=NORM_NONE, device=device)

    def search(self, queries, k):
        queries = np.asarray(queries)
        if queries.dtype != np.float32:           # src/utils/knn.py:28-29
            queries = queries.astype(np.float32)
        ids, sims, _ = self.gallery.search(queries, int(k))
        return sims, ids

    def close(self):
        self.gallery.close()
