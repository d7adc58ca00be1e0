"""Counterpart of the retrieval part of src/online.py:108-152 (the Flask route minus the CNN):
`Searcher.query(qvec)` = dispatch on --matching_method, then qge1 re-ranking, then the top-K paths.
A lock serialises concurrent callers (Flask's threaded server)."""
import argparse
import threading

import numpy as np

from ..nnsearch import matching_HIP
from ..reranking import qge1_hip
from .features import load_database

parser = argparse.ArgumentParser(description="Online retrieval (HIP exhaustive matcher)")
parser.add_argument("--datasets", "-d", default="database")
parser.add_argument("--matching_method", "-mm", default="HIP")
parser.add_argument("--K-nearest-neighbour", "-K", dest="K_nearest_neighbour", type=int, default=30)
parser.add_argument("--ifgenerate", "-gen", dest="ifgenerate", action="store_true")
parser.add_argument("--gpu-id", "-g", default="0")
parser.add_argument("--query-npy", help="file with query descriptors [D] or [D,Q] (stands in for the uploaded image)")


class Searcher:
    def __init__(self, vecs, img_paths, K, matching_method="HIP", ifgenerate=False, device=0):
        self.vecs, self.img_paths, self.K = vecs, img_paths, K
        self.method, self.ifgenerate, self.device = matching_method, ifgenerate, device
        self._lock = threading.Lock()

    def _device_chain(self):
        """The two prepared galleries of the chain -- L2-normalised rows for the search (matching_L2's ranking), the rows as
        stored for qge1's expansion and re-search (src/online.py:132,148: `vecs` itself) -- wrapped for device tensors."""
        if getattr(self, "_chain", None) is None:
            from ..nnsearch import get_gallery, NORM_L2, NORM_NONE
            from ..sharded import ShardedGallery
            g1 = get_gallery(self.vecs.T, "database", self.ifgenerate, NORM_L2, self.device)
            g2 = get_gallery(self.vecs.T, "database_raw", self.ifgenerate, NORM_NONE, self.device)
            self.ifgenerate = False
            self._chain = (ShardedGallery(g1), ShardedGallery(g2))
        return self._chain

    def query_device(self, desc, return_indices=False):
        """The online chain WITHOUT a host round trip (src/online.py:121-152 copies the descriptor to the CPU,
        searches there and re-ranks there): `desc` is the extractor tail's output on the device (float32 cuda tensor [D] or
        [Q, D]: isehr_amd.extractor.DescriptorTail / extract_ms_device -> mi_desc_tail_device), the search (K nearest by cosine),
        the qge1 expansion from the top-3 rows (k = 3, w = 4, float64 sum, eps-normalised) and the re-search of the expanded
        query all run on the device through the asynchronous entry points; ONE device-to-host copy of Q x K indices ends it."""
        if desc.dim() == 1:
            desc = desc[None, :]
        desc = desc.contiguous().float()
        with self._lock:
            sg1, sg2 = self._device_chain()
            idx, _ = sg1.search(desc, self.K)
            idx2, _, _ = sg2.aqe_search(idx.t(), 3, 4.0, self.K)
            out = idx2.cpu().numpy()                                  # the one D2H copy (synchronises)
        if return_indices:
            return out
        return [[self.img_paths[i] for i in row] for row in out]

    def query(self, qvec):
        qvec = np.asarray(qvec)
        if qvec.ndim == 1:
            qvec = qvec[:, None]                                  # src/online.py:123
        with self._lock:
            if self.method == "HIP":
                match_idx, _ = matching_HIP(self.K, self.vecs.T, qvec.T, dataset="database",
                                            ifgenerate=self.ifgenerate, device=self.device)
            else:
                raise ValueError("Invalid method")               # the reference prints and then fails on a NameError
            self.ifgenerate = False
            ranks = match_idx.T
            ranks2 = qge1_hip(ranks, qvec, self.vecs, self.K, dataset="database_raw", device=self.device)
        idx2 = ranks2.T
        return [[self.img_paths[i] for i in row[:self.K]] for row in idx2]


def main(argv=None):
    args = parser.parse_args(argv)
    vecs, paths = load_database(args.datasets.split(","))
    s = Searcher(vecs, paths, args.K_nearest_neighbour, args.matching_method, args.ifgenerate, int(args.gpu_id))
    if args.query_npy:
        for row in s.query(np.load(args.query_npy)):
            print("\n".join(map(str, row)))
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
