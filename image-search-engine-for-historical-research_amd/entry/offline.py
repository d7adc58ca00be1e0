"""python -m isehr_amd.entry.offline --datasets a,b --matching_method HIP [--ifgenerate]
Counterpart of src/offline.py:99-118 for the HIP method: the "offline" step of an exhaustive GPU
matcher is preparing the device gallery (normalised f32 rows + tile-blocked bf16 image) and
persisting it under outputs/database/, exactly where the ANN methods keep their indexes."""
import argparse

import numpy as np

from ..nnsearch import matching_HIP
from .features import load_database

parser = argparse.ArgumentParser(description="Offline preparation (HIP exhaustive matcher)")
parser.add_argument("--datasets", "-d", default="database")
parser.add_argument("--matching_method", "-mm", default="HIP", help="only 'HIP' is implemented by this build")
parser.add_argument("--K-nearest-neighbour", "-K", dest="K_nearest_neighbour", type=int, default=30)
parser.add_argument("--ifgenerate", "-gen", dest="ifgenerate", action="store_true")
parser.add_argument("--gpu-id", "-g", default="0")


def main(argv=None):
    args = parser.parse_args(argv)
    vecs, _ = load_database(args.datasets.split(","))
    qvec = np.ones((vecs.shape[0], 1), dtype=vecs.dtype)        # dummy query, src/offline.py:101
    if args.matching_method == "HIP":
        match_idx, t = matching_HIP(args.K_nearest_neighbour, vecs.T, qvec.T, dataset="database",
                                    ifgenerate=args.ifgenerate, device=int(args.gpu_id))
        print(">> gallery of %d x %d prepared on GPU %s (%.3f s)" % (vecs.shape[1], vecs.shape[0], args.gpu_id, t))
    else:
        print("Invalid method")                                  # src/offline.py:117-118
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
