"""Counterpart of src/test_reranking.py:38-117, the reference's re-ranking test bench: descriptors come from the numpy files
`outputs/<dataset>_vecs.npy` / `outputs/<dataset>_qvecs.npy` ([D, N] / [D, Q], :44-61), the initial ranking is the
exhaustive matcher at K = 4000 (:66-71), its mAP is printed (:80), then one re-ranker runs: QGE (:98, the live line --
alpha-QE + graph diffusion below 120 000 images, alpha-QE alone above) or one of the alternatives the reference keeps
commented out next to it (average query expansion :101, database augmentation :104, k-reciprocal :107).  The local-feature
verifiers listed there (SAHA, LoFTR, RANSAC-SIFT: :110-116) are outside this build.

  python -m isehr_amd.entry.test_reranking --datasets roxford5k --rerank qge
  python -m isehr_amd.entry.test_reranking --datasets roxford5k --rerank kr --gnd-dir data/test
"""
import argparse
import os
import pickle

import numpy as np

from .. import evaluate
from ..nnsearch import matching_HIP
from ..reranking import QGE_hip, average_query_expansion_hip, database_augmentation_hip, kr_reranking_hip

parser = argparse.ArgumentParser(description="Re-ranking test (HIP exhaustive matcher + HIP re-rankers)")
parser.add_argument("--matching_method", "-mm", default="HIP",
                    help="kept for CLI parity (the reference parses it and calls matching_L2 regardless, src/test_reranking.py:32,70)")
parser.add_argument("--datasets", "-d", default="roxford5k")
parser.add_argument("--outputs-dir", default="outputs", help="directory of <dataset>_vecs.npy / <dataset>_qvecs.npy")
parser.add_argument("--gnd-dir", default="data/test", help="<gnd-dir>/<dataset>/gnd_<dataset>.pkl")
parser.add_argument("-K", type=int, default=4000, help="depth of the initial ranking (src/test_reranking.py:66)")
parser.add_argument("--rerank", default="qge", choices=["qge", "aqe", "dba", "kr", "none"])
parser.add_argument("--no-AQE", action="store_true", help="QGE: diffuse from the original queries (AQE = False, :94)")
parser.add_argument("--cache-dir", default="", help="QGE below 120 000 images: directory of the diffusion cache "
                                                    "offline.jbl (default diffusion/tmp/<dataset>, like :91)")
parser.add_argument("--gpu-id", "-g", default="0")


def load_npy_features(outputs_dir, dataset):
    """[D, N] and [D, Q] arrays as np.save wrote them; memory-mapped: the matcher copies column blocks to the GPU itself."""
    vecs = np.load(os.path.join(outputs_dir, dataset + "_vecs.npy"), mmap_mode="r")
    qvecs = np.load(os.path.join(outputs_dir, dataset + "_qvecs.npy"), mmap_mode="r")
    if vecs.ndim != 2 or qvecs.ndim != 2 or vecs.shape[0] != qvecs.shape[0]:
        raise ValueError("expected [D, N] and [D, Q] arrays, got %s and %s" % (vecs.shape, qvecs.shape))
    return vecs, qvecs


def run_dataset(dataset, vecs, qvecs, gnd, K=4000, rerank="qge", AQE=True, cache_dir=None, device=0):
    n = vecs.shape[1]
    K = min(int(K), n)
    match_idx, time_per_query = matching_HIP(K, vecs.T, qvecs.T, device=device)
    print("matching time per query: ", time_per_query)
    ranks = match_idx.T
    print("------------------------------------------------------")
    print("mAP:")
    res = {"ranks": ranks, "map": evaluate.compute_map_and_print(dataset, ranks, gnd)}
    print("------------------------------------------------------")
    if rerank == "qge":
        res["qge"] = QGE_hip(ranks, qvecs, vecs, dataset, gnd, cache_dir, None, AQE, device=device)
    elif rerank in ("aqe", "dba"):
        fn = average_query_expansion_hip if rerank == "aqe" else database_augmentation_hip
        res["ranks_" + rerank] = fn(np.asarray(qvecs), np.asarray(vecs), min(100, n), device=device)   # K = 100, :93
        print("mAP after " + ("average query expansion" if rerank == "aqe" else "database augmentation"))
        res["map_" + rerank] = evaluate.compute_map_and_print(dataset, res["ranks_" + rerank], gnd)
    elif rerank == "kr":
        indices = kr_reranking_hip(np.asarray(qvecs), np.asarray(vecs), device=device)                # [Q, N], :623
        res["ranks_kr"] = np.ascontiguousarray(indices.T)
        print("mAP after k-reciprocal re-ranking")
        res["map_kr"] = evaluate.compute_map_and_print(dataset, res["ranks_kr"], gnd)
    return res


def main(argv=None):
    args = parser.parse_args(argv)
    dev = int(args.gpu_id)
    for dataset in args.datasets.split(","):
        vecs, qvecs = load_npy_features(args.outputs_dir, dataset)
        with open(os.path.join(args.gnd_dir, dataset, "gnd_%s.pkl" % dataset), "rb") as f:
            gnd = pickle.load(f)["gnd"]
        cache_dir = args.cache_dir or os.path.join("diffusion", "tmp", dataset)
        run_dataset(dataset, vecs, qvecs, gnd, args.K, args.rerank, not args.no_AQE, cache_dir, dev)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
