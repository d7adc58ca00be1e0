"""Counterpart of src/test_custom.py:26-43: a directory-labelled collection (`custom/query`, `custom/database`) read from the
feature store (`outputs/features/custom_<part>_path_feature_2.pkl`, :18-23), ranked IN FULL by the exhaustive matcher
(K = number of database images, :30-32), scored by mAP_custom (:33) and written as {query path: [database paths in
ranked order]} to outputs/ranks/custom_ranking_result.pkl (:36-43).  The matplotlib panel behind it (:45-84) is a viewer,
not part of the path.

  python -m isehr_amd.entry.test_custom [--suffix _path_feature_2.pkl]
"""
import argparse
import os
import pickle

import numpy as np

from .. import evaluate
from ..nnsearch import matching_HIP

parser = argparse.ArgumentParser(description="Custom collection: full ranking + mAP (HIP exhaustive matcher)")
parser.add_argument("--query", default="custom/query")
parser.add_argument("--database", default="custom/database")
parser.add_argument("--features-dir", default="outputs/features")
parser.add_argument("--suffix", default="_path_feature_2.pkl", help="src/test_custom.py:18 reads <dataset>_path_feature_2.pkl")
parser.add_argument("--ranks-file", default="outputs/ranks/custom_ranking_result.pkl")
parser.add_argument("--gpu-id", "-g", default="0")


def load_path_features(features_dir, dataset, suffix):
    with open(os.path.join(features_dir, dataset.replace("/", "_") + suffix), "rb") as f:
        pf = pickle.load(f)
    return pf["feature"], pf["path"]


def run(custom_q, relpaths_q, custom_d, relpaths_d, device=0):
    K = custom_d.shape[1]                                                      # the whole database, :30-31
    match_idx, time_per_query = matching_HIP(K, np.asarray(custom_d).T, np.asarray(custom_q).T, device=device)
    mAP = evaluate.map_custom(K, match_idx, relpaths_q, relpaths_d)
    print("mean average precision: ", mAP)
    rank_res = {relpaths_q[i]: [relpaths_d[j] for j in match_idx[i, :]] for i in range(len(relpaths_q))}
    return mAP, match_idx, rank_res, time_per_query


def main(argv=None):
    args = parser.parse_args(argv)
    custom_q, relpaths_q = load_path_features(args.features_dir, args.query, args.suffix)
    custom_d, relpaths_d = load_path_features(args.features_dir, args.database, args.suffix)
    _, _, rank_res, _ = run(custom_q, relpaths_q, custom_d, relpaths_d, int(args.gpu_id))
    os.makedirs(os.path.dirname(args.ranks_file) or ".", exist_ok=True)
    with open(args.ranks_file, "wb") as f:
        pickle.dump(rank_res, f)
    print("end")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
