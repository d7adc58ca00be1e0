"""Counterpart of src/test_rOP1m.py:99-168 for pre-extracted (or synthetic) descriptors:
match (HIP), print the average matching time and mAP, then QGE re-ranking.

  python -m isehr_amd.entry.test_rOP1m --datasets roxford5k --ifextracted --mode 100
  python -m isehr_amd.entry.test_rOP1m --synthetic 20000 --mode 100          (no files needed)
"""
import argparse
import pickle

import numpy as np

from .. import evaluate
from ..nnsearch import matching_HIP
from ..reranking import QGE_hip
from ..synth import planted_dataset
from .features import load_path_features

parser = argparse.ArgumentParser(description="Retrieval test (HIP exhaustive matcher)")
parser.add_argument("--datasets", "-d", default="roxford5k,rparis6k")
parser.add_argument("--ifextracted", action="store_true", help="kept for CLI parity; features are always read")
parser.add_argument("--include1m", action="store_true", help="append outputs/features/revisitop1m_path_feature.pkl")
parser.add_argument("--distractors", default="", help="torch tensor file [D, N] of the 1M distractors (src/extract_1m.py:98)")
parser.add_argument("--mode", default="100", help="'mAP' (rank as deep as the HIP path allows) or top-K as int")
parser.add_argument("--gnd-dir", default="data/test")
parser.add_argument("--synthetic", type=int, default=0, help="use a planted synthetic dataset with this many rows")
parser.add_argument("--dim", type=int, default=2048)
parser.add_argument("--gpu-id", "-g", default="0", help="one GPU id, or a comma-separated list: the matcher then splits the "
                                                        "gallery rows over those GPUs inside this process (top-K modes)")


def run_dataset(dataset, vecs, qvecs, gnd, mode, device=0, devices=None):
    from ..nnsearch import ColumnBlocks
    blocks = isinstance(vecs, ColumnBlocks)
    n = vecs.shape[0] if blocks else vecs.shape[1]
    K = n if mode == "mAP" else int(mode)                 # 'mAP' ranks the whole database (src/test_rOP1m.py:144-149)
    use_list = bool(devices) and K <= 2048 and not blocks
    if devices and not use_list:
        print(">> {}: --gpu-id list ignored ({}): the rows are searched on GPU {} alone".format(
            dataset, "K = %d > 2048 takes the full-length ranking path" % K if K > 2048 else "gallery given as column blocks",
            device))
    match_idx, time_per_query = matching_HIP(K, vecs if blocks else vecs.T, qvecs.T, device=device,
                                             devices=devices if use_list else None)
    ranks = match_idx.T
    print(">> {}: average matching time: {}".format(dataset, time_per_query))
    res = {"map": evaluate.compute_map_and_print(dataset, ranks, gnd)}
    res["qge"] = QGE_hip(ranks, qvecs, vecs, dataset, gnd, K=min(K, 2048), device=device)
    return res


def main(argv=None):
    args = parser.parse_args(argv)
    ids = [int(v) for v in str(args.gpu_id).split(",")]
    dev, devices = ids[0], (ids if len(ids) > 1 else None)
    if args.synthetic:
        vecs, qvecs, gnd = planted_dataset(1234, args.synthetic, args.dim, 70)
        run_dataset("roxford5k-synthetic", vecs, qvecs, gnd, args.mode, dev, devices)
        return 0
    for dataset in args.datasets.split(","):
        vecs, _ = load_path_features(dataset + "_db")
        qvecs, _ = load_path_features(dataset + "_query")
        if args.include1m:
            # src/test_rOP1m.py:136-139 loads `<network>_vecs_revisitop1m.pt` (torch tensor [D, 1001001]) and concatenates
            # on the host; here the file is memory-mapped and the gallery is built block by block
            from ..nnsearch import ColumnBlocks
            from .features import load_torch_vecs
            v1m = load_torch_vecs(args.distractors) if args.distractors else load_path_features("revisitop1m")[0]
            vecs = ColumnBlocks([vecs, v1m])
        with open("{}/{}/gnd_{}.pkl".format(args.gnd_dir, dataset, dataset), "rb") as f:
            gnd = pickle.load(f)["gnd"]
        run_dataset(dataset, vecs, qvecs, gnd, args.mode, dev, devices)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
