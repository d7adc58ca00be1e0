"""Counterpart of src/main_retrieve.py:114-180 (and of the evaluation loop of src/main_train.py:680-716): database and query
descriptors of every test dataset -> feature store (`save_path_feature`, :159-160) -> the inner-product ranker
`ranks = argsort(-(vecs.T @ qvecs), axis=0)` IN FULL (:175-176) -> mAP E / M / H and mP@k through the first-generation
evaluator (:177) -> optionally the same again on whitened descriptors (`whitenapply` with a learned {m, P},
src/main_train.py:709-716).

Descriptors come from numpy files `<outputs-dir>/<dataset>_vecs.npy` / `_qvecs.npy` ([D, N] / [D, Q], the files
src/test_reranking.py:44-61 reads), or -- `--extract` -- from image tensors `<outputs-dir>/<dataset>_images.pt` /
`_qimages.pt` ([N, 3, H, W] float32, already normalised) through the ResNet101-SOA trunk + the HIP descriptor tail, rows
going straight into a device gallery (extract_to_gallery: no CPU round trip).  No checkpoint ships, so `--extract` runs the
random-initialised architecture (`--network-path` loads a state dict where one exists).

  python -m isehr_amd.entry.main_retrieve --datasets roxford5k --outputs-dir outputs [--whitening Lw.pkl]
"""
import argparse
import os
import pickle
import time

import numpy as np

from .. import evaluate
from ..nnsearch import ip_rank_hip
from ..whiten import whitenapply_hip
from .features import save_path_feature

parser = argparse.ArgumentParser(description="Retrieval evaluation: inner-product ranker on the GPU")
parser.add_argument("--datasets", "-d", default="roxford5k,rparis6k")
parser.add_argument("--outputs-dir", default="outputs")
parser.add_argument("--gnd-dir", default="data/test", help="<gnd-dir>/<dataset>/gnd_<dataset>.pkl")
parser.add_argument("--whitening", "-w", default="", help="pickle {'m': [D,1], 'P': [D,D]} of a learned whitening "
                                                           "(src/main_train.py:670): evaluate the whitened descriptors too")
parser.add_argument("--extract", action="store_true", help="descriptors from <dataset>_images.pt / _qimages.pt")
parser.add_argument("--network-path", default="", help="--extract: torch state dict of the trunk (default: random init)")
parser.add_argument("--multiscale", default="1", help="--extract: comma-separated scales, e.g. '1,1.41421356,0.70710678' (:96)")
parser.add_argument("--blocks", default="3,4,23,3", help="--extract: bottleneck blocks per stage (ResNet101)")
parser.add_argument("--width", type=int, default=64, help="--extract: base width (descriptor dimension = 32 x width)")
parser.add_argument("--batch", type=int, default=8)
parser.add_argument("--no-save", action="store_true", help="do not write the feature store")
parser.add_argument("--gpu-id", "-g", default="0")


def extract_descriptors(trunk, images, ms=(1.0,), batch=8, device=0):
    """[N, 3, H, W] float32 tensor -> descriptors [D, N] float32 (numpy), extracted on the GPU and read back from the
    device gallery's stored rows (NORM_NONE keeps them as the tail produced them)."""
    import torch
    from .. import _lib
    from ..extractor import DescriptorTail, extract_to_gallery
    dev = torch.device("cuda", device)
    gal = _lib.Gallery.empty(images.shape[0], trunk.outputdim, norm_mode=_lib.NORM_NONE, device=device)
    try:
        extract_to_gallery(trunk, DescriptorTail(), (images[i:i + batch].to(dev) for i in range(0, images.shape[0], batch)),
                           gal, ms=ms)
        return np.ascontiguousarray(gal.get_rows(0, gal.n).T)
    finally:
        gal.close()


def evaluate_dataset(dataset, vecs, qvecs, gnd, Lw=None, device=0, kappas=(1, 5, 10)):
    res = {}
    ranks = ip_rank_hip(vecs, qvecs, device=device)                              # [N, Q], :175-176
    res["ranks"] = ranks
    res["map"] = evaluate.compute_map_and_print(dataset, ranks, gnd, kappas=list(kappas))
    if Lw is not None:
        vecs_lw = whitenapply_hip(vecs, Lw["m"], Lw["P"], device=device)       # src/main_train.py:711-712
        qvecs_lw = whitenapply_hip(qvecs, Lw["m"], Lw["P"], device=device)
        res["ranks_lw"] = ip_rank_hip(vecs_lw, qvecs_lw, device=device)
        res["map_lw"] = evaluate.compute_map_and_print(dataset + " + whiten", res["ranks_lw"], gnd, kappas=list(kappas))
    return res


def main(argv=None):
    args = parser.parse_args(argv)
    dev = int(args.gpu_id)
    Lw = None
    if args.whitening:
        with open(args.whitening, "rb") as f:
            Lw = pickle.load(f)
    trunk = None
    if args.extract:
        import torch
        from ..extractor import ResNet101SOA
        trunk = ResNet101SOA(blocks=tuple(int(v) for v in args.blocks.split(",")), width=args.width)
        trunk = trunk.to(torch.device("cuda", dev)).eval()
        if args.network_path:
            trunk.load_state_dict(torch.load(args.network_path, map_location="cpu", weights_only=True), strict=False)
    ms = tuple(float(v) for v in args.multiscale.split(","))
    for dataset in args.datasets.split(","):
        start = time.time()
        if args.extract:
            print(">> {}: Extracting...".format(dataset))
            load = lambda name: torch.load(os.path.join(args.outputs_dir, dataset + name), weights_only=True)  # noqa: E731
            vecs = extract_descriptors(trunk, load("_images.pt"), ms, args.batch, dev)
            qvecs = extract_descriptors(trunk, load("_qimages.pt"), ms, args.batch, dev)
        else:
            vecs = np.load(os.path.join(args.outputs_dir, dataset + "_vecs.npy"), mmap_mode="r")
            qvecs = np.load(os.path.join(args.outputs_dir, dataset + "_qvecs.npy"), mmap_mode="r")
        with open(os.path.join(args.gnd_dir, dataset, "gnd_%s.pkl" % dataset), "rb") as f:
            cfg = pickle.load(f)
        print(">> {}: Evaluating...".format(dataset))
        if not args.no_save:
            n, nq = vecs.shape[1], qvecs.shape[1]
            save_path_feature(dataset + "_database", vecs, cfg.get("imlist", ["%d" % i for i in range(n)]))
            save_path_feature(dataset + "_query", qvecs, cfg.get("qimlist", ["%d" % i for i in range(nq)]))
        evaluate_dataset(dataset, vecs, qvecs, cfg["gnd"], Lw, dev)
        print("")
        print(">> {}: elapsed time: {:.1f}s".format(dataset, time.time() - start))
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
