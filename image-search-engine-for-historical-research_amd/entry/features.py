"""Feature store IO in the reference's on-disk format (src/utils/general.py:67-92):
outputs/features/<dataset>_path_feature.pkl = pickle {'path': [...], 'feature': ndarray [D,N]}."""
import os
import pickle

import numpy as np


def feature_file(dataset):
    return os.path.join("outputs", "features", dataset.replace("/", "_") + "_path_feature.pkl")


def save_path_feature(dataset, vecs, img_r_path):
    os.makedirs(os.path.join("outputs", "features"), exist_ok=True)
    with open(feature_file(dataset), "wb") as f:
        pickle.dump({"path": list(img_r_path), "feature": np.asarray(vecs)}, f)


def load_path_features(dataset):
    with open(feature_file(dataset), "rb") as f:
        pf = pickle.load(f)
    return pf["feature"], pf["path"]


def load_database(datasets, dim_vec=2048):
    """Concatenation as src/online.py:95-102 / src/offline.py:85-97 do it.  The reference starts from
    np.empty((2048, 0)) (float64), which silently promotes the float32 pickles; the dtype of the
    pickles is kept here (the HIP ingest accepts both and stores f32 rows either way)."""
    vecs, paths = None, []
    for ds in datasets:
        v, p = load_path_features(ds)
        vecs = v if vecs is None else np.concatenate([vecs, v], axis=1)
        paths += list(p)
    return vecs, paths


def load_torch_vecs(path):
    """The 1M-distractor descriptors the reference keeps as a torch tensor [D, N] float32
    (`torch.save(vecs, network + '_vecs_revisitop1m.pt')`, src/extract_1m.py:97-98; loaded and concatenated on the
    host at src/test_rOP1m.py:136-139).  Returned as a numpy view of the memory-mapped file where torch supports it
    (zip-format checkpoints), so the 8.2 GB never sit in host memory twice; pass it to Gallery.from_blocks."""
    import torch
    try:
        t = torch.load(path, map_location="cpu", mmap=True, weights_only=True)
    except (RuntimeError, TypeError, ValueError):
        t = torch.load(path, map_location="cpu", weights_only=True)
    if not isinstance(t, torch.Tensor) or t.dim() != 2:
        raise ValueError("%s does not hold a [D, N] tensor" % path)
    return t.numpy()
