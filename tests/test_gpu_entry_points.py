"""GPU: the three drivers that mirror the reference's entry points (src/offline.py, src/online.py, src/test_rOP1m.py) run
end to end on a synthetic feature store in the reference's on-disk format (outputs/features/<ds>_path_feature.pkl,
src/utils/general.py:67-92): offline prepares and persists the gallery, online answers from it (search + qge1), the test
driver prints the matching time and mAP and runs QGE."""
import os
import pickle

import numpy as np
import pytest

import oracle
from isehr_amd.synth import planted_dataset

pytestmark = pytest.mark.gpu


@pytest.fixture()
def feature_store(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    from isehr_amd import nnsearch
    from isehr_amd.entry.features import save_path_feature
    nnsearch.drop_cached_galleries()
    vecs, qvecs, gnd = planted_dataset(77, 9000, 96, 20)            # [D, N], [D, Q], revisited-style gnd
    half = vecs.shape[1] // 2
    save_path_feature("dsA", vecs[:, :half], ["a/%05d.jpg" % i for i in range(half)])
    save_path_feature("dsB", vecs[:, half:].astype(np.float64), ["b/%05d.jpg" % i for i in range(vecs.shape[1] - half)])
    save_path_feature("roxford5k_db", vecs, ["db/%05d.jpg" % i for i in range(vecs.shape[1])])
    save_path_feature("roxford5k_query", qvecs, ["q/%03d.jpg" % i for i in range(qvecs.shape[1])])
    os.makedirs("data/test/roxford5k", exist_ok=True)
    with open("data/test/roxford5k/gnd_roxford5k.pkl", "wb") as f:
        pickle.dump({"gnd": gnd}, f)
    yield vecs, qvecs, gnd
    nnsearch.drop_cached_galleries()


def test_offline_then_online(feature_store, capsys):
    from isehr_amd import nnsearch
    from isehr_amd.entry import offline, online
    from isehr_amd.entry.features import load_database
    vecs, qvecs, _ = feature_store
    assert offline.main(["--datasets", "dsA,dsB", "--matching_method", "HIP", "--ifgenerate"]) == 0
    assert "prepared on GPU" in capsys.readouterr().out
    assert any(f.startswith("mi355_gallery") for f in os.listdir("outputs/database"))     # persisted like the ANN indexes
    nnsearch.drop_cached_galleries()                                  # a new process: the gallery comes from the file
    db, paths = load_database(["dsA", "dsB"])
    s = online.Searcher(db, paths, 30)
    got = s.query(qvecs[:, 3])                                        # one descriptor, like an uploaded image
    assert len(got) == 1 and len(got[0]) == 30
    # the same through the oracle: matching_l2 then qge1 (alpha-QE k = 3, w = 4) on the concatenated database
    base = oracle.matching_l2(30, db.T.astype(np.float32), qvecs[:, 3:4].T.astype(np.float32)).T
    ref = oracle.qge1(base, qvecs[:, 3:4], db.astype(np.float32), 30)[:30, 0]
    want = [paths[i] for i in ref]
    assert got[0][:10] == want[:10]
    assert len(set(got[0]) & set(want)) >= 28                         # float32 near-ties may swap the last places
    np.save("q.npy", qvecs[:, :2])
    assert online.main(["--datasets", "dsA,dsB", "--query-npy", "q.npy"]) == 0
    assert ".jpg" in capsys.readouterr().out


@pytest.mark.parametrize("mode", ["100", "mAP"])
def test_test_rop1m_driver(feature_store, capsys, mode):
    from isehr_amd import evaluate
    from isehr_amd.entry import test_rOP1m
    vecs, qvecs, gnd = feature_store
    assert test_rOP1m.main(["--datasets", "roxford5k", "--ifextracted", "--mode", mode]) == 0
    out = capsys.readouterr().out
    assert "average matching time" in out and "mAP E:" in out
    # the printed mAP is the oracle's for the same ranking depth
    K = vecs.shape[1] if mode == "mAP" else 100
    ranks = oracle.matching_l2(K, vecs.T, qvecs.T).T
    e, m, h = evaluate.compute_map_revisited(ranks, gnd)
    line = [ln for ln in out.splitlines() if "mAP E:" in ln][0]
    for v in (e, m, h):
        assert str(np.around(v * 100, decimals=2)) in line, (line, e, m, h)


@pytest.mark.parametrize("rerank", ["qge", "aqe", "dba", "kr"])
def test_test_reranking_driver(tmp_path, monkeypatch, capsys, rerank):
    """src/test_reranking.py's flow on numpy feature files (outputs/<dataset>_vecs.npy, [D, N]): matcher at K = 4000 ->
    mAP -> one re-ranker.  The printed mAPs equal the oracle's for the same steps; QGE below 120 000 images writes and then
    re-uses the diffusion cache the reference keeps under its cache_dir."""
    from isehr_amd import evaluate
    from isehr_amd.entry import test_reranking
    monkeypatch.chdir(tmp_path)
    vecs, qvecs, gnd = planted_dataset(78, 6000, 64, 12)
    os.makedirs("outputs", exist_ok=True)
    os.makedirs("data/test/roxford5k", exist_ok=True)
    np.save("outputs/roxford5k_vecs.npy", vecs)
    np.save("outputs/roxford5k_qvecs.npy", qvecs)
    with open("data/test/roxford5k/gnd_roxford5k.pkl", "wb") as f:
        pickle.dump({"gnd": gnd}, f)
    assert test_reranking.main(["--datasets", "roxford5k", "--rerank", rerank]) == 0
    out = capsys.readouterr().out
    assert "matching time per query" in out
    lines = [ln for ln in out.splitlines() if "mAP E:" in ln]
    base = oracle.matching_l2(4000, vecs.T, qvecs.T).T
    e, m, h = evaluate.compute_map_revisited(base, gnd)
    for v in (e, m, h):
        assert str(np.around(v * 100, decimals=2)) in lines[0], (lines[0], e, m, h)
    assert len(lines) == 2                                        # initial ranking + the re-ranker's
    if rerank == "qge":
        assert os.path.exists("diffusion/tmp/roxford5k/offline.jbl") and "Obtaining cache" in out
        # the oracle's small-database QGE (alpha-QE k = 10, w = 4, then diffusion from the expanded queries)
        _, _, ranks_dfs = oracle.qge_small(base, qvecs, vecs, AQE=True, truncation_number=2000, k_gallery=200, k_query=3)
        want = evaluate.compute_map_revisited(ranks_dfs, gnd)
        got = [float(x) for x in lines[1].replace(",", " ").split() if x.replace(".", "", 1).isdigit()]
        assert np.abs(np.array(got[-3:]) - np.around(np.array(want) * 100, 2)).max() <= 0.05, (lines[1], want)
        assert test_reranking.main(["--datasets", "roxford5k", "--rerank", "qge"]) == 0    # second run: cache hit
        assert "Loading cache" in capsys.readouterr().out
    else:
        ref = {"aqe": oracle.average_query_expansion, "dba": oracle.database_augmentation}.get(rerank)
        if ref is not None:
            want = evaluate.compute_map_revisited(ref(qvecs, vecs, 100), gnd)
            got = [float(x) for x in lines[1].replace(",", " ").split() if x.replace(".", "", 1).isdigit()]
            assert np.abs(np.array(got[-3:]) - np.around(np.array(want) * 100, 2)).max() <= 0.05, (lines[1], want)
