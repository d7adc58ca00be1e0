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
    nnsearch.wait_for_saves()                                         # (written behind the call that built it)
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
    s.close()
    # the same query through the device chain (descriptor on the device -> search -> qge1 -> one D2H of K indices)
    import torch
    got_dev = s.query_device(torch.from_numpy(np.ascontiguousarray(qvecs[:, 3])).cuda())
    assert got_dev == got


def test_online_device_chain_vs_reference_qge1_golden(golden_dir, tmp_path, monkeypatch):
    """`Searcher.query_device` -- extractor-tail descriptor on the device -> mi_knn_search_device (K) -> qge1
    expansion (k = 3, w = 4) -> re-search, one D2H of K indices -- against the reference's own feature_enhancement
    (tests/golden/qge.npz: `base` ranks, expanded queries, re-ranked lists; src/utils/Reranking.py:195-208, 287-306) and
    against the host chain `Searcher.query`."""
    import torch
    from isehr_amd import nnsearch
    from isehr_amd.entry import online
    from isehr_amd.synth import synth_rows
    monkeypatch.chdir(tmp_path)
    z = np.load(os.path.join(golden_dir, "qge.npz"))
    seed, n, d, nq = (int(v) for v in z["meta"])
    vecs = np.ascontiguousarray(synth_rows(seed, 0, n, d).T)          # tests/test_gpu_rerank_and_shards.py::_setup
    qv = np.ascontiguousarray(synth_rows(seed + 1000, 0, nq, d).T)
    vecs = vecs / np.linalg.norm(vecs, axis=0, keepdims=True)
    K = 200
    s = online.Searcher(vecs, list(range(n)), K)
    try:
        got = s.query_device(torch.from_numpy(np.ascontiguousarray(qv.T)).cuda(), return_indices=True)      # [Q, K]
        host = np.array(s.query(qv))                                  # the host chain, same galleries
        assert np.array_equal(got, host)
        # the search stage of the chain equals the golden's `base` ranks (top-3 feed the expansion), the result its re-ranking
        base = z["base"]
        first, _ = nnsearch.matching_HIP(K, vecs.T, qv.T)
        assert np.array_equal(first.T[:3], base[:3])
        ref_qx, ref_ranks = z["qx3"], z["ranks3_top"]
        s64 = (vecs.astype(np.float64).T @ ref_qx).T
        assert oracle.check_topk_parity(got, s64, K, 1e-6) == []
        assert (got.T == ref_ranks).mean() > 0.99
        # the galleries of a live chain are not dropped underneath it (the library refuses; they stay cached) ...
        nnsearch.drop_cached_galleries()
        assert np.array_equal(s.query_device(torch.from_numpy(np.ascontiguousarray(qv.T)).cuda(), return_indices=True), got)
    finally:
        s.close()                                                     # ... and go once the chain is closed
        nnsearch.drop_cached_galleries()
    assert not nnsearch._cache


@pytest.mark.parametrize("native", [True, False])
def test_online_concurrent_callers_are_coalesced_and_get_the_sequential_answers(tmp_path, monkeypatch, native):
    """src/online.py:163 runs Flask's threaded server: request threads call the route concurrently on module-level globals.
    `Searcher.query_device` hands concurrent descriptors to one worker that answers them in ONE search -> qge1 -> re-search
    chain.  64 threads x 20 queries: every answer equals the uncoalesced call's, and the requests really
    were batched.  native: the worker thread of the library (mi_online_*, the callers block outside the interpreter lock);
    otherwise the Python worker of entry/online.py."""
    import threading
    import torch
    from isehr_amd import nnsearch
    from isehr_amd.entry import online
    from isehr_amd.synth import synth_rows
    monkeypatch.chdir(tmp_path)
    n, d, K, nthr, per = 30000, 256, 30, 64, 20
    vecs = np.ascontiguousarray(synth_rows(91, 0, n, d).T)
    vecs = vecs / np.linalg.norm(vecs, axis=0, keepdims=True)
    qd = torch.from_numpy(synth_rows(92, 0, nthr * per, d)).cuda()
    plain = online.Searcher(vecs, list(range(n)), K, coalesce=False)
    srv = online.Searcher(vecs, list(range(n)), K, coalesce=True, native=native)
    try:
        want = plain.query_device(qd[:128], return_indices=True)
        want = np.concatenate([want] + [plain.query_device(qd[i:i + 128], return_indices=True)
                                        for i in range(128, nthr * per, 128)])
        # one sequential caller is not delayed and not batched
        assert np.array_equal(srv.query_device(qd[5], return_indices=True)[0], want[5]) and srv.chain_stats == (1, 1)
        assert (srv._native_chain is not None) == native
        got = np.full((nthr * per, K), -1, dtype=np.int64)
        errs = []

        def client(t):
            try:
                for i in range(per):
                    j = t * per + i
                    got[j] = srv.query_device(qd[j], return_indices=True)[0]
            except Exception as e:                                    # noqa: BLE001
                errs.append(e)
        ths = [threading.Thread(target=client, args=(t,)) for t in range(nthr)]
        for th in ths:
            th.start()
        for th in ths:
            th.join(timeout=120)
        assert not errs, errs[:1]
        assert np.array_equal(got, want)
        chains, answered = srv.chain_stats
        assert answered == nthr * per + 1 and chains < nthr * per / 4, (chains, answered)
        # a request wider than the batch limit is answered directly, a [Q, D] request keeps its rows together
        assert np.array_equal(srv.query_device(qd[:200], return_indices=True), want[:200])
        assert np.array_equal(srv.query_device(qd[300:307], return_indices=True), want[300:307])
        # descriptors still being produced on a stream of the caller's: the chain waits for them on the device
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            late = torch.zeros((3, d), device="cuda")
            torch.cuda._sleep(20_000_000)                             # ~10 ms of device time in front of the producer
            late.copy_(qd[40:43], non_blocking=True)
            got_late = srv.query_device(late, return_indices=True)
        assert np.array_equal(got_late, want[40:43])
        # and host descriptors (a CPU tensor) take the same chain (the library's worker copies them into its pinned slots)
        if native:
            assert np.array_equal(srv.query_device(qd[50:52].cpu(), return_indices=True), want[50:52])
    finally:
        srv.close()
        nnsearch.drop_cached_galleries()


def test_online_chain_of_the_library_alone():
    """mi_online_* without the Searcher around it (include/mi355_retrieval.h): the plain-search mode (no second gallery) with
    scores and host descriptors; a chain that fails fails EVERY request it carried, with the chain's message on each caller's
    thread; the handle recovers; destroy answers what is queued; a closed gallery is refused on the Python side."""
    import threading
    from isehr_amd import _lib
    from isehr_amd.synth import synth_rows
    n, d, K = 20000, 128, 10
    g = synth_rows(95, 0, n, d)
    g[5000:5400] = g[4999]                                          # 401 identical rows: more candidates than a rescore_cap of 64
    q = synth_rows(96, 0, 8, d)
    G = _lib.Gallery.from_host(g)
    chain = _lib.OnlineChain(G, None, K, max_batch=128, max_wait_us=500)
    try:
        want_i, want_s, _ = G.search(q, K)
        for j in range(8):
            idx, sc = chain.query(q[j:j + 1].ctypes.data, 1, _lib.MI_HOST, scores=True)
            assert np.array_equal(idx[0], want_i[j]) and np.array_equal(sc[0], want_s[j])
        idx = chain.query(q.ctypes.data, 8, _lib.MI_HOST)             # one request of eight rows
        assert np.array_equal(idx, want_i) and chain.stats() == {"chains": 9, "requests": 9}
        with pytest.raises(RuntimeError, match="max_batch"):
            chain.query(np.zeros((129, d), np.float32).ctypes.data, 129, _lib.MI_HOST)
        # a failing chain: the duplicated row as a query overflows the candidate buffer, and overflow is an error now
        G.set_option("rescore_cap", 64)
        G.set_option("exact_fallback", 0)
        bad = np.ascontiguousarray(g[4999:5000])
        with pytest.raises(RuntimeError, match="overflow"):
            chain.query(bad.ctypes.data, 1, _lib.MI_HOST)
        msgs = []

        def client(j):
            try:
                chain.query((bad if j == 3 else q[j:j + 1]).ctypes.data, 1, _lib.MI_HOST)
                msgs.append(None)
            except RuntimeError as e:
                msgs.append(str(e))
        G.set_option("exact_fallback", 1)
        G.set_option("rescore_cap", 1024)
        # healthy again: the same requests, concurrently, all answered (the duplicated row's best rows are the 401 copies)
        ths = [threading.Thread(target=client, args=(j,)) for j in range(8)]
        for th in ths:
            th.start()
        for th in ths:
            th.join(timeout=60)
        assert msgs == [None] * 8
        idx = chain.query(bad.ctypes.data, 1, _lib.MI_HOST)
        assert set(idx[0]) <= set(range(4999, 5400)) and np.array_equal(idx[0], np.arange(4999, 4999 + K))
        # the gallery cannot be destroyed under the handle (its worker would search freed memory with the next request)
        with pytest.raises(RuntimeError, match="online handle"):
            G.close()
        assert np.array_equal(chain.query(q[2:3].ctypes.data, 1, _lib.MI_HOST)[0], want_i[2])
    finally:
        chain.close()
        chain.close()                                               # idempotent
        G.close()
    with pytest.raises(RuntimeError, match="closed"):
        _lib.OnlineChain.query(chain, q.ctypes.data, 1, _lib.MI_HOST)


def test_online_chain_packs_whole_requests_and_drains_on_destroy():
    """mi_online_*: a chain holds WHOLE requests up to max_batch rows (a request that does not fit waits for the next chain, its
    rows stay together), device and host descriptors mix in one chain, handles come and go, and destroy answers what is queued."""
    import threading
    import torch
    from isehr_amd import _lib
    from isehr_amd.synth import synth_rows
    n, d, K = 12000, 192, 7
    G = _lib.Gallery.from_host(synth_rows(97, 0, n, d))
    R = _lib.Gallery.from_host(synth_rows(97, 0, n, d), norm_mode=_lib.NORM_NONE)
    q = synth_rows(98, 0, 36, d)
    qd = torch.from_numpy(q).cuda()
    torch.cuda.synchronize()
    try:
        for _ in range(3):                                          # create / destroy with nothing queued
            _lib.OnlineChain(G, R, K, max_batch=8, max_wait_us=0).close()
        chain = _lib.OnlineChain(G, R, K, k_qe=3, w=4.0, max_batch=8, max_wait_us=300)
        want = np.concatenate([chain.query(q[i:i + 3].ctypes.data, 3, _lib.MI_HOST) for i in range(0, 36, 3)])
        base = chain.stats()
        assert base == {"chains": 12, "requests": 12}
        got = np.full((36, K), -1, dtype=np.int64)
        errs = []

        def client(t):                                              # requests of 3 rows: at most two fit a chain of 8
            try:
                for rep in range(4):
                    i = 3 * t
                    if t % 2:
                        got[i:i + 3] = chain.query(qd[i:i + 3].data_ptr(), 3, _lib.MI_DEVICE)
                    else:
                        got[i:i + 3] = chain.query(q[i:i + 3].ctypes.data, 3, _lib.MI_HOST)
            except Exception as e:                                  # noqa: BLE001
                errs.append(e)
        ths = [threading.Thread(target=client, args=(t,)) for t in range(12)]
        for th in ths:
            th.start()
        for th in ths:
            th.join(timeout=120)
        assert not errs, errs[:1]
        assert np.array_equal(got, want)
        st = chain.stats()
        assert st["requests"] - base["requests"] == 48 and (st["chains"] - base["chains"]) * 2 >= 48
        chain.close()
        # destroy while requests are waiting: every one of them is answered first.  The worker is held inside a chain whose only
        # descriptor is still being produced (behind ~0.1 s of device sleep on its producer's stream); twelve requests queue
        # behind it, then the handle is destroyed
        c2 = _lib.OnlineChain(G, R, K, k_qe=3, w=4.0, max_batch=64, max_wait_us=300)
        try:
            out = np.full((13, K), -1, dtype=np.int64)
            side = torch.cuda.Stream()
            late = torch.zeros((1, d), device="cuda")
            torch.cuda.synchronize()
            with torch.cuda.stream(side):
                torch.cuda._sleep(250_000_000)
                late.copy_(qd[35:36], non_blocking=True)

            def held():
                out[12] = c2.query(late.data_ptr(), 1, _lib.MI_DEVICE, True, side.cuda_stream)[0]

            def single(t):
                out[t] = c2.query(q[3 * t:3 * t + 1].ctypes.data, 1, _lib.MI_HOST)[0]
            ths = [threading.Thread(target=held)]
            ths[0].start()
            import time
            for _ in range(400):                                    # until the worker has taken the held request, alone
                if c2.stats()["requests"] == 1:
                    break
                time.sleep(0.0005)
            ths += [threading.Thread(target=single, args=(t,)) for t in range(12)]
            for th in ths[1:]:
                th.start()
            time.sleep(0.02)                                        # all twelve are inside the call, queued behind the held chain
            assert (out == -1).all() and c2.stats() == {"chains": 1, "requests": 1}
        finally:
            c2.close()                                              # mi_online_destroy: drains the queue, then stops the worker
        for th in ths:
            th.join(timeout=60)
        assert np.array_equal(out[:12], want[0::3]) and np.array_equal(out[12], want[35])
    finally:
        G.close()
        R.close()


@pytest.mark.parametrize("mode,gpus", [("100", "0"), ("mAP", "0"), ("100", "0,0")])
def test_test_rop1m_driver(feature_store, capsys, mode, gpus):
    """--gpu-id 0,0: two row shards inside the driver's one process (both on the test box's only GPU)."""
    from isehr_amd import evaluate
    from isehr_amd.entry import test_rOP1m
    vecs, qvecs, gnd = feature_store
    assert test_rOP1m.main(["--datasets", "roxford5k", "--ifextracted", "--mode", mode, "--gpu-id", gpus]) == 0
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


def test_main_retrieve_driver(tmp_path, monkeypatch, capsys):
    """src/main_retrieve.py's evaluation flow on numpy feature files: feature store written, the FULL inner-product ranking
    (:175-176) equal to the oracle's argsort wherever scores are not within 1e-6, mAP and mP@k lines equal to the
    first-generation evaluator's (pinned by map_v1.npz), and the whitened second pass of src/main_train.py:709-716."""
    from isehr_amd.entry import main_retrieve
    from isehr_amd.entry.features import load_path_features
    monkeypatch.chdir(tmp_path)
    vecs, qvecs, gnd = planted_dataset(79, 3000, 64, 10)
    os.makedirs("outputs", exist_ok=True)
    os.makedirs("data/test/rparis6k", exist_ok=True)
    np.save("outputs/rparis6k_vecs.npy", vecs)
    np.save("outputs/rparis6k_qvecs.npy", qvecs)
    with open("data/test/rparis6k/gnd_rparis6k.pkl", "wb") as f:
        pickle.dump({"gnd": gnd, "imlist": ["im%d" % i for i in range(3000)], "qimlist": ["q%d" % i for i in range(10)]}, f)
    rng = np.random.default_rng(5)
    Lw = {"m": vecs.mean(axis=1, keepdims=True).astype(np.float64), "P": rng.standard_normal((64, 64)) / 8 + np.eye(64)}
    with open("Lw.pkl", "wb") as f:
        pickle.dump(Lw, f)
    assert main_retrieve.main(["--datasets", "rparis6k", "--whitening", "Lw.pkl"]) == 0
    out = capsys.readouterr().out.splitlines()
    v2, paths = load_path_features("rparis6k_database")
    assert np.array_equal(v2, vecs) and paths[5] == "im5"
    want_ranks, scores = oracle.ip_rank(vecs.astype(np.float64), qvecs.astype(np.float64))
    res = main_retrieve.evaluate_dataset("rparis6k", vecs, qvecs, gnd, Lw)
    capsys.readouterr()
    for got, sc, r in ((res["ranks"], scores, want_ranks),):
        assert got.shape == r.shape
        diff = np.argwhere(got != r)
        for pos, qi in diff:
            assert abs(sc[got[pos, qi], qi] - sc[r[pos, qi], qi]) <= 1e-6
    vl, ql = oracle.whitenapply(vecs.astype(np.float64), Lw["m"], Lw["P"]), oracle.whitenapply(qvecs.astype(np.float64), Lw["m"], Lw["P"])
    r_lw, s_lw = oracle.ip_rank(vl, ql)
    for pos, qi in np.argwhere(res["ranks_lw"] != r_lw):
        assert abs(s_lw[res["ranks_lw"][pos, qi], qi] - s_lw[r_lw[pos, qi], qi]) <= 1e-6
    # printed lines: mAP + mP@k for the plain and the whitened descriptors, values of the oracle's evaluator
    map_lines = [ln for ln in out if "mAP E:" in ln]
    pr_lines = [ln for ln in out if "mP@k[1, 5, 10]" in ln]
    assert len(map_lines) == 2 and len(pr_lines) == 2 and "rparis6k + whiten" in map_lines[1]
    for ranks_, line, pline in ((want_ranks, map_lines[0], pr_lines[0]), (r_lw, map_lines[1], pr_lines[1])):
        vals, prs = [], []
        for okk, jk in evaluate_emh():
            gt = [{"ok": np.concatenate([g[k] for k in okk]), "junk": np.concatenate([g[k] for k in jk])} for g in gnd]
            mp, _, pr, _ = oracle.compute_map_kappas(ranks_, gt, (1, 5, 10))
            vals.append(mp)
            prs.append(pr)
        for v in vals:
            assert str(np.around(v * 100, decimals=2)) in line, (line, vals)
        for p in prs:
            assert str(np.around(p * 100, decimals=2)) in pline, (pline, prs)


def evaluate_emh():
    from isehr_amd import evaluate
    return evaluate.EMH


def test_main_retrieve_extract_flow(tmp_path, monkeypatch, capsys):
    """--extract: image tensors -> ResNet-SOA trunk (reduced depth / width: the flow, not the weights) -> HIP descriptor
    tail -> device gallery -> descriptors; the ranking of the driver equals the oracle's on the descriptors it stored."""
    import torch
    from isehr_amd.entry import main_retrieve
    from isehr_amd.entry.features import load_path_features
    monkeypatch.chdir(tmp_path)
    os.makedirs("outputs", exist_ok=True)
    os.makedirs("data/test/roxford5k", exist_ok=True)
    g = torch.Generator().manual_seed(3)
    imgs = torch.randn((21, 3, 64, 48), generator=g)
    torch.save(imgs, "outputs/roxford5k_images.pt")
    torch.save(imgs[[2, 7, 11]] + 0.05 * torch.randn((3, 3, 64, 48), generator=g), "outputs/roxford5k_qimages.pt")
    gnd = [{"easy": np.array([i]), "hard": np.array([(i + 1) % 21]), "junk": np.array([], dtype=np.int64)} for i in (2, 7, 11)]
    with open("data/test/roxford5k/gnd_roxford5k.pkl", "wb") as f:
        pickle.dump({"gnd": gnd}, f)
    torch.manual_seed(11)
    assert main_retrieve.main(["--datasets", "roxford5k", "--extract", "--blocks", "1,1,1,1", "--width", "8",
                               "--multiscale", "1,1.4142135,0.70710678", "--batch", "5"]) == 0
    out = capsys.readouterr().out
    vecs, _ = load_path_features("roxford5k_database")
    qvecs, _ = load_path_features("roxford5k_query")
    assert vecs.shape == (256, 21) and qvecs.shape == (256, 3)
    assert np.allclose(np.linalg.norm(vecs, axis=0), 1.0, atol=1e-5)          # extract_ms ends with v / ||v||
    want = oracle.compute_map_revisited(oracle.ip_rank(vecs.astype(np.float64), qvecs.astype(np.float64))[0], gnd)
    line = [ln for ln in out.splitlines() if "mAP E:" in ln][0]
    for v in want:
        assert str(np.around(v * 100, decimals=2)) in line, (line, want)


def test_test_custom_driver(tmp_path, monkeypatch, capsys):
    """src/test_custom.py: full ranking of a directory-labelled collection (K = database size), mAP_custom, the pickle of
    ranked database paths per query."""
    from isehr_amd.entry import test_custom
    monkeypatch.chdir(tmp_path)
    rng = np.random.default_rng(8)
    centres = rng.standard_normal((6, 48))
    lab_d = np.arange(240) % 6
    d = centres[lab_d] + 2.0 * rng.standard_normal((240, 48))
    lab_q = np.arange(12) % 6
    q = centres[lab_q] + 2.0 * rng.standard_normal((12, 48))
    paths_d = ["data/test/custom/database/c%d/%03d.jpg" % (lab_d[i], i) for i in range(240)]
    paths_q = ["data/test/custom/query/c%d/q%02d.jpg" % (lab_q[i], i) for i in range(12)]
    os.makedirs("outputs/features", exist_ok=True)
    for name, feat, paths in (("custom_query", q, paths_q), ("custom_database", d, paths_d)):
        with open("outputs/features/%s_path_feature_2.pkl" % name, "wb") as f:
            pickle.dump({"path": paths, "feature": feat.T.astype(np.float32)}, f)
    assert test_custom.main([]) == 0
    out = capsys.readouterr().out
    want_idx = oracle.matching_l2(240, d.astype(np.float32), q.astype(np.float32))
    want = oracle.map_custom(240, want_idx, paths_q, paths_d)
    got = float([ln for ln in out.splitlines() if ln.startswith("mean average precision")][0].split(":")[1])
    assert abs(got - want) <= 2e-3 and 0.3 < want < 1.0                      # near-ties may swap neighbours of one label
    with open("outputs/ranks/custom_ranking_result.pkl", "rb") as f:
        rank_res = pickle.load(f)
    assert set(rank_res) == set(paths_q) and all(sorted(v) == sorted(paths_d) for v in rank_res.values())
    gn = d / np.linalg.norm(d, axis=1, keepdims=True)
    qn = q / np.linalg.norm(q, axis=1, keepdims=True)
    s = qn @ gn.T
    for i, pq in enumerate(paths_q):
        order = [paths_d.index(p) for p in rank_res[pq]]
        assert (np.diff(s[i, order]) <= 1e-6).all()
