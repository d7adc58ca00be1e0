"""CPU: the build's own mAP implementation against the oracle (pinned by the reference goldens)."""
import os

import numpy as np

import oracle
from isehr_amd import evaluate
from isehr_amd.synth import planted_dataset


def test_map_matches_oracle_and_golden(golden_dir):
    z = np.load(os.path.join(golden_dir, "map.npz"))
    vecs, qv, gnd = planted_dataset(41, 1200, 64, 12)
    rk = np.argsort(-(vecs.T @ qv), axis=0)
    for name, r in (("full", rk), ("top100", rk[:100])):
        got = evaluate.compute_map_revisited(r, gnd)
        assert np.allclose(got, z[f"{name}_map_EMH"], rtol=0, atol=1e-12)
        assert np.allclose(got, oracle.compute_map_revisited(r, gnd), rtol=0, atol=1e-12)


def test_map_edge_cases():
    # a query without positives is skipped; junk before a positive shifts it up
    ranks = np.array([[3, 0], [1, 1], [2, 2], [0, 3]])
    gnd = [{"ok": np.array([2]), "junk": np.array([3, 1])}, {"ok": np.array([], dtype=int), "junk": np.array([])}]
    mp, aps = evaluate.compute_map(ranks, gnd)
    assert mp == 1.0 and np.isnan(aps[1])          # positive 2 sits behind two junk entries -> rank 0
    mo, _ = oracle.compute_map2(ranks, gnd)
    assert mo == mp


def test_map_from_positions_equals_map_of_the_full_ranking():
    """compute_map_from_positions (positions of the labelled images only) == compute_map on the complete ranking, and a
    truncated ranking scores lower when a positive lies beyond the cut -- the gap the QGE large-N branch must not have."""
    import numpy as np
    from isehr_amd import evaluate
    rng = np.random.default_rng(3)
    n, nq = 5000, 9
    ranks = np.stack([rng.permutation(n) for _ in range(nq)], axis=1)           # [N, Q]
    gnd = []
    for i in range(nq):
        ids = rng.choice(n, size=40, replace=False)
        gnd.append({"easy": ids[:10], "hard": ids[10:25], "junk": ids[25:]})
    inv = np.empty_like(ranks)
    for i in range(nq):
        inv[ranks[:, i], i] = np.arange(n)
    position_of = [{int(v): int(inv[v, i]) for k in ("easy", "hard", "junk") for v in gnd[i][k]} for i in range(nq)]
    full = evaluate.compute_map_revisited(ranks, gnd)
    got = evaluate.compute_map_revisited_from_positions(gnd, position_of)
    assert np.allclose(full, got, rtol=0, atol=1e-15)
    cut = evaluate.compute_map_revisited(ranks[:1000], gnd)
    assert all(c < f for c, f in zip(cut, full))


def test_first_generation_evaluator_and_custom_map_vs_reference_golden(golden_dir):
    """src/utils/evaluate.py: compute_map with kappas (mAP + precision at 1 / 5 / 10, the evaluator behind
    src/main_retrieve.py:176) and mAP_custom (src/test_custom.py:33), both captured from the reference's own functions
    (oracle/make_golden.py -> map_v1.npz): the oracle restatement and the build's vectorised implementation reproduce them."""
    z = np.load(os.path.join(golden_dir, "map_v1.npz"))
    vecs, qv, gnd = planted_dataset(41, 1200, 64, 12)
    rk = np.argsort(-(vecs.T @ qv), axis=0)
    for (okk, jk), tag in zip(evaluate.EMH, "EMH"):
        gt = [{"ok": np.concatenate([g[k] for k in okk]), "junk": np.concatenate([g[k] for k in jk])} for g in gnd]
        mo, aps_o, pr_o, prs_o = oracle.compute_map_kappas(rk, gt, (1, 5, 10))
        assert mo == float(z["map_" + tag]) and np.array_equal(aps_o, z["aps_" + tag], equal_nan=True)
        assert np.array_equal(pr_o, z["pr_" + tag]) and np.array_equal(prs_o, z["prs_" + tag], equal_nan=True)
        mp, aps = evaluate.compute_map(rk, gt)
        pr, prs = evaluate.precision_at(rk, gt, (1, 5, 10))
        assert abs(mp - float(z["map_" + tag])) <= 1e-12 and np.allclose(aps, z["aps_" + tag], rtol=0, atol=1e-12, equal_nan=True)
        assert np.allclose(pr, z["pr_" + tag], rtol=0, atol=1e-12)
        assert np.allclose(prs, z["prs_" + tag], rtol=0, atol=1e-12, equal_nan=True)
    idx, keep = z["custom_idx"], z["custom_keep"]
    pd_, pq_ = [str(p) for p in z["custom_paths_d"]], [str(p) for p in z["custom_paths_q"]]
    K = idx.shape[1]
    want = float(z["custom_map"])
    assert oracle.map_custom(K, idx[keep], [pq_[i] for i in keep], pd_) == want
    assert abs(evaluate.map_custom(K, idx[keep], [pq_[i] for i in keep], pd_) - want) <= 1e-12
    assert 0.0 < want < 1.0


def test_map_and_print_with_kappas_prints_the_precision_line(capsys):
    vecs, qv, gnd = planted_dataset(41, 1200, 64, 12)
    rk = np.argsort(-(vecs.T @ qv), axis=0)
    evaluate.compute_map_and_print("roxford5k", rk, gnd, kappas=[1, 5, 10])
    out = capsys.readouterr().out.splitlines()
    assert out[0].startswith(">> roxford5k: mAP E: ") and out[1].startswith(">> roxford5k: mP@k[1, 5, 10] E: [")
