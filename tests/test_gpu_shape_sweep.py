"""GPU: seeded sweep over shapes, K, image types and data with structure (clusters, duplicates, heavy ties) -- every answer
against the float64 oracle and against the f32 scorer of the same handle (`force_exact`).

The uniform synthetic rows of the other tests are the benign case for the sample-based threshold (src/utils/nnsearch.py:699-703
is an exhaustive search: whatever the data, the answer is THE top-K).  Clustered galleries put the K-th score far out in the
tail of the score distribution, duplicates put ties across it, so the verification / repair / fallback branches of the
certificate run here, on both scoring kernels (<= 128 queries: streaming kernel; more: MFMA tile kernel).
"""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu

TAU = 1e-6


@pytest.fixture(scope="module")
def lib():
    from isehr_amd import _lib
    _lib.load()
    return _lib


def _clustered(rng, n, d, ncl, spread):
    """Rows around `ncl` random centres; queries near some of the centres (so the K nearest are one cluster's members)."""
    c = rng.standard_normal((ncl, d))
    c /= np.linalg.norm(c, axis=1, keepdims=True)
    lab = rng.integers(0, ncl, n)
    g = c[lab] + spread * rng.standard_normal((n, d)) / np.sqrt(d)
    return g.astype(np.float32), c


def _case(rng, kind, n, d, nq, k):
    if kind == "uniform":
        g = rng.standard_normal((n, d)).astype(np.float32)
        q = rng.standard_normal((nq, d)).astype(np.float32)
    elif kind == "clustered":
        g, c = _clustered(rng, n, d, max(2, n // 400), 0.6)
        q = (c[rng.integers(0, len(c), nq)] + 0.3 * rng.standard_normal((nq, d)) / np.sqrt(d)).astype(np.float32)
    elif kind == "few_big_clusters":
        # clusters much larger than K: thousands of rows within a hair of the K-th score
        g, c = _clustered(rng, n, d, 3, 0.05)
        q = (c[rng.integers(0, 3, nq)] + 0.02 * rng.standard_normal((nq, d)) / np.sqrt(d)).astype(np.float32)
    elif kind == "duplicates":
        base = rng.standard_normal((max(k // 2, 8), d)).astype(np.float32)
        g = rng.standard_normal((n, d)).astype(np.float32)
        rows = rng.choice(n, size=min(n, 6 * len(base)), replace=False)
        g[rows] = base[np.arange(len(rows)) % len(base)]                  # every base row six times: exact ties
        q = (base[rng.integers(0, len(base), nq)] + 0.5 * rng.standard_normal((nq, d))).astype(np.float32)
    else:
        raise AssertionError(kind)
    return g, q


def _check(lib, g, q, k, image_f16, options=()):
    from isehr_amd._lib import Gallery
    lib.set_global_option("image_dtype", 1 if image_f16 else 0)
    try:
        G = Gallery.from_host(g)
    finally:
        lib.set_global_option("image_dtype", 1)
    try:
        for name, val in options:
            G.set_option(name, val)
        i1, s1, _ = G.search(q, k)
        G.set_option("force_exact", 1)
        i2, s2, _ = G.search(q, k)
    finally:
        G.close()
    assert i1.shape == (q.shape[0], k)
    assert np.array_equal(i1, i2), "certified MFMA path and f32 scorer disagree"
    assert np.array_equal(s1, s2)
    s = oracle.exact_scores_f64(g, q)
    assert oracle.check_topk_parity(i1, s, k, TAU) == []
    assert np.abs(np.take_along_axis(s, i1, 1) - s1).max() < 3e-7


def _draw_cases():
    rng = np.random.default_rng(20261004)
    kinds = ["uniform", "clustered", "few_big_clusters", "duplicates"]
    cases = []
    for i in range(24):
        kind = kinds[i % 4]
        tile_kernel = (i // 4) % 2 == 1
        nq = int(rng.integers(129, 600)) if tile_kernel else int(rng.integers(1, 129))
        d = int(rng.choice([32, 100, 128, 200, 256, 320]))
        n = int(rng.integers(17000, 90000)) if i % 3 else int(rng.integers(300, 16000))
        k = int(rng.choice([1, 10, 100, 100, 257, 1000]))
        k = min(k, n)
        cases.append((i, kind, n, d, nq, k, bool(i % 5 != 0)))
    return cases


@pytest.mark.parametrize("case", _draw_cases(), ids=lambda c: "%02d-%s-n%d-d%d-q%d-k%d-%s" % (
    c[0], c[1], c[2], c[3], c[4], c[5], "f16" if c[6] else "bf16"))
def test_shape_sweep_against_oracle(lib, case):
    i, kind, n, d, nq, k, f16 = case
    rng = np.random.default_rng(1000 + i)
    g, q = _case(rng, kind, n, d, nq, k)
    _check(lib, g, q, k, f16)


@pytest.mark.parametrize("kind", ["clustered", "few_big_clusters", "duplicates"])
@pytest.mark.parametrize("ladder", [0, 1])
def test_structured_data_on_the_sample_schedule(lib, kind, ladder):
    """Shards large enough for the single-launch sample schedule (>= 64 tiles) and a full 256-query tile and a half."""
    rng = np.random.default_rng(77 + ladder)
    g, q = _case(rng, kind, 70000, 128, 384, 100)
    _check(lib, g, q, 100, True, options=(("ladder", ladder),))


@pytest.mark.parametrize("k", [100, 1000])
def test_massive_ties_take_the_dense_f64_path(lib, k):
    """Thousands of rows within 1e-7 of the K-th score: no candidate buffer holds them, the filtered passes overflow and the
    search falls through to dense f64 scores + exact selection (csrc/api.hip search_sync).  The contract does not change:
    the order is that of the exact scores of the stored rows, so against the float64 truth only the f32 rounding of the
    stored rows remains (a few 1e-8 here)."""
    from isehr_amd._lib import Gallery
    rng = np.random.default_rng(5)
    n, d, nq = 30000, 128, 140
    g = rng.standard_normal((n, d)).astype(np.float32)
    base = rng.standard_normal(d)
    near = rng.choice(n, size=9000, replace=False)
    g[near] = (base + 2e-4 * rng.standard_normal((len(near), d))).astype(np.float32)
    g[near[:40]] = g[near[40]]                                             # and a block of exact duplicates
    q = (base + 0.05 * rng.standard_normal((nq, d))).astype(np.float32)
    G = Gallery.from_host(g)
    try:
        idx, sc, _ = G.search(q, k)
        st = G.status()
    finally:
        G.close()
    assert st["overflow_batches"] >= 2, st                                 # both filtered passes gave up
    s = oracle.exact_scores_f64(g, q)
    assert oracle.check_topk_parity(idx, s, k, 2e-7) == []
    assert (np.diff(sc, axis=1) <= 0).all()
    assert np.abs(np.take_along_axis(s, idx, 1) - sc).max() < 3e-7
    # exact duplicates are identical stored rows: equal scores, lower index first
    dup = set(int(v) for v in near[:41])
    for r in range(0, nq, 17):
        pos = [p for p, v in enumerate(idx[r]) if int(v) in dup]
        got = [int(idx[r, p]) for p in pos]
        assert got == sorted(got)
