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
    # ISEHR_SWEEP_SEED draws another set of shapes (wider fuzzing by hand; the default set is what CI runs)
    import os
    rng = np.random.default_rng(int(os.environ.get("ISEHR_SWEEP_SEED", "20261004")))
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
    import os
    rng = np.random.default_rng(1000 + i + 7919 * (int(os.environ.get("ISEHR_SWEEP_SEED", "20261004")) % 1000))
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
    search falls through to dense f64 scores + exact selection (csrc/api_schedule.hip search_sync).  The contract does not change:
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


def _simulated_shards(g, qh, k, nshards, norm_mode, options=()):
    """The sharded protocol with every shard on this GPU (stack == all-gather), including the agreement on the error
    norms and the image type that ShardedGallery.__init__ performs with two all-reduces."""
    import torch
    from isehr_amd import _lib
    from isehr_amd.sharded import shard_bounds
    n, nq = g.shape[0], qh.shape[0]
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    q = torch.from_numpy(qh).to(dev)
    shards = []
    for r in range(nshards):
        lo, hi = shard_bounds(n, nshards, r)
        shards.append(_lib.Gallery.from_host(g[lo:hi], norm_mode=norm_mode, row_offset=lo))
    try:
        f16 = min(int(sh.get_option("image_dtype")) for sh in shards)
        for sh in shards:
            if int(sh.get_option("image_dtype")) != f16:
                sh.set_image_dtype(f16)
        bounds = np.max(np.array([sh.norm_bounds() for sh in shards]), axis=0)
        for sh in shards:
            sh.norm_bounds(raise_to=[float(v) for v in bounds])
            for name, val in options:
                sh.set_option(name, val)
        approx = torch.empty((nshards, nq, k), dtype=torch.float32, device=dev)
        for r, sh in enumerate(shards):
            sh.phase1_device(q.data_ptr(), nq, k, approx[r].data_ptr(), stream)
        L = torch.empty((nq,), dtype=torch.float32, device=dev)
        _lib.kth_of_gathered_device(approx.data_ptr(), nshards, nq, k, L.data_ptr(), stream)
        idx = torch.empty((nshards, nq, k), dtype=torch.int64, device=dev)
        sc = torch.empty((nshards, nq, k), dtype=torch.float32, device=dev)
        sc64 = torch.empty((nshards, nq, k), dtype=torch.float64, device=dev)
        for r, sh in enumerate(shards):
            sh.phase2_device(nq, k, L.data_ptr(), idx[r].data_ptr(), sc[r].data_ptr(), sc64[r].data_ptr(), stream)
        oi = torch.empty((nq, k), dtype=torch.int64, device=dev)
        os_ = torch.empty((nq, k), dtype=torch.float32, device=dev)
        _lib.topk_merge_device(sc64.data_ptr(), idx.data_ptr(), nshards, nq, k, oi.data_ptr(), os_.data_ptr(), stream)
        torch.cuda.synchronize()
        flags = [sh.flags() for sh in shards]
    finally:
        for sh in shards:
            sh.close()
    return oi.cpu().numpy(), os_.cpu().numpy(), flags


@pytest.mark.parametrize("nq", [40, 300])
@pytest.mark.parametrize("nshards", [2, 5])
def test_heterogeneous_shards_equal_the_single_gallery(lib, nshards, nq):
    """Rows sorted by cluster: one shard holds all of a query's neighbours, the others none, so the shards' local K-th
    scores are far apart and only the gathered L prunes the poor shards' candidates."""
    from isehr_amd._lib import Gallery
    rng = np.random.default_rng(31)
    n, d, k = 60000, 128, 100
    c = rng.standard_normal((12, d))
    lab = np.sort(rng.integers(0, 12, n))
    g = (c[lab] + 0.8 * rng.standard_normal((n, d))).astype(np.float32)
    qh = (c[rng.integers(0, 12, nq)] + 0.5 * rng.standard_normal((nq, d))).astype(np.float32)
    single = Gallery.from_host(g)
    ref_idx, ref_sc, _ = single.search(qh, k)
    single.close()
    oi, os_, flags = _simulated_shards(g, qh, k, nshards, lib.NORM_L2)
    assert not any(flags), flags
    assert np.array_equal(oi, ref_idx)
    assert np.array_equal(os_, ref_sc)
    assert oracle.check_topk_parity(oi, oracle.exact_scores_f64(g, qh), k, TAU) == []


@pytest.mark.parametrize("scale", [1.0, 300.0])
def test_raw_inner_product_with_heavy_tailed_norms(lib, scale):
    """src/main_retrieve.py:175-176 on un-normalised vectors whose norms span two decades (and, scaled by 300, leave the
    fp16 range: the image falls back to bf16): the error margin follows the LARGEST norms, single gallery and shards."""
    from isehr_amd._lib import Gallery
    rng = np.random.default_rng(41)
    n, d, nq, k = 40000, 96, 150, 50
    g = rng.standard_normal((n, d)) * np.exp(1.5 * rng.standard_normal((n, 1)))
    g = (scale * g).astype(np.float32)
    qh = rng.standard_normal((nq, d)).astype(np.float32)
    G = Gallery.from_host(g, norm_mode=lib.NORM_NONE)
    try:
        idx, sc, _ = G.search(qh, k)
        f16 = int(G.get_option("image_dtype"))
    finally:
        G.close()
    assert f16 == 0                                          # rows far longer than 4: the image is bf16 at either scale
    s = oracle.exact_scores_f64(g, qh, normalize=False)
    tol = TAU * float(np.abs(s).max())                       # the tolerance is stated on the score scale
    assert oracle.check_topk_parity(idx, s, k, tol) == []
    assert np.abs(np.take_along_axis(s, idx, 1) - sc).max() <= 2e-7 * float(np.abs(s).max())
    # shards whose largest norms differ: identical answer after the agreement on the norm bounds
    order = np.argsort(np.linalg.norm(g, axis=1), kind="stable")
    gs = g[order]                                            # shard 0 = the small rows, the last shard = the large ones
    Gs = Gallery.from_host(gs, norm_mode=lib.NORM_NONE)
    ref_idx, ref_sc, _ = Gs.search(qh, k)
    Gs.close()
    # the shard of small rows cannot filter with a margin that follows the largest norms of the gallery: its buffers
    # overflow and raise the sticky flags; ShardedGallery.search(verify=True) then answers again with these fallbacks
    for options in ((), (("speculative", 0),), (("force_exact", 1),)):
        oi, os_, flags = _simulated_shards(gs, qh, k, 3, lib.NORM_NONE, options)
        if not any(flags):
            break
    assert not any(flags), flags
    assert options, "expected the small-norm shard to need a fallback"
    assert np.array_equal(oi, ref_idx)
    assert np.array_equal(os_, ref_sc)


@pytest.mark.parametrize("n,k", [(2048, 2048), (2300, 2048), (5000, 1), (257, 256), (70000, 2048)])
def test_k_extremes_on_the_tile_kernel(lib, n, k):
    rng = np.random.default_rng(n + k)
    g = rng.standard_normal((n, 64)).astype(np.float32)
    q = rng.standard_normal((260, 64)).astype(np.float32)
    _check(lib, g, q, k, True)


@pytest.mark.parametrize("nshards,k", [(4, 1000), (3, 2048), (8, 384)])
def test_merge_of_long_lists(lib, nshards, k):
    """nshards * k entries per query beyond what the merge stages in LDS (48 KiB = 3072 entries): the ranks are then
    searched in the gathered lists where they lie.  Same answer as the single gallery, bit for bit."""
    from isehr_amd._lib import Gallery
    rng = np.random.default_rng(nshards * 1000 + k)
    n, d, nq = 50000, 64, 37
    g = rng.standard_normal((n, d)).astype(np.float32)
    g[5000:5040] = g[4999]                                   # a tie block that straddles nothing but must keep index order
    qh = rng.standard_normal((nq, d)).astype(np.float32)
    single = Gallery.from_host(g)
    ref_idx, ref_sc, _ = single.search(qh, k)
    single.close()
    oi, os_, flags = _simulated_shards(g, qh, k, nshards, lib.NORM_L2)
    assert not any(flags), flags
    assert np.array_equal(oi, ref_idx)
    assert np.array_equal(os_, ref_sc)


def test_two_workspace_slots_keep_two_batches_apart(lib):
    """Option "workspace_slot": phase 1 of a second batch may run before phase 2 of the first (the pipelined sharded search
    does exactly that while a batch waits for its collective).  Interleaved on one shard: both answers equal the
    sequential ones; flags and statistics stay one set."""
    import torch
    from isehr_amd._lib import Gallery
    rng = np.random.default_rng(9)
    n, d, k = 50000, 96, 64
    g = rng.standard_normal((n, d)).astype(np.float32)
    qa = rng.standard_normal((300, d)).astype(np.float32)
    qb = rng.standard_normal((77, d)).astype(np.float32)
    G = Gallery.from_host(g)
    try:
        ra, rb = G.search(qa, k), G.search(qb, k)
        dev = torch.device("cuda", 0)
        stream = torch.cuda.current_stream().cuda_stream
        ta, tb = torch.from_numpy(qa).to(dev), torch.from_numpy(qb).to(dev)

        def bufs(nq):
            return dict(approx=torch.empty((nq, k), dtype=torch.float32, device=dev),
                        L=torch.empty((nq,), dtype=torch.float32, device=dev),
                        idx=torch.empty((nq, k), dtype=torch.int64, device=dev),
                        sc=torch.empty((nq, k), dtype=torch.float32, device=dev),
                        sc64=torch.empty((nq, k), dtype=torch.float64, device=dev))
        A, B = bufs(300), bufs(77)
        G.set_option("workspace_slot", 0)
        G.phase1_device(ta.data_ptr(), 300, k, A["approx"].data_ptr(), stream)
        G.set_option("workspace_slot", 1)
        G.phase1_device(tb.data_ptr(), 77, k, B["approx"].data_ptr(), stream)       # before phase 2 of the first batch
        for slot, X, nq in ((0, A, 300), (1, B, 77)):
            G.set_option("workspace_slot", slot)
            lib.kth_of_gathered_device(X["approx"].data_ptr(), 1, nq, k, X["L"].data_ptr(), stream)
            G.phase2_device(nq, k, X["L"].data_ptr(), X["idx"].data_ptr(), X["sc"].data_ptr(), X["sc64"].data_ptr(), stream)
        torch.cuda.synchronize()
        assert G.get_option("workspace_slot") == 1
        G.set_option("workspace_slot", 0)
        assert G.flags() == 0
        assert np.array_equal(A["idx"].cpu().numpy(), ra[0]) and np.array_equal(A["sc"].cpu().numpy(), ra[1])
        assert np.array_equal(B["idx"].cpu().numpy(), rb[0]) and np.array_equal(B["sc"].cpu().numpy(), rb[1])
        # a larger K re-allocates the active workspace and drops the parked one: both slots keep working
        G.set_option("workspace_slot", 1)
        i2, s2, _ = G.search(qb, 200)
        G.set_option("workspace_slot", 0)
        i3, s3, _ = G.search(qb, 200)
        assert np.array_equal(i2, i3) and np.array_equal(s2, s3)
        st = G.status()
        assert st["overflow_batches"] == 0 and st["queries"] > 0
    finally:
        G.close()
