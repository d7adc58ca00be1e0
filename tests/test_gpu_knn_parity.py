"""GPU: the HIP kNN path (through the C ABI) against the oracle and the golden vectors."""
import os

import numpy as np
import pytest

import oracle
from isehr_amd.synth import synth_rows

pytestmark = pytest.mark.gpu

# Tolerance on the cosine score scale (DESIGN.md "Parity"): normalised rows are stored in f32, so two
# rows whose exact f64 cosines differ by less than this may swap places; everything else is exact.
TAU = 1e-6


@pytest.fixture(scope="module")
def lib():
    from isehr_amd import _lib
    _lib.load()
    return _lib


@pytest.mark.parametrize("seed", [11, 12, 13])
@pytest.mark.parametrize("dt", ["float32", "float64"])
def test_matching_hip_vs_reference_golden(lib, golden_dir, seed, dt):
    """Same seeded inputs as the reference run that produced the goldens; indices must agree up to
    near-ties of the reference's own f32/f64 arithmetic."""
    from isehr_amd.nnsearch import matching_HIP
    z = np.load(os.path.join(golden_dir, "matching_l2.npz"))
    tag = f"s{seed}_{dt}"
    _, n, d, nq, k = (int(v) for v in z[tag + "_meta"])
    g = synth_rows(seed, 0, n, d, np.dtype(dt))
    q = synth_rows(seed + 1000, 0, nq, d, np.dtype(dt))
    idx, tpq = matching_HIP(k, g, q)
    assert idx.shape == (nq, k) and idx.dtype == np.int64 and tpq > 0
    s = oracle.exact_scores_f64(g, q)
    assert oracle.check_topk_parity(idx, s, k, TAU) == []
    ref = z[tag + "_idx"]
    # position-wise agreement with the reference run, outside its near-ties
    gs = np.take_along_axis(s, idx, 1)
    rs = np.take_along_axis(s, ref, 1)
    assert np.abs(gs - rs).max() <= TAU
    agree = (idx == ref).mean()
    assert agree > 0.99, agree
    if dt == "float64":
        assert np.array_equal(idx, ref)


@pytest.mark.parametrize("layout", ["rowmajor", "transposed_view", "f64_transposed"])
def test_strided_inputs_like_the_callers(lib, layout):
    """Callers hand over `.T` views of [D,N] arrays (src/offline.py:107, src/online.py:96,132)."""
    from isehr_amd.nnsearch import matching_HIP
    n, d, nq, k = 3000, 192, 9, 50
    g = synth_rows(3, 0, n, d)
    q = synth_rows(4, 0, nq, d)
    if layout == "rowmajor":
        G, Q = g, q
    elif layout == "transposed_view":
        G, Q = np.ascontiguousarray(g.T).T, np.ascontiguousarray(q.T).T
    else:
        G, Q = np.ascontiguousarray(g.T.astype(np.float64)).T, np.ascontiguousarray(q.T.astype(np.float64)).T
    idx, _ = matching_HIP(k, G, Q)
    s = oracle.exact_scores_f64(g, q)
    assert oracle.check_topk_parity(idx, s, k, TAU) == []
    # position by position against the float64 ranking: a position may hold a different row only if the two rows'
    # exact scores are within the tolerance (the layouts must not change the answer at all: same ids for all three)
    ref = oracle.exact_topk_f64(g, q, k)[0]
    assert np.abs(np.take_along_axis(s, idx, 1) - np.take_along_axis(s, ref, 1)).max() <= TAU
    assert np.array_equal(idx, matching_HIP(k, g, q)[0])


def test_edge_cases_duplicates_and_tiny(lib, golden_dir):
    from isehr_amd.nnsearch import matching_HIP
    z = np.load(os.path.join(golden_dir, "matching_l2_edge.npz"))
    g, q = z["g"], z["q"]
    idx, _, sc = matching_HIP(8, g, q, return_scores=True)
    s = oracle.exact_scores_f64(g, q)
    assert oracle.check_topk_parity(idx, s, 8, TAU) == []
    # rows 3, 7 (exact duplicate) and 9 (same direction) tie at the top of query 0: all three returned first
    assert set(idx[0, :3]) == {3, 7, 9}
    assert set(z["idx"][0, :3]) == {3, 7, 9}
    # zero gallery row: NaN in the reference, ranked last -> never inside a top-K < N
    idxz, _ = matching_HIP(63, z["gz"], q)
    assert not (idxz == 20).any()
    assert np.array_equal(np.sort(idxz, 1), np.sort(z["idxz"][:, :63], 1))
    # K == N on a tiny gallery
    idxa, _ = matching_HIP(64, g, q)
    assert oracle.check_topk_parity(idxa, s, 64, TAU) == []


def test_k_larger_than_gallery_raises(lib):
    from isehr_amd.nnsearch import matching_HIP
    g = synth_rows(1, 0, 10, 32)
    with pytest.raises(RuntimeError):
        matching_HIP(11, g, g[:2])


def test_bf16_path_equals_forced_exact_path(lib):
    """The MFMA bf16 pass + certificate must return exactly what the f32 scorer returns."""
    from isehr_amd._lib import Gallery
    n, d, nq, k = 20000, 256, 33, 100
    g = synth_rows(21, 0, n, d)
    q = synth_rows(22, 0, nq, d)
    G = Gallery.from_host(g)
    i1, s1, _ = G.search(q, k)
    G.set_option("force_exact", 1)
    i2, s2, _ = G.search(q, k)
    st = G.status()
    G.close()
    assert np.array_equal(i1, i2)
    assert np.array_equal(s1, s2)
    assert st["overflow_batches"] == 0
    s = oracle.exact_scores_f64(g, q)
    assert oracle.check_topk_parity(i1, s, k, TAU) == []
    # returned scores are the exact cosine to f32 rounding
    assert np.abs(np.take_along_axis(s, i1, 1) - s1).max() < 2e-7


def test_synth_device_matches_host(lib):
    import torch
    from isehr_amd import _lib
    t = torch.empty((37, 96), dtype=torch.float32, device="cuda")
    _lib.synth_fill_device(t.data_ptr(), 1234, 5, 37, 96, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(t.cpu().numpy(), synth_rows(1234, 5, 37, 96))


def test_massive_ties_fall_through_to_the_dense_path(lib):
    """3000 exact duplicates of the best match: more rows within the error margin of the K-th score than the
    candidate buffers hold.  The host API must still answer (dense path: ties to the lower index)."""
    from isehr_amd._lib import Gallery
    n, d, k = 20000, 128, 100
    g = synth_rows(61, 0, n, d)
    dup = np.arange(500, 3500)
    g[dup] = g[17]
    q = np.stack([g[17], synth_rows(62, 0, 1, d)[0]])
    G = Gallery.from_host(g)
    idx, sc, _ = G.search(q, k)
    st = G.status()
    G.close()
    assert st["overflow_batches"] >= 1                     # the filter paths did overflow
    expect = np.sort(np.concatenate([[17], dup]))[:k]
    assert np.array_equal(idx[0], expect)
    assert np.abs(sc[0] - 1.0).max() < 1e-6
    s = oracle.exact_scores_f64(g, q)
    assert oracle.check_topk_parity(idx, s, k, TAU) == []


@pytest.mark.parametrize("n,d,nq,k", [(1, 1, 1, 1), (3, 5, 2, 3), (257, 100, 3, 17), (1000, 33, 1, 1000),
                                      (513, 2050, 2, 5), (5000, 64, 1500, 10)])
def test_ragged_shapes(lib, n, d, nq, k):
    """Rows / columns / queries that are not multiples of the tile sizes, D > 2048, Q > one batch of 1024,
    K = N, single-row and single-column galleries."""
    from isehr_amd.nnsearch import matching_HIP
    g = synth_rows(100 + n, 0, n, d) + 0.25
    q = synth_rows(200 + n, 0, nq, d) + 0.25
    idx, _, sc = matching_HIP(k, g, q, return_scores=True)
    assert idx.shape == (nq, k)
    s = oracle.exact_scores_f64(g, q)
    assert oracle.check_topk_parity(idx, s, k, TAU) == []
    assert np.abs(np.take_along_axis(s, idx, 1) - sc).max() < 3e-7


def test_degenerate_queries(lib):
    """A zero query normalises to NaN in the reference (its ranking is then arbitrary); here the call must not
    fail or disturb the other queries of the batch.  Huge magnitudes must not break the 16-bit image."""
    from isehr_amd.nnsearch import matching_HIP
    g = synth_rows(7, 0, 3000, 96)
    q = synth_rows(8, 0, 4, 96)
    q[1] = 0.0
    q[2] *= 1e30
    g[5] *= 1e-30
    with np.errstate(all="ignore"):
        idx, _ = matching_HIP(20, g, q)
    qq = q.astype(np.float64)
    qq[2] /= 1e30
    s = oracle.exact_scores_f64(g.astype(np.float64) * np.where(np.arange(3000) == 5, 1e30, 1.0)[:, None], qq[[0, 2, 3]])
    assert oracle.check_topk_parity(idx[[0, 2, 3]], s, 20, 1e-5) == []


def test_unnormalised_gallery_with_large_values_uses_bf16_image(lib):
    """MI_NORM_NONE rows far outside fp16's comfortable range: the image falls back to bf16 and results stay exact."""
    from isehr_amd.nnsearch import ip_topk_hip
    vecs = (synth_rows(9, 0, 2000, 80) * 3000.0).T.copy()
    qv = (synth_rows(10, 0, 5, 80) * 2.0).T.copy()
    ranks, scores = ip_topk_hip(vecs, qv, 30)
    s64 = (vecs.astype(np.float64).T @ qv.astype(np.float64)).T
    assert oracle.check_topk_parity(ranks.T, s64, 30, 1e-6 * float(np.abs(s64).max())) == []


@pytest.mark.parametrize("ndup,device_repair,expect_fallback",
                         [(0, 1, False), (30, 1, False), (90, 1, True), (0, -1, False), (30, -1, True), (30, 0, True)])
def test_speculative_threshold_verification_and_repair(lib, ndup, device_repair, expect_fallback):
    """The single-launch schedule uses a speculative threshold taken from the bootstrap sample (32 tiles of 256 rows
    worth of rows drawn evenly from the gallery) and verifies it afterwards.  Near-duplicates of the query planted INSIDE the sample make that threshold too high:
    30 of them -> the device-side repair pass (looser threshold) must fix the query; 90 of them -> the repair fails
    too and the host API falls back to the rigorous schedule.  Batches of <= 128 queries (this one has 6) launch no repair
    pass by default (option "device_repair" = -1): there a failed verification goes straight to the fallback.  Results
    must be exact in every case."""
    from isehr_amd._lib import Gallery
    n, d, nq, k = 200000, 64, 6, 100
    g = synth_rows(71, 0, n, d)
    q = synth_rows(72, 0, nq, d)
    rng = np.random.default_rng(3)
    sample_rows = lib.sample_source_rows(n)                 # one hashed draw per stratum of n / 8192 rows
    assert len(set(sample_rows)) == 8192 and sample_rows.min() >= 0 and sample_rows.max() < n
    rows = rng.choice(sample_rows, size=ndup, replace=False)
    for j, r in enumerate(rows):
        g[r] = q[0] * (1.0 + 0.01 * j) + 0.02 * synth_rows(73, j, 1, d)[0]
    G = Gallery.from_host(g)
    G.set_option("chunk0_tiles", 32)                        # the 8192-row sample the planted rows were drawn from
    G.set_option("device_repair", device_repair)
    assert G.get_option("sample_rows") == 8192 and G.get_option("device_repair") == device_repair
    idx, sc, _ = G.search(q, k)
    st = G.status()
    G.close()
    s = oracle.exact_scores_f64(g, q)
    assert oracle.check_topk_parity(idx, s, k, TAU) == []
    assert set(rows) <= set(idx[0])
    assert (st["spec_retries"] >= 1) == expect_fallback and st["overflow_batches"] == 0


@pytest.mark.parametrize("nq", [1, 70])
def test_small_batch_spec_failure_is_repaired_on_the_asynchronous_entry_points(lib, nq):
    """a failed speculative threshold in a batch of <= 128 queries used to leave the flagged query's
    output row empty (-1 / -inf) on the device / phase entry points, whose callers -- ShardedGallery.search(verify=False),
    search_stream, the alpha-QE re-search -- never read the sticky flag.  On those entry points the workgroup of a failed query
    now repairs it INSIDE the maintain launch (a scan of the shard's stored rows; no repair launches behind batches of <= 128
    queries) and counts it in mi_search_stats.inkernel_repairs, the only trace of why that call took a scan longer: the
    reference's own batch shapes (1 and 70 queries) with 30 near-duplicates of query 0 planted inside the threshold sample come
    back complete and exact without any host fallback, through the one-call device search and through the phase API (phase 1 ->
    K-th of the gathered lists -> phase 2 -> merge, as the sharded protocol runs it)."""
    import torch
    from isehr_amd import _lib
    from isehr_amd._lib import Gallery
    from isehr_amd.sharded import ShardedGallery
    n, d, k = 200000, 64, 100
    g = synth_rows(71, 0, n, d)
    q = synth_rows(72, 0, nq, d)
    rng = np.random.default_rng(3)
    rows = rng.choice(lib.sample_source_rows(n), size=30, replace=False)
    for j, r in enumerate(rows):
        g[r] = q[0] * (1.0 + 0.01 * j) + 0.02 * synth_rows(73, j, 1, d)[0]
    s = oracle.exact_scores_f64(g, q)
    G = Gallery.from_host(g)
    try:
        G.set_option("chunk0_tiles", 32)
        assert G.get_option("device_repair") == -1
        # the host entry point of the same batch does fall back (no repair launches there): the failure is real
        idx_h, sc_h, _ = G.search(q, k)
        st_h = G.status(reset=True)
        assert st_h["spec_retries"] >= 1 and st_h["inkernel_repairs"] == 0
        qd = torch.from_numpy(q).cuda()
        stream = torch.cuda.current_stream().cuda_stream
        sg = ShardedGallery(G)
        idx, sc = sg.search(qd, k)                              # verify=False: nobody reads the flags
        torch.cuda.synchronize()
        idx, sc = idx.cpu().numpy(), sc.cpu().numpy()
        assert G.flags() == 0
        assert oracle.check_topk_parity(idx, s, k, TAU) == [] and set(rows) <= set(idx[0])
        assert np.array_equal(idx, idx_h) and np.array_equal(sc, sc_h)
        assert G.status()["inkernel_repairs"] >= 1          # (status does not reset: the phase API below adds its own)
        # phase API, one shard
        approx = torch.empty((nq, k), dtype=torch.float32, device="cuda")
        L = torch.empty((nq,), dtype=torch.float32, device="cuda")
        pack = torch.empty((2, nq, k), dtype=torch.int64, device="cuda")
        scp = torch.empty((nq, k), dtype=torch.float32, device="cuda")
        oidx = torch.empty((nq, k), dtype=torch.int64, device="cuda")
        osc = torch.empty((nq, k), dtype=torch.float32, device="cuda")
        G.phase1_device(qd.data_ptr(), nq, k, approx.data_ptr(), stream)
        _lib.kth_of_gathered_device(approx.data_ptr(), 1, nq, k, L.data_ptr(), stream)
        G.phase2_device(nq, k, L.data_ptr(), pack[1].data_ptr(), scp.data_ptr(), pack[0].data_ptr(), stream)
        _lib.topk_merge_strided_device(pack[0].data_ptr(), pack[1].data_ptr(), 2 * nq * k, 1, nq, k, oidx.data_ptr(),
                                       osc.data_ptr(), stream)
        torch.cuda.synchronize()
        assert G.flags() == 0
        assert np.array_equal(oidx.cpu().numpy(), idx_h) and np.array_equal(osc.cpu().numpy(), sc_h)
        st = G.status()
        assert st["spec_retries"] == 0 and st["overflow_batches"] == 0 and st["inkernel_repairs"] >= 2
    finally:
        G.close()


@pytest.mark.parametrize("nq", [1, 16, 17, 64, 65, 70, 80, 81, 96, 97, 112, 113, 128, 129])
def test_small_batch_kernel_equals_tile_kernel(lib, nq):
    """Batches of <= 128 queries are scored by the HBM-bound kernel of stream_select.hip (4 or 8 blocks of 16
    queries); it must produce the answers of the 256 x 256-tile kernel bit for bit, and the exact top-K."""
    from isehr_amd._lib import Gallery
    n, d, k = 70001, 320, 50                                # ragged last tile, dp = 320 (10 K-slices)
    g = synth_rows(91, 0, n, d)
    q = synth_rows(92, 0, nq, d)
    G = Gallery.from_host(g)
    idx1, sc1, _ = G.search(q, k)
    st1 = G.status()
    G.set_option("small_batch_kernel", 0)
    idx0, sc0, _ = G.search(q, k)
    G.close()
    assert np.array_equal(idx0, idx1) and np.array_equal(sc0, sc1)
    assert st1["overflow_batches"] == 0
    s = oracle.exact_scores_f64(g, q)
    assert oracle.check_topk_parity(idx1, s, k, TAU) == []


@pytest.mark.parametrize("nq", [1, 16, 70, 128, 300])
def test_ksplit_bootstrap_gives_the_same_answers_and_needs_no_fallback(lib, nq):
    """Small batches: the bootstrap launch on the sample splits K over several workgroups per (tile, query group) and ADDS
    its partial scores (option "boot_ksplit", kernels.h ScoreArgs::ksplit) -- the sample scores then come from another
    summation order than the scoring launch's, which is why the threshold carries half a margin more: with 300 exact
    duplicates at the top (every one of them AT the speculative threshold) no query may lose them to an ulp and fall back.
    Same answers with the option off; 300 queries (5 groups: no split) take the one-workgroup form either way."""
    from isehr_amd._lib import Gallery
    n, d, k = 60000, 512, 100
    g = synth_rows(95, 0, n, d)
    g[2000:2300] = g[11]                                    # 300 exact ties of query 0's best row
    q = np.concatenate([g[11:12], synth_rows(96, 0, max(1, nq - 1), d)])[:nq]
    G = Gallery.from_host(g)
    try:
        assert G.get_option("boot_ksplit") == 1
        for rep in range(3):                                # the order of the atomic adds differs from run to run
            idx1, sc1, _ = G.search(q, k)
            assert G.status(reset=True)["overflow_batches"] == 0
        G.set_option("boot_ksplit", 0)
        idx0, sc0, _ = G.search(q, k)
        assert G.status()["overflow_batches"] == 0
    finally:
        G.close()
    assert np.array_equal(idx0, idx1) and np.array_equal(sc0, sc1)
    assert list(idx1[0][:3]) == [11, 2000, 2001]
    assert oracle.check_topk_parity(idx1, oracle.exact_scores_f64(g, q), k, TAU) == []


def test_ordered_gallery_keeps_the_fast_path(lib):
    """The reference's 1M gallery is [rOxford | distractors] (src/test_rOP1m.py:136-139): every true positive of a
    query sits in the first rows.  The bootstrap sample is drawn evenly from the whole shard, so that such an ordering
    neither pushes the speculative threshold up nor costs a fallback; the answers are exact either way."""
    from isehr_amd._lib import Gallery
    n, d, nq, k = 150000, 128, 12, 100
    g = synth_rows(75, 0, n, d)
    q = synth_rows(76, 0, nq, d)
    for qi in range(nq):                                    # 150 positives per query, all inside rows 0..1799
        for j in range(150):
            g[qi * 150 + j] = q[qi] + (0.3 + 0.004 * j) * synth_rows(77, qi * 150 + j, 1, d)[0]
    G = Gallery.from_host(g)
    idx, sc, _ = G.search(q, k)
    st = G.status()
    G.close()
    s = oracle.exact_scores_f64(g, q)
    assert oracle.check_topk_parity(idx, s, k, TAU) == []
    assert all(set(idx[qi]) <= set(range(qi * 150, qi * 150 + 150)) for qi in range(nq))
    assert st["overflow_batches"] == 0
    assert st["survivors"] / st["queries"] < 4096           # the filter stayed selective


def test_bf16_image_path_gives_the_same_answers(lib, golden_dir):
    """The image element type only changes the width of the error margin, never the result."""
    from isehr_amd import _lib
    n, d, nq, k = 30000, 256, 40, 100
    g = synth_rows(81, 0, n, d)
    q = synth_rows(82, 0, nq, d)
    res = {}
    try:
        for name, flag in (("f16", 1), ("bf16", 0)):
            _lib.set_global_option("image_dtype", flag)
            G = _lib.Gallery.from_host(g)
            idx, sc, _ = G.search(q, k)
            st = G.status()
            G.close()
            res[name] = (idx, sc, st["candidates"] / st["queries"])
    finally:
        _lib.set_global_option("image_dtype", 1)
    assert np.array_equal(res["f16"][0], res["bf16"][0]) and np.array_equal(res["f16"][1], res["bf16"][1])
    assert res["f16"][2] < res["bf16"][2]            # tighter certificate -> fewer rows re-scored
    s = oracle.exact_scores_f64(g, q)
    assert oracle.check_topk_parity(res["bf16"][0], s, k, TAU) == []


@pytest.mark.parametrize("tag", ["a", "b"])
@pytest.mark.parametrize("dt", ["float32", "float64"])
def test_matching_fractional_dis_vs_reference_golden(lib, golden_dir, tag, dt):
    """matching_fractional_dis (src/utils/nnsearch.py:709-731, p = 2): same ordering as matching_L2, and the reference's
    return shape [min(Q, K), K] (it slices the query axis by K)."""
    from isehr_amd.nnsearch import matching_fractional_dis_hip
    z = np.load(os.path.join(golden_dir, "fractional.npz"))
    seed, n, d, nq, k = (int(v) for v in z[f"{tag}_{dt}_meta"])
    g = synth_rows(seed, 0, n, d, np.dtype(dt))
    q = synth_rows(seed + 1000, 0, nq, d, np.dtype(dt))
    idx, tpq = matching_fractional_dis_hip(k, g, q)
    ref = z[f"{tag}_{dt}_idx"]
    assert idx.shape == ref.shape == (min(nq, k), k) and idx.dtype == np.int64 and tpq > 0
    s = oracle.exact_scores_f64(g, q[:k])
    assert oracle.check_topk_parity(idx, s, k, TAU) == []
    assert np.abs(np.take_along_axis(s, idx, 1) - np.take_along_axis(s, ref, 1)).max() <= TAU
    if dt == "float64":
        assert np.array_equal(idx, ref)


def test_raised_caps_answer_massive_ties_without_the_dense_path(lib):
    """MI_ERR_OVERFLOW tells the user to raise survivor_cap / rescore_cap: at their maxima (16384 / 8192; the maintain
    kernel keeps 8192 keys in LDS and reads the rest from global memory, the emit kernel ranks in tiles of 512) 3000
    exact duplicates of the best match fit the candidate buffers, so the filter path itself answers -- with the ties in
    index order -- and nothing overflows."""
    from isehr_amd._lib import Gallery
    n, d, k = 20000, 128, 100
    g = synth_rows(61, 0, n, d)
    dup = np.arange(500, 3500)
    g[dup] = g[17]
    q = np.stack([g[17], synth_rows(62, 0, 1, d)[0]])
    G = Gallery.from_host(g)
    try:
        G.set_option("survivor_cap", 16384)
        G.set_option("rescore_cap", 8192)
        G.set_option("exact_fallback", 0)                      # an overflow would now be an error, not a fallback
        idx, sc, _ = G.search(q, k)
        assert G.status()["overflow_batches"] == 0
    finally:
        G.close()
    expect = np.sort(np.concatenate([[17], dup]))[:k]
    assert np.array_equal(idx[0], expect)
    s = oracle.exact_scores_f64(g, q)
    assert oracle.check_topk_parity(idx, s, k, TAU) == []


def test_more_survivors_than_the_maintain_kernel_keeps_in_lds(lib):
    """The final maintain launch keeps the first 8192 survivor keys of a query in LDS and reads the keys beyond from the
    entries in global memory (csrc/select.hip MAINT_LDS_KEYS).  9000 gallery rows at graded similarity 0.5 .. 0.9 to the
    query, none of them among the 8192 rows of the bootstrap sample (so the speculative threshold stays where random 64-d
    rows put it, near 0.37), all pass the filter: the select runs over ~9500 keys with survivor_cap = 16384, while only
    ~100 of them are candidates.  The answer must be the exact one and nothing may overflow."""
    from isehr_amd._lib import Gallery
    n, d, k = 60000, 64, 100
    g = synth_rows(81, 0, n, d)
    q = synth_rows(82, 0, 2, d)
    qn = q[0] / np.linalg.norm(q[0])
    rng = np.random.default_rng(9)
    outside = np.setdiff1d(np.arange(n), lib.sample_source_rows(n))
    rows = rng.choice(outside, size=9000, replace=False)
    cosv = np.linspace(0.5, 0.9, 9000)
    noise = synth_rows(83, 0, 9000, d).astype(np.float64)
    noise -= (noise @ qn)[:, None] * qn[None, :]
    noise /= np.linalg.norm(noise, axis=1, keepdims=True)
    g[rows] = (cosv[:, None] * qn[None, :] + np.sqrt(1 - cosv ** 2)[:, None] * noise).astype(np.float32) * 1.7
    G = Gallery.from_host(g)
    try:
        G.set_option("survivor_cap", 16384)
        G.set_option("exact_fallback", 0)                      # an overflow would be an error, not a silent fallback
        G.status(reset=True)
        idx, sc, _ = G.search(q, k)
        st = G.status()
        assert st["overflow_batches"] == 0
        assert st["survivors"] / st["queries"] > 8192 / 2      # query 0 alone has > 9000 (query 1: the usual few hundred)
    finally:
        G.close()
    assert set(idx[0]) == set(rows[-k:])                       # the hundred most similar planted rows
    s = oracle.exact_scores_f64(g, q)
    assert oracle.check_topk_parity(idx, s, k, TAU) == []
