"""GPU: BASELINE.json's full size (rOxford5k + 1M distractors: N = 1,005,994 x D = 2048, K = 100): against the oracle itself
for 8 queries (every stored row scored in float64 on the host, tests/_fullsize.py) and through size-independent properties
for whole batches.  Gallery rows are generated on the device."""
import numpy as np
import pytest

from isehr_amd.synth import synth_rows
from _fullsize import assert_oracle_parity, host_f64_scores_and_topk, stored_rows_are_the_normalised_raw_rows

pytestmark = pytest.mark.gpu
N, D, K = 1005994, 2048, 100


@pytest.fixture(scope="module")
def big():
    import torch
    from isehr_amd import _lib
    dev = torch.device("cuda", 0)
    s = torch.cuda.current_stream().cuda_stream
    raw = torch.empty((N, D), dtype=torch.float32, device=dev)
    _lib.synth_fill_device(raw.data_ptr(), 1234, 0, N, D, s)
    # plant: gallery rows 7, 500000 and N-1 are (scaled) copies of query rows 0, 1, 2
    q = torch.from_numpy(synth_rows(4321, 0, 1024, D)).to(dev)
    raw[7] = q[0] * 3.0
    raw[500000] = q[1] * 0.5
    raw[N - 1] = q[2]
    torch.cuda.synchronize()
    g = _lib.Gallery.from_device_ptr(raw.data_ptr(), N, D)
    half = (N + 1) // 2
    g0 = _lib.Gallery.from_device_ptr(raw.data_ptr(), half, D, row_offset=0)
    g1 = _lib.Gallery.from_device_ptr(raw[half:].data_ptr(), N - half, D, row_offset=half)
    del raw
    torch.cuda.empty_cache()
    yield g, g0, g1, q
    for h in (g, g0, g1):
        h.close()


def _search(g, q, nq):
    import torch
    idx = torch.empty((nq, K), dtype=torch.int64, device=q.device)
    sc = torch.empty((nq, K), dtype=torch.float32, device=q.device)
    g.search_device(q.data_ptr(), nq, K, idx.data_ptr(), sc.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return idx.cpu().numpy(), sc.cpu().numpy()


def test_full_size_against_the_oracle_itself(big):
    """8 queries x 1 005 994 rows scored in float64 ON THE HOST in 64 k-row chunks of get_rows() (numpy:
    oracle.exact_scores_f64 / exact_topk_f64 per chunk, oracle.merge_topk over the chunks) -- the ranking matching_L2 computes
    (src/utils/nnsearch.py:699-703) -- and the library's answers judged by oracle.check_topk_parity at 1e-6: the 8 queries
    alone (streaming kernel), inside a 64-query batch, and inside the benchmarked 1024-query batch (tile kernel).  The stored
    rows themselves are checked against the reference's normalisation on two 4096-row stretches regenerated on the host."""
    g, _, _, q = big
    pick = np.array([0, 1, 2, 5, 77, 300, 640, 1023])
    qh = q.cpu().numpy()
    scores, top_i, top_s = host_f64_scores_and_topk(g, qh[pick], K)
    assert list(top_i[:3, 0]) == [7, 500000, N - 1]                  # the planted copies, found by the oracle too
    for nq in (1024, 64):
        idx, sc = _search(g, q, nq)
        sel = pick[pick < nq]
        assert_oracle_parity(idx[sel], sc[sel], scores[:len(sel)], top_i[:len(sel)], top_s[:len(sel)], K)
    idx8, sc8 = _search(g, q[pick].contiguous(), 8)
    assert_oracle_parity(idx8, sc8, scores, top_i, top_s, K)
    assert g.status()["overflow_batches"] == 0
    for row0 in (8192, N - 4096):                                    # (clear of the planted rows 7, 500000; N - 1 is planted)
        raw = synth_rows(1234, row0, 4095, D)
        stored_rows_are_the_normalised_raw_rows(g, raw, row0)


def test_full_size_properties(big):
    g, g0, g1, q = big
    idx, sc = _search(g, q, 64)
    assert g.status()["overflow_batches"] == 0
    assert (np.diff(sc, axis=1) <= 0).all()                          # sorted
    assert all(len(set(r)) == K for r in idx)                        # no duplicates
    assert idx.min() >= 0 and idx.max() < N
    # planted copies are the nearest neighbour, cosine 1 regardless of their scale
    assert idx[0, 0] == 7 and idx[1, 0] == 500000 and idx[2, 0] == N - 1
    assert np.abs(sc[:3, 0] - 1.0).max() < 1e-6
    # random 2048-d unit vectors: 100th best cosine of 1M is ~3.7 sigma = 0.082
    assert 0.07 < sc[5:, K - 1].mean() < 0.095
    # determinism / idempotence
    idx2, sc2 = _search(g, q, 64)
    assert np.array_equal(idx, idx2) and np.array_equal(sc, sc2)
    # the returned scores are exact: recompute 64 x 100 of them in float64 from the stored rows
    qn = q.cpu().numpy().astype(np.float64)
    qn /= np.linalg.norm(qn, axis=1, keepdims=True)
    for qi in (0, 9, 63):
        rows = np.stack([g.get_rows(int(r), 1)[0] for r in idx[qi, :10]]).astype(np.float64)
        assert np.abs(rows @ qn[qi] - sc[qi, :10]).max() < 3e-7


def test_full_size_filter_path_equals_exact_path(big):
    """The 16-bit MFMA filter (default image: fp16; 16 queries -> the streaming kernel) + certificate must reproduce the
    f32-scored path bit for bit."""
    g, _, _, q = big
    assert int(g.get_option("image_dtype")) == 1
    idx, sc = _search(g, q, 16)
    g.set_option("force_exact", 1)
    try:
        idx_e, sc_e = _search(g, q, 16)
    finally:
        g.set_option("force_exact", 0)
    assert np.array_equal(idx, idx_e) and np.array_equal(sc, sc_e)


def test_full_size_two_shards_equal_one(big):
    import torch
    from isehr_amd import _lib
    g, g0, g1, q = big
    nq = 64
    ref_idx, ref_sc = _search(g, q, nq)
    dev, s = q.device, torch.cuda.current_stream().cuda_stream
    approx = torch.empty((2, nq, K), dtype=torch.float32, device=dev)
    for r, sh in enumerate((g0, g1)):
        sh.phase1_device(q.data_ptr(), nq, K, approx[r].data_ptr(), s)
    L = torch.empty((nq,), dtype=torch.float32, device=dev)
    _lib.kth_of_gathered_device(approx.data_ptr(), 2, nq, K, L.data_ptr(), s)
    idx = torch.empty((2, nq, K), dtype=torch.int64, device=dev)
    sc = torch.empty((2, nq, K), dtype=torch.float32, device=dev)
    sc64 = torch.empty((2, nq, K), dtype=torch.float64, device=dev)
    for r, sh in enumerate((g0, g1)):
        sh.phase2_device(nq, K, L.data_ptr(), idx[r].data_ptr(), sc[r].data_ptr(), sc64[r].data_ptr(), s)
    oi = torch.empty((nq, K), dtype=torch.int64, device=dev)
    osc = torch.empty((nq, K), dtype=torch.float32, device=dev)
    _lib.topk_merge_device(sc64.data_ptr(), idx.data_ptr(), 2, nq, K, oi.data_ptr(), osc.data_ptr(), s)
    torch.cuda.synchronize()
    assert np.array_equal(oi.cpu().numpy(), ref_idx) and np.array_equal(osc.cpu().numpy(), ref_sc)


def test_full_size_tile_kernel_equals_exact_path_and_two_shards(big):
    """The benchmarked configuration itself: 1024 queries (the 256 x 256-tile MFMA kernel, four query tiles, 64 K-slices,
    speculative threshold) on all 1,005,994 rows -- bit-equal to the f32-scored path and to the two-shard protocol."""
    import torch
    from isehr_amd import _lib
    g, g0, g1, q = big
    nq = 1024
    idx, sc = _search(g, q, nq)
    assert g.status()["overflow_batches"] == 0
    assert idx[0, 0] == 7 and idx[1, 0] == 500000 and idx[2, 0] == N - 1
    assert (np.diff(sc, axis=1) <= 0).all() and all(len(set(r)) == K for r in idx)
    g.set_option("force_exact", 1)
    try:
        idx_e, sc_e = _search(g, q, nq)
    finally:
        g.set_option("force_exact", 0)
    assert np.array_equal(idx, idx_e) and np.array_equal(sc, sc_e)
    # float64 re-computation of 16 queries' 100 scores from the stored rows
    qn = q.cpu().numpy().astype(np.float64)
    qn /= np.linalg.norm(qn, axis=1, keepdims=True)
    for qi in range(0, nq, 64):
        rows = np.stack([g.get_rows(int(r), 1)[0] for r in idx[qi]]).astype(np.float64)
        assert np.abs(rows @ qn[qi] - sc[qi]).max() < 3e-7
    # two shards, phase protocol
    dev, s = q.device, torch.cuda.current_stream().cuda_stream
    approx = torch.empty((2, nq, K), dtype=torch.float32, device=dev)
    for r, sh in enumerate((g0, g1)):
        sh.phase1_device(q.data_ptr(), nq, K, approx[r].data_ptr(), s)
    L = torch.empty((nq,), dtype=torch.float32, device=dev)
    _lib.kth_of_gathered_device(approx.data_ptr(), 2, nq, K, L.data_ptr(), s)
    idx2 = torch.empty((2, nq, K), dtype=torch.int64, device=dev)
    sc2 = torch.empty((2, nq, K), dtype=torch.float32, device=dev)
    sc64 = torch.empty((2, nq, K), dtype=torch.float64, device=dev)
    for r, sh in enumerate((g0, g1)):
        sh.phase2_device(nq, K, L.data_ptr(), idx2[r].data_ptr(), sc2[r].data_ptr(), sc64[r].data_ptr(), s)
    oi = torch.empty((nq, K), dtype=torch.int64, device=dev)
    osc = torch.empty((nq, K), dtype=torch.float32, device=dev)
    _lib.topk_merge_device(sc64.data_ptr(), idx2.data_ptr(), 2, nq, K, oi.data_ptr(), osc.data_ptr(), s)
    torch.cuda.synchronize()
    assert g0.status()["overflow_batches"] == 0 and g1.status()["overflow_batches"] == 0
    assert np.array_equal(oi.cpu().numpy(), idx) and np.array_equal(osc.cpu().numpy(), sc)


def test_full_size_bf16_image_equals_exact_path(big):
    """The same gallery re-imaged as bf16 (`mi_gallery_set_image_dtype(0)`: 8-bit significand, ~3x the candidates) at
    1,005,994 x 2048: tile kernel (1024 queries) and streaming kernel (70 queries) against the f32-scored path."""
    g, _, _, q = big
    g.set_image_dtype(0)
    try:
        assert int(g.get_option("image_dtype")) == 0
        for nq in (1024, 70):
            g.status(reset=True)
            idx, sc = _search(g, q, nq)
            st = g.status()
            assert st["overflow_batches"] == 0 and g.flags() == 0
            assert st["candidates"] / st["queries"] > 200            # bf16 margins: ~358 candidate rows per query, fp16 ~127
            assert idx[0, 0] == 7 and idx[1, 0] == 500000 and idx[2, 0] == N - 1
            g.set_option("force_exact", 1)
            try:
                idx_e, sc_e = _search(g, q, nq)
            finally:
                g.set_option("force_exact", 0)
            assert np.array_equal(idx, idx_e) and np.array_equal(sc, sc_e)
    finally:
        g.set_image_dtype(1)
    idx, sc = _search(g, q, 64)                                       # back on fp16: same answers as before
    assert int(g.get_option("image_dtype")) == 1 and np.array_equal(idx, idx_e[:64]) and np.array_equal(sc, sc_e[:64])


def test_full_size_answers_equal_the_dense_float64_search(big):
    """Completeness by an INDEPENDENT path (missing #3): `mi_knn_dense64_search` computes every one of the
    1,005,994 scores of a query in float64 and selects the exact top-100 of the dense row -- no sample, no threshold, no
    survivor or candidate buffers.  A row that the filter path AND its f32-scored twin both dropped would show here.  The
    tile kernel's answer (16 queries of the 1024-query batch, the planted ones among them) and the streaming kernel's (a
    16-query batch) must be those indices; scores agree to the last float32 bit or its neighbour (two float64 summation orders)."""
    g, _, _, q = big
    qh = q.cpu().numpy()
    pick = np.r_[0:6, np.arange(100, 1024, 100)][:16]
    didx, dsc, dsc64, _ = g.dense64_search(qh[pick], K)
    assert didx[0, 0] == 7 and didx[1, 0] == 500000 and didx[2, 0] == N - 1
    idx, sc = _search(g, q, 1024)                                     # tile kernel, speculative single launch
    assert np.array_equal(idx[pick], didx)
    assert np.abs(sc[pick] - dsc).max() <= 6e-8 and np.abs(sc[pick].astype(np.float64) - dsc64).max() < 6e-8
    didx16, dsc16, _, _ = g.dense64_search(qh[:16], K)
    idx16, sc16 = _search(g, q, 16)                                   # streaming kernel
    assert np.array_equal(idx16, didx16) and np.abs(sc16 - dsc16).max() <= 6e-8


def test_full_size_every_mode_gives_the_same_answers(big):
    """The benchmarked gallery through every mode a stream of batches can run in -- synchronous tail / deferred tail, after a
    calibration, batches of 1024 / 700 / 129 queries in one stream -- against the plain one-call-at-a-time answers, bit for
    bit."""
    import torch
    g, _, _, q = big
    dev, s = q.device, torch.cuda.current_stream().cuda_stream
    batches = [q[:1024], q[100:800], q[200:329], q[:1024], q[300:1000]]
    ref = []
    for b in batches:
        i_, s_ = _search(g, b.contiguous(), b.shape[0])
        ref.append((i_, s_))
    qs = [b.contiguous() for b in batches]
    g.calibrate(4, s)
    for tail in (0, 3):
        g.set_option("async_tail", tail)
        try:
            outs = [(torch.empty((b.shape[0], K), dtype=torch.int64, device=dev),
                     torch.empty((b.shape[0], K), dtype=torch.float32, device=dev)) for b in qs]
            for rep in range(2):
                for b, (oi, os_) in zip(qs, outs):
                    g.search_device(b.data_ptr(), b.shape[0], K, oi.data_ptr(), os_.data_ptr(), None, s)
            g.join(s)
            torch.cuda.synchronize()
            assert g.flags() == 0
            for (ri, rs), (oi, os_) in zip(ref, outs):
                assert np.array_equal(oi.cpu().numpy(), ri) and np.array_equal(os_.cpu().numpy(), rs), tail
        finally:
            g.set_option("async_tail", 0)
