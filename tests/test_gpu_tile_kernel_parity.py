"""GPU: the 256 x 256-tile MFMA scoring kernel (gemm_select.hip) at the shape bench.py times -- 1024 queries
(4 query tiles), D = 2048 (64 K-slices), speculative single-launch schedule, XCD walk -- against the float64
oracle.  Batches of <= 128 queries take the HBM-bound kernel of stream_select.hip, so every test here uses 1024."""
import numpy as np
import pytest

import oracle
from isehr_amd.synth import synth_rows

pytestmark = pytest.mark.gpu
TAU = 1e-6          # cosine scale; stored rows are f32 (|ds| <= 2.4e-7), DESIGN.md "Parity definition"
Q, D, K = 1024, 2048, 100


@pytest.fixture(scope="module")
def lib():
    from isehr_amd import _lib
    _lib.load()
    return _lib


def _device_rows(lib, seed, n, d):
    import torch
    t = torch.empty((n, d), dtype=torch.float32, device="cuda")
    lib.synth_fill_device(t.data_ptr(), seed, 0, n, d, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return t


def _search(g, q, k):
    import torch
    nq = q.shape[0]
    idx = torch.empty((nq, k), dtype=torch.int64, device=q.device)
    sc = torch.empty((nq, k), dtype=torch.float32, device=q.device)
    g.search_device(q.data_ptr(), nq, k, idx.data_ptr(), sc.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return idx.cpu().numpy(), sc.cpu().numpy()


@pytest.mark.parametrize("n", [131072, 100003])
@pytest.mark.parametrize("img", ["f16", "bf16"])
def test_tile_kernel_headline_shape_vs_oracle(lib, n, img):
    """Q = 1024, D = 2048, N = 131 072 (and a ragged N): every returned index against float64 ground truth
    (src/utils/nnsearch.py:699-703 computes the same ordering through ||q - g||)."""
    raw = _device_rows(lib, 31, n, D)
    q = _device_rows(lib, 32, Q, D)
    lib.set_global_option("image_dtype", 1 if img == "f16" else 0)
    try:
        g = lib.Gallery.from_device_ptr(raw.data_ptr(), n, D)
    finally:
        lib.set_global_option("image_dtype", 1)
    try:
        idx, sc = _search(g, q, K)
        st = g.status()
        assert st["overflow_batches"] == 0
        # the single-launch speculative schedule ran (no chunking): few survivors per query
        assert K <= st["survivors"] / st["queries"] < 4096
        s = oracle.exact_scores_f64(raw.cpu().numpy(), q.cpu().numpy())
        assert oracle.check_topk_parity(idx, s, K, TAU) == []
        assert np.abs(np.take_along_axis(s, idx, 1) - sc).max() < 3e-7
        # and bit for bit what the f32-scored path returns
        g.set_option("force_exact", 1)
        idx_e, sc_e = _search(g, q, K)
        assert np.array_equal(idx, idx_e) and np.array_equal(sc, sc_e)
    finally:
        g.close()


@pytest.mark.parametrize("img", ["f16", "bf16"])
def test_ragged_batches_answer_like_the_full_batch(lib, img):
    """129 .. 192 / 257 .. 448 / ... queries leave one to three of the last query tile's four 64-query columns empty (their
    waves multiply zeros; a variant that let them sit such tiles out was measured and lost: profiles/r06aa_idle_columns_ab.txt).
    The full batch against float64 ground truth, and every ragged prefix of it bit for bit the full batch's answers -- whatever
    the number of query tiles, the padding, the kernel (one-tile nt launches, the streaming kernel's neighbours)."""
    n = 100003
    raw = _device_rows(lib, 33, n, D)
    q = _device_rows(lib, 34, Q, D)
    lib.set_global_option("image_dtype", 1 if img == "f16" else 0)
    try:
        g = lib.Gallery.from_device_ptr(raw.data_ptr(), n, D)
    finally:
        lib.set_global_option("image_dtype", 1)
    try:
        full_i, full_s = _search(g, q, K)                                 # 1024 queries: no padding anywhere
        s = oracle.exact_scores_f64(raw.cpu().numpy(), q.cpu().numpy())
        assert oracle.check_topk_parity(full_i, s, K, TAU) == []
        for nq in (129, 150, 192, 193, 257, 300, 320, 321, 384, 448, 449, 513, 600, 769, 832, 1000):
            g.status(reset=True)
            idx, sc = _search(g, q[:nq].contiguous(), K)
            assert g.flags() == 0 and g.status()["overflow_batches"] == 0, nq
            assert np.array_equal(idx, full_i[:nq]) and np.array_equal(sc, full_s[:nq]), nq
    finally:
        g.close()


def test_laboratory_switches_are_not_in_the_product(lib):
    """Round 5: the diagnostic / A-B instantiations of the tile kernel (stages switched off, other issue orders, DMA cache
    policies, the paired-XCD walk, the first structure) are compiled into scripts/kbench.hip's own program only
    (-DMI_KBENCH), which also checks that they emit the same records.  The library holds six instantiations and the options
    that selected the others -- or variants measured to lose: the one-launch small tail, the lookahead, a second ladder level,
    in-kernel repair of large batches -- are errors."""
    raw = _device_rows(lib, 41, 5000, 64)
    g = lib.Gallery.from_device_ptr(raw.data_ptr(), 5000, 64)
    try:
        for name, value in (("debug", 4), ("kernel_variant", 1), ("small_tail", 1), ("stream_lookahead", 1),
                            ("inkernel_repair_max", 1024), ("ladder", 2)):
            with pytest.raises(RuntimeError):
                g.set_option(name, value)
        for name in ("debug", "kernel_variant", "small_tail", "stream_lookahead", "inkernel_repair_max"):
            with pytest.raises(RuntimeError):
                g.get_option(name)
        assert g.get_option("ladder") == 1
        assert not hasattr(lib.load(), "mi_knn_set_lookahead") and not hasattr(g, "set_lookahead")
    finally:
        g.close()


@pytest.mark.parametrize("img", ["f16", "bf16"])
def test_large_shard_chunk_schedule_equals_exact_path(lib, img):
    """A shard of more than 160 x 8192 rows (the per-GPU shard of BASELINE configs[3], 10M rows over 8 GPUs, has
    1.25 M) is too large for the 8192-row sample.  Round 6: it gets a hashed sample of 24 576 rows and the single filtered
    launch of every other shard; with `spec_max_ratio` lowered the geometric chunk schedule of csrc/api_schedule.hip
    phase1_batch (rigorous thresholds from the rows seen so far, then one speculative launch for the rest) answers it, as it
    does shards beyond 160 x 24 576 = 3.93 M rows.  All must equal the f32-scored path bit for bit; bf16 is the image type configs[3] names."""
    n, d, nq, k = 1600000, 256, 300, 100
    raw = _device_rows(lib, 51, n, d)
    q = _device_rows(lib, 52, nq, d)
    import torch
    raw[1234567] = q[5] * 2.0                                   # a planted exact match deep inside the shard
    torch.cuda.synchronize()
    lib.set_global_option("image_dtype", 1 if img == "f16" else 0)
    try:
        g = lib.Gallery.from_device_ptr(raw.data_ptr(), n, d)
    finally:
        lib.set_global_option("image_dtype", 1)
    del raw
    try:
        assert g.get_option("sample_rows") == 24576
        idx, sc = _search(g, q, k)
        st = g.status()
        assert st["overflow_batches"] == 0 and st["spec_retries"] == 0
        assert idx[5, 0] == 1234567 and abs(sc[5, 0] - 1.0) < 1e-6
        g.set_option("spec_max_ratio", 20)                        # 1.6 M / 24 576 = 65 > 20: the chunk schedule
        assert g.get_option("sample_rows") == 8192
        idx_c, sc_c = _search(g, q, k)
        assert np.array_equal(idx, idx_c) and np.array_equal(sc, sc_c)
        g.set_option("spec_max_ratio", 160)
        g.set_option("force_exact", 1)
        idx_e, sc_e = _search(g, q, k)
        assert np.array_equal(idx, idx_e) and np.array_equal(sc, sc_e)
        # the speculative part off: rigorous chunks only
        g.set_option("force_exact", 0)
        g.set_option("speculative", 0)
        idx_r, sc_r = _search(g, q, k)
        assert np.array_equal(idx, idx_r) and np.array_equal(sc, sc_r)
        assert g.status()["overflow_batches"] == 0
    finally:
        g.close()


@pytest.mark.parametrize("mode", [1, 2, 3])
def test_async_tail_gives_the_same_answers(lib, mode):
    """mi_set_option("async_tail", 1 | 2 | 3): re-score + sort of batch i on the handle's own stream beside the scoring launch
    (1), only beside the query ingest + bootstrap (2) of batch i + 1, or -- deferred (3) -- enqueued by the call of batch
    i + 1 right before its scoring launch, so that it shares the device with that launch alone.  Several different batches in flight, distinct
    output buffers: after mi_search_join every batch equals the synchronous answer bit for bit; a synchronous host search
    on the same handle in between is unaffected."""
    import torch
    n, d, k = 300000, 256, 100
    raw = _device_rows(lib, 61, n, d)
    g = lib.Gallery.from_device_ptr(raw.data_ptr(), n, d)
    try:
        qs = [_device_rows(lib, 70 + i, 1024 if i % 2 == 0 else 300, d) for i in range(5)]
        ref = [_search(g, q, k) for q in qs]
        g.set_option("async_tail", mode)
        stream = torch.cuda.current_stream().cuda_stream
        outs = []
        for rep in range(2):
            for q in qs:
                idx = torch.empty((q.shape[0], k), dtype=torch.int64, device=q.device)
                sc = torch.empty((q.shape[0], k), dtype=torch.float32, device=q.device)
                g.search_device(q.data_ptr(), q.shape[0], k, idx.data_ptr(), sc.data_ptr(), None, stream)
                outs.append((idx, sc))
        g.join(stream)
        torch.cuda.synchronize()
        assert g.flags() == 0
        for j, (idx, sc) in enumerate(outs):
            ri, rs = ref[j % len(qs)]
            assert np.array_equal(idx.cpu().numpy(), ri) and np.array_equal(sc.cpu().numpy(), rs)
        # one call of 2500 queries = three batches through both buffer sets
        big = _device_rows(lib, 90, 2500, d)
        g.set_option("async_tail", 0)
        bi, bs = _search(g, big, k)
        g.set_option("async_tail", mode)
        idx = torch.empty((2500, k), dtype=torch.int64, device=big.device)
        sc = torch.empty((2500, k), dtype=torch.float32, device=big.device)
        g.search_device(big.data_ptr(), 2500, k, idx.data_ptr(), sc.data_ptr(), None, stream)
        hi, hs, _ = g.search(qs[1].cpu().numpy(), k)                     # host API: synchronous, own staging
        g.join(stream)
        torch.cuda.synchronize()
        assert np.array_equal(idx.cpu().numpy(), bi) and np.array_equal(sc.cpu().numpy(), bs)
        assert np.array_equal(hi, ref[1][0]) and np.array_equal(hs, ref[1][1])
    finally:
        g.close()


def test_ladder_thresholds_keep_the_answer_and_cut_the_survivors(lib):
    """In-launch ladder (csrc/common.h QueryState::lad_*): a tighter sample order statistic t_c becomes a RIGOROUS threshold
    for a query once K rows with approximate score >= t_c have been counted during the launch.  Answers are identical with
    the ladder on and off; the filter keeps fewer rows with it; an ordered gallery (every strong match in the first rows,
    the layout of src/test_rOP1m.py:136-139) and a query with fewer than K rows above its level are handled."""
    import torch
    n, d, k = 400000, 256, 100
    raw = _device_rows(lib, 81, n, d)
    q = _device_rows(lib, 82, 1024, d)
    for qi in range(40):                                   # 300 strong matches of 40 queries, all inside the first 12000 rows
        raw[qi * 300:(qi + 1) * 300] = q[qi] * 0.5 + 0.6 * raw[qi * 300:(qi + 1) * 300]
    raw[300000] = q[900] * 3.0                             # a single outlier match for one query
    torch.cuda.synchronize()
    g = lib.Gallery.from_device_ptr(raw.data_ptr(), n, d)
    try:
        res = {}
        for lad in (0, 1):
            g.set_option("ladder", lad)
            g.status(reset=True)
            idx, sc = _search(g, q, k)
            st = g.status()
            assert st["overflow_batches"] == 0 and st["spec_retries"] == 0
            res[lad] = (idx, sc, st["survivors"] / st["queries"], st["candidates"] / st["queries"])
        assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
        assert res[0][3] == res[1][3]                      # same candidate sets
        assert res[1][2] < 0.8 * res[0][2], (res[0][2], res[1][2])
        assert res[1][0][900, 0] == 300000
        g.set_option("force_exact", 1)
        idx_e, sc_e = _search(g, q, k)
        assert np.array_equal(res[1][0], idx_e) and np.array_equal(res[1][1], sc_e)
    finally:
        g.close()


@pytest.mark.parametrize("tiles", [8, 16, 32])
def test_sample_sizes_2048_4096_8192(tiles):
    """The threshold sample is 8192 rows by default; option "chunk0_tiles" selects 2048 / 4096 (each has its own threshold
    kernel).  Every size takes the single-launch schedule with the ladder and returns the exact answer."""
    from isehr_amd._lib import Gallery
    from isehr_amd.synth import synth_rows
    n, d, nq, k = 150000, 64, 300, 100
    g = synth_rows(300, 0, n, d)
    q = synth_rows(301, 0, nq, d)
    G = Gallery.from_host(g)
    try:
        assert G.get_option("sample_rows") == 8192
        G.set_option("chunk0_tiles", tiles)
        assert G.get_option("sample_rows") == tiles * 256
        i1, s1, _ = G.search(q, k)
        st = G.status(reset=True)
        G.set_option("force_exact", 1)
        i2, s2, _ = G.search(q, k)
    finally:
        G.close()
    assert st["overflow_batches"] == 0
    assert st["survivors"] / st["queries"] < 2500            # one filtered launch with a useful threshold
    assert np.array_equal(i1, i2) and np.array_equal(s1, s2)
    assert oracle.check_topk_parity(i1, oracle.exact_scores_f64(g, q), k, TAU) == []


def test_multi_batch_host_search_streams_with_the_deferred_tail_and_falls_back_on_a_flag():
    """Round 4: a host call with more than 1024 queries (the N x N callers, src/utils/Reranking.py:314-432) runs its internal
    batches as a stream with the deferred tail and reads the sticky flags once at the end (option "stream_tail", default on).
    Same answers bit for bit as one verified batch after the other; and a batch that overflows (3000 exact duplicates of one
    query's best row with the default caps) sends the call through the verified loop, which answers it through the fallbacks."""
    from isehr_amd._lib import Gallery
    n, d, nq, k = 150000, 256, 2500, 100
    g = synth_rows(401, 0, n, d)
    q = synth_rows(402, 0, nq, d)
    G = Gallery.from_host(g)
    try:
        assert G.get_option("stream_tail") == 1
        idx1, sc1, _ = G.search(q, k)
        st = G.status(reset=True)
        assert st["overflow_batches"] == 0 and st["spec_retries"] == 0 and st["queries"] == nq
        G.set_option("stream_tail", 0)
        idx0, sc0, _ = G.search(q, k)
        assert np.array_equal(idx0, idx1) and np.array_equal(sc0, sc1)
        s = oracle.exact_scores_f64(g, q[:64])
        assert oracle.check_topk_parity(idx1[:64], s, k, 1e-6) == []
    finally:
        G.close()
    g[5000:8000] = g[77]                                    # 3000 exact copies: the candidate lists of query 1500 overflow
    q[1500] = g[77]
    G = Gallery.from_host(g)
    try:
        idx2, sc2, _ = G.search(q, k)
        st = G.status()
        assert st["overflow_batches"] >= 1                  # the stream raised the flag; the verified loop answered
        assert set(idx2[1500]) <= set(range(5000, 8000)) | {77} and np.abs(sc2[1500] - 1.0).max() < 1e-6
        assert list(idx2[1500]) == sorted(idx2[1500])       # ties to the lower index
        s = oracle.exact_scores_f64(g, q[1400:1410])
        assert oracle.check_topk_parity(idx2[1400:1410], s, k, 1e-6) == []
    finally:
        G.close()


def test_calibrate_converges_the_xcd_shares_and_changes_no_answer():
    """mi_gallery_calibrate: scoring launches of the gallery's own first rows against the whole gallery, answers discarded,
    sticky flags cleared; a no-op below 512 gallery tiles.  The search after it equals the search before it."""
    import torch
    from isehr_amd import _lib
    n, d, nq, k = 140000, 128, 300, 50                      # 547 tiles: the weighted split is live
    dev = torch.device("cuda", 0)
    s = torch.cuda.current_stream().cuda_stream
    raw = torch.empty((n, d), dtype=torch.float32, device=dev)
    _lib.synth_fill_device(raw.data_ptr(), 411, 0, n, d, s)
    raw[1:400] = raw[0]                                     # duplicate rows at the start: the calibration's own queries tie massively
    q = torch.from_numpy(synth_rows(412, 0, nq, d)).to(dev)
    torch.cuda.synchronize()
    G = _lib.Gallery.from_device_ptr(raw.data_ptr(), n, d)
    try:
        idx = torch.empty((nq, k), dtype=torch.int64, device=dev)
        sc = torch.empty((nq, k), dtype=torch.float32, device=dev)
        G.search_device(q.data_ptr(), nq, k, idx.data_ptr(), sc.data_ptr(), None, s)
        torch.cuda.synchronize()
        ref = (idx.cpu().numpy().copy(), sc.cpu().numpy().copy())
        assert G.flags() == 0
        G.calibrate(6, s)
        torch.cuda.synchronize()
        assert G.flags() == 0                               # whatever the calibration launches flagged is gone
        G.status(reset=True)
        G.profile(True)
        G.search_device(q.data_ptr(), nq, k, idx.data_ptr(), sc.data_ptr(), None, s)
        torch.cuda.synchronize()
        ms = G.launch_ms()
        st = G.status()
        assert len(ms) == st["gemm_launches"] >= 1 and abs(float(ms.sum()) - st["gemm_ms"]) < 1e-3
        assert np.array_equal(idx.cpu().numpy(), ref[0]) and np.array_equal(sc.cpu().numpy(), ref[1])
    finally:
        G.close()
    small = _lib.Gallery.from_host(synth_rows(413, 0, 3000, 64))
    try:
        small.calibrate(8)                                  # 12 tiles: nothing to measure, nothing launched
        assert small.status()["searches"] == 0
    finally:
        small.close()
