"""GPU: the 256 x 256-tile MFMA scoring kernel (gemm_select.hip) at the shape bench.py times -- 1024 queries
(4 query tiles), D = 2048 (64 K-slices), speculative single-launch schedule, XCD walk -- against the float64
oracle.  Batches of <= 128 queries take the HBM-bound kernel of stream_select.hip, so every test here uses 1024."""
import numpy as np
import pytest

import oracle

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


def test_tile_kernel_structures_agree(lib):
    """kernel_variant 1 (the first structure of the tile kernel, kept for A/B) and the default structure return
    identical answers and identical candidate sets."""
    n = 70001
    raw = _device_rows(lib, 41, n, 320)
    q = _device_rows(lib, 42, 700, 320)
    g = lib.Gallery.from_device_ptr(raw.data_ptr(), n, 320)
    try:
        res = {}
        for v in (0, 1, 2):
            g.set_option("kernel_variant", v)
            g.status(reset=True)
            res[v] = _search(g, q, 50) + (g.status()["candidates"],)
        for v in (1, 2):
            assert np.array_equal(res[0][0], res[v][0]) and np.array_equal(res[0][1], res[v][1])
            assert res[0][2] == res[v][2]
        s = oracle.exact_scores_f64(raw.cpu().numpy(), q.cpu().numpy())
        assert oracle.check_topk_parity(res[0][0], s, 50, TAU) == []
    finally:
        g.close()


@pytest.mark.parametrize("img", ["f16", "bf16"])
def test_large_shard_chunk_schedule_equals_exact_path(lib, img):
    """A shard of more than 160 x 8192 rows (the per-GPU shard of BASELINE configs[3], 10M rows over 8 GPUs, has
    1.25 M) is too large for the sample-based single launch: the geometric chunk schedule of csrc/api.hip phase1_batch
    (rigorous thresholds from the rows seen so far, then one speculative launch for the rest) answers it.  Must equal
    the f32-scored path bit for bit; bf16 is the image type configs[3] names."""
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
        idx, sc = _search(g, q, k)
        st = g.status()
        assert st["overflow_batches"] == 0
        assert idx[5, 0] == 1234567 and abs(sc[5, 0] - 1.0) < 1e-6
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
