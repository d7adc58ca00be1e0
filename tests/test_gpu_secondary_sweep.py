"""GPU: seeded sweeps of the secondary operators of the path -- whitenapply, alpha-QE weights, k-reciprocal re-ranking --
over shapes, strides, dtypes and constants other than the reference's defaults, each against the oracle."""
import numpy as np
import pytest

import oracle
from isehr_amd.synth import synth_rows

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", range(8))
def test_whitenapply_shapes_and_layouts(case):
    """src/utils/whiten.py:4-12 for [D, N] inputs of either dtype, C- or F-ordered, dims < D, D not a multiple of 16."""
    from isehr_amd.whiten import whitenapply_hip
    rng = np.random.default_rng(100 + case)
    d = int(rng.choice([7, 24, 100, 130, 257]))
    n = int(rng.integers(1, 900))
    dims = int(rng.integers(1, d + 1))
    X = rng.standard_normal((d, n))
    if case % 2:
        X = X.astype(np.float32)
    if case % 3 == 0:
        X = np.asfortranarray(X)
    m = rng.standard_normal((d, 1))
    P = rng.standard_normal((d, d)) / np.sqrt(d)
    got = whitenapply_hip(X, m, P, dims)
    ref = oracle.whitenapply(np.asarray(X, dtype=np.float64), m, P, dims)
    assert got.shape == (dims, n) and got.dtype == np.float64
    assert np.abs(got - ref).max() < 1e-12


@pytest.mark.parametrize("k_qe,w", [(1, 4.0), (5, 0.0), (10, 4.0), (7, 1.5), (16, 2.0)])
def test_alpha_qe_weights(k_qe, w):
    """feature_enhancement (src/utils/Reranking.py:195-208) for k and w other than (3, 4) and (10, 4): the expanded query
    against the oracle's float64 sum, the re-search against the float64 scores of that query."""
    from isehr_amd._lib import Gallery, NORM_NONE
    n, d, nq, k = 30000, 72, 45, 60
    v = synth_rows(200 + k_qe, 0, n, d).astype(np.float64) * 0.6 + 1.2 * synth_rows(9, 0, 50, d)[np.arange(n) % 50]
    v = (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)
    q = v[::661][:nq] + 0.1 * synth_rows(10, 0, nq, d)
    G = Gallery.from_host(v, norm_mode=NORM_NONE)
    try:
        base, _, _ = G.search(q, max(k_qe, 20))
        idx, sc, qx, _ = G.aqe_search(np.ascontiguousarray(base.T), k_qe, w, k, return_qexp=True)
    finally:
        G.close()
    qx_ref, _ = oracle.feature_enhancement(k_qe, base.T, v.T.astype(np.float64), w)
    assert np.abs(qx - qx_ref.T).max() < 2e-7
    s = v.astype(np.float64) @ qx.astype(np.float64).T
    assert oracle.check_topk_parity(idx, s.T, k, 1e-6) == []
    assert np.abs(np.take_along_axis(s.T, idx, 1) - sc).max() < 3e-7


@pytest.mark.parametrize("case", range(5))
def test_kr_rerank_shapes_and_constants(case):
    """kr_reranking (src/utils/Reranking.py:447-624) over sizes and (k1, k2, lambda) the reference does not use, incl.
    duplicated gallery images (ties in the initial ranking)."""
    from isehr_amd.reranking import kr_reranking_hip
    rng = np.random.default_rng(300 + case)
    n = int(rng.integers(120, 2500))
    d = int(rng.choice([16, 48, 128]))
    nq = int(rng.integers(1, 30))
    k1 = int(rng.choice([5, 12, 20, 16]))                    # the reciprocal-set buffers hold k1 <= 20 (the reference's constant)
    k2 = int(rng.choice([1, 3, 6]))
    lam = float(rng.choice([0.0, 0.3, 0.7, 1.0]))
    ncl = max(4, n // 40)
    v = rng.standard_normal((n, d)) * 0.6 + 1.3 * rng.standard_normal((ncl, d))[np.arange(n) % ncl]
    if case % 2:
        v[n // 2:n // 2 + 5] = v[3]                          # duplicates of one image
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    q = v[rng.choice(n, nq, replace=False)] + 0.15 * rng.standard_normal((nq, d))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    qv, vecs = q.T.astype(np.float32), v.T.astype(np.float32)
    got, dist = kr_reranking_hip(qv, vecs, k1=k1, k2=k2, lambda_value=lam, return_dist=True)
    ref, final = oracle.kr_reranking(qv, vecs, k1=k1, k2=k2, lambda_value=lam, return_dist=True)
    assert got.shape == ref.shape == (nq, n)
    assert all(len(set(r)) == n for r in got)
    # V and V_qe are float16 by the reference's definition (`V_qe = np.zeros_like(V, dtype=np.float16)`, src/utils/Reranking.py:581): one float32 summation-order
    # difference can flip the float16 rounding of a V_qe entry (relative 1e-3 of a ~0.01 weight), i.e. ~1e-5 in a distance
    tol = 2e-5
    assert np.abs(np.take_along_axis(final, got, 1) - dist).max() < tol
    assert (np.diff(dist, axis=1) >= 0).all()
    # position by position: a different image only where the final distances agree within that noise
    assert np.abs(np.take_along_axis(final, got, 1) - np.take_along_axis(final, ref, 1)).max() < tol


def test_kr_rerank_rejects_k1_beyond_its_buffers():
    from isehr_amd.reranking import kr_reranking_hip
    v = synth_rows(1, 0, 300, 16)
    with pytest.raises(RuntimeError, match="k1 too large"):
        kr_reranking_hip(v[:4].T.copy(), v.T.copy(), k1=30)


@pytest.mark.parametrize("b,c,cout,h,w,p", [(4, 2048, 2048, 24, 32, 3.0), (1, 2048, 2048, 32, 43, 3.0), (5, 512, 128, 7, 9, 2.2),
                                           (2, 100, 100, 1, 1, 3.0), (3, 2048, 512, 16, 16, 1.0)])
def test_descriptor_tail_at_real_shapes(b, c, cout, h, w, p):
    """GeM(p, eps) -> L2N -> whiten Linear -> L2N (src/networks/imageretrievalnet.py:183-187, src/layers/functional.py) at
    the network's real shape (2048 channels, 2048 outputs, feature maps of a 1024-pixel image) and odd ones, against the
    same chain in float64."""
    import torch
    from isehr_amd.extractor import DescriptorTail
    g = torch.Generator().manual_seed(b * 1000 + c + h)
    feat = torch.rand((b, c, h, w), generator=g) * 2.0 - 0.3          # post-ReLU-like, with values below eps to clamp
    W = torch.randn((cout, c), generator=g) / c ** 0.5
    bias = torch.randn((cout,), generator=g) * 0.1
    got = DescriptorTail(p, 1e-6, W.cuda(), bias.cuda())(feat.cuda()).cpu().double()
    x = feat.double().clamp(min=1e-6).pow(p).mean(dim=(2, 3)).pow(1.0 / p)               # LF.gem
    x = x / (x.norm(dim=1, keepdim=True) + 1e-6)                                          # LF.l2n
    y = x @ W.double().t() + bias.double()
    y = y / (y.norm(dim=1, keepdim=True) + 1e-6)
    assert got.shape == (b, cout)
    assert (got - y).abs().max().item() < 2e-6
