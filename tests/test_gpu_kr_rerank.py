"""GPU: k-reciprocal re-ranking (csrc/kr_rerank.hip) against the reference's own output (golden captured from
src/utils/Reranking.py:447-624 kr_reranking) and against the oracle's restatement on a larger case."""
import os

import numpy as np
import pytest

import oracle
from isehr_amd.synth import synth_rows

pytestmark = pytest.mark.gpu


def _clustered(seed, n, d, ncl, nq, step):
    v = synth_rows(seed, 0, n, d).astype(np.float64)
    c = synth_rows(seed + 1, 0, ncl, d).astype(np.float64)
    v = 0.6 * v + 1.3 * c[np.arange(n) % ncl]
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    q = v[::step][:nq] + 0.15 * synth_rows(seed + 2, 0, nq, d)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    return q.T.astype(np.float32), v.T.astype(np.float32)        # [D, Q], [D, N] like the reference's arguments


def _same_up_to_near_ties(got, ref, final, tol):
    """position by position: a different image only where the two final distances agree within tol"""
    a = np.take_along_axis(final, got, 1)
    b = np.take_along_axis(final, ref, 1)
    return float(np.abs(a - b).max()) <= tol and all(len(set(r)) == len(r) for r in got)


def test_kr_rerank_vs_reference_golden(golden_dir):
    from isehr_amd.reranking import kr_reranking_hip
    qv, vecs = _clustered(98, 400, 32, 25, 7, 57)
    ref = np.load(os.path.join(golden_dir, "kr_rerank.npz"))["indices"]
    got, dist = kr_reranking_hip(qv, vecs, return_dist=True)
    assert got.shape == ref.shape == (7, 400) and got.dtype == np.int64
    assert (np.diff(dist, axis=1) >= 0).all()
    _, final = oracle.kr_reranking(qv, vecs, return_dist=True)
    assert np.abs(np.take_along_axis(final, got, 1) - dist).max() < 2e-6        # the distances themselves
    assert _same_up_to_near_ties(got, ref, final, 2e-6)
    assert (got == ref).mean() > 0.99


def test_kr_rerank_vs_oracle_larger_case():
    from isehr_amd.reranking import kr_reranking_hip
    qv, vecs = _clustered(108, 3000, 64, 60, 24, 101)
    got, dist = kr_reranking_hip(qv, vecs, return_dist=True)
    ref, final = oracle.kr_reranking(qv, vecs, return_dist=True)
    assert got.shape == (24, 3000)
    assert np.abs(np.take_along_axis(final, got, 1) - dist).max() < 2e-6
    assert _same_up_to_near_ties(got, ref, final, 2e-6)
    # the re-ranking does something: the jaccard term moves images relative to the plain cosine order
    plain = np.argsort(-(qv.T @ vecs), axis=1, kind="stable")
    assert (got[:, :50] != plain[:, :50]).mean() > 0.05
    # other constants than the reference's
    got2 = kr_reranking_hip(qv, vecs, k1=10, k2=1, lambda_value=0.5)
    ref2, final2 = oracle.kr_reranking(qv, vecs, k1=10, k2=1, lambda_value=0.5, return_dist=True)
    assert _same_up_to_near_ties(got2, ref2, final2, 2e-6)
