"""whitenapply (src/utils/whiten.py:4-12) through the HIP path: X[D,N], m[D,1], P[D,D] ->
P[:dimensions] @ (X - m), columns divided by (||.|| + 1e-6); float64 like the reference.
The learning side (whitenlearn / pcawhitenlearn, one-off eig / cholesky) is out of scope."""
import numpy as np

from . import _lib


def whitenapply_hip(X, m, P, dimensions=None, device=0):
    X = np.asarray(X)
    if not dimensions:
        dimensions = np.shape(P)[0]
    out = _lib.whiten_apply(X.T, m, P, int(dimensions), 1e-6, device)      # [N, dims]
    return out.T
