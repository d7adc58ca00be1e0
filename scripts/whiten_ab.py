"""mi_whiten_apply (host entry point, present in every build) of one library: python scripts/whiten_ab.py <lib.so> [rows]
Run under `rocprofv3 --kernel-trace --stats` to read the GEMM kernel's own duration (the host copies are inside the wall time)."""
import ctypes as C
import sys
import time

import numpy as np
import torch  # noqa: F401  (the HIP runtime of the process)

lib = C.CDLL(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
d = 2048
rng = np.random.default_rng(5)
X = rng.standard_normal((n, d), dtype=np.float32)
m = X[:4096].mean(axis=0).astype(np.float64)
P = rng.standard_normal((d, d)) / np.sqrt(d)
out = np.empty((n, d), dtype=np.float64)
lib.mi_whiten_apply.restype = C.c_int
for rep in range(2):
    t0 = time.time()
    rc = lib.mi_whiten_apply(C.c_void_p(X.ctypes.data), C.c_int64(n), C.c_int32(d), C.c_int(0), C.c_int64(d), C.c_int64(1),
                             C.c_void_p(m.ctypes.data), C.c_void_p(P.ctypes.data), C.c_int32(d), C.c_double(1e-6), C.c_int(0),
                             C.c_void_p(out.ctypes.data))
    dt = time.time() - t0
    assert rc == 0
    print("%s rows %d: %.3f s wall (incl. H2D of X, D2H of Y) = %.2f TFLOP/s wall" % (sys.argv[1], n, dt, 2.0 * n * d * d / dt / 1e12))
ref = (X[:8].astype(np.float64) - m) @ P.T
ref /= (np.linalg.norm(ref, axis=1, keepdims=True) + 1e-6)
print("max |d| vs numpy float64 on 8 rows: %.2e" % np.abs(out[:8] - ref).max())
