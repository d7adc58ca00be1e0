"""One-off validation of the GPU diffusion against the oracle at rOxford5k size (N = 4993, n_trunc = 2000,
kd = 200).  The oracle needs a few minutes of CPU, so this is a script, not a test."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import isehr_amd, oracle
from isehr_amd.synth import synth_rows
from isehr_amd.diffusion import Diffusion
n, d, T, kd = 4993, 128, 2000, 200
f = synth_rows(5, 0, n, d).astype(np.float64)
c = synth_rows(6, 0, 40, d).astype(np.float64)
f = 0.8 * f + 1.5 * c[np.arange(n) % 40]
f /= np.linalg.norm(f, axis=1, keepdims=True)
f = f.astype(np.float32)
dd = Diffusion(f)
t0 = time.time(); ids, vals = dd.gallery.diffusion_offline(T, kd); t_gpu = time.time() - t0
dd.close()
t0 = time.time(); ref_off, ref_sims, ref_ids, lap, ref_scores = oracle.diffusion_offline(f, T, kd, return_parts=True); t_cpu = time.time() - t0
same = (ids == ref_ids).all(axis=1)
err = np.abs(vals[same].astype(np.float64) - ref_scores[same]).max()
print("GPU %.2f s, oracle (scipy, 1 core) %.1f s; rows with identical truncation support: %.4f; max |offline - oracle| on them: %.2e; max |oracle| %.3f"
      % (t_gpu, t_cpu, same.mean(), err, np.abs(ref_scores).max()))
