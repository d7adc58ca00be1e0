"""GPU: the boundary is a C ABI -- a plain C99 program (gcc; no Python, no torch, no C++) reads a float32 [D, N] array and a
query block from files, calls mi_gallery_create(MI_HOST) on the `.T` view (row stride 1, column stride N: what the reference's
callers pass, src/test_rOP1m.py:155-157), mi_knn_search, mi_gallery_save / mi_gallery_load / a second search, and prints the
indices; pytest compares them with the oracle."""
import os
import shutil
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PROGRAM = r"""
#include <stdio.h>
#include <stdlib.h>
#include "mi355_retrieval.h"
#define CHECK(e) do { int rc_ = (e); if (rc_ != MI_OK) { fprintf(stderr, "%s: %s\n", #e, mi_last_error()); return 10 + rc_; } } while (0)
static float* slurp(const char* path, size_t count) {
  float* p = (float*)malloc(count * sizeof(float));
  FILE* f = fopen(path, "rb");
  if (!p || !f || fread(p, sizeof(float), count, f) != count) return NULL;
  fclose(f);
  return p;
}
int main(int argc, char** argv) {
  const long n = atol(argv[3]), d = atol(argv[4]), nq = atol(argv[5]), k = atol(argv[6]);
  float* vecs = slurp(argv[1], (size_t)n * d);      /* [D, N] */
  float* qvecs = slurp(argv[2], (size_t)nq * d);    /* [D, Q] */
  int64_t* idx = (int64_t*)malloc((size_t)nq * k * sizeof(int64_t));
  int64_t* idx2 = (int64_t*)malloc((size_t)nq * k * sizeof(int64_t));
  float* score = (float*)malloc((size_t)nq * k * sizeof(float));
  double seconds = 0.0;
  mi_gallery *g = NULL, *h = NULL;
  long i;
  (void)argc;
  if (!vecs || !qvecs || !idx || !idx2 || !score) return 3;
  CHECK(mi_gallery_create(vecs, n, (int32_t)d, MI_F32, 1, n, MI_HOST, MI_NORM_L2, 0, 0, &g));
  CHECK(mi_knn_search(g, qvecs, nq, MI_F32, 1, nq, (int32_t)k, idx, score, &seconds));
  CHECK(mi_gallery_save(g, argv[7]));
  CHECK(mi_gallery_destroy(g));
  CHECK(mi_gallery_load(argv[7], 0, &h));
  CHECK(mi_knn_search(h, qvecs, nq, MI_F32, 1, nq, (int32_t)k, idx2, NULL, NULL));
  CHECK(mi_gallery_destroy(h));
  for (i = 0; i < nq * k; ++i) if (idx[i] != idx2[i]) return 4;
  for (i = 0; i < nq * k; ++i) printf("%lld %.9g\n", (long long)idx[i], (double)score[i]);
  return seconds > 0.0 ? 0 : 5;
}
"""


def test_a_c_program_searches_through_the_abi(tmp_path):
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    from isehr_amd.synth import synth_rows
    from oracle import retrieval_oracle as oracle
    n, d, nq, k = 5000, 2048, 6, 15
    rows, q = synth_rows(31, 0, n, d), synth_rows(32, 0, nq, d)
    np.ascontiguousarray(rows.T).tofile(tmp_path / "vecs.f32")
    np.ascontiguousarray(q.T).tofile(tmp_path / "qvecs.f32")
    (tmp_path / "consumer.c").write_text(PROGRAM)
    libdir = os.path.join(ROOT, "image-search-engine-for-historical-research_amd")
    exe = tmp_path / "consumer"
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                        str(tmp_path / "consumer.c"), "-o", str(exe), "-L", libdir, "-l:libmi355_retrieval.so",
                        "-Wl,-rpath," + libdir, "-Wl,--allow-shlib-undefined"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-3000:]
    env = dict(os.environ, LD_LIBRARY_PATH=os.pathsep.join(p for p in ("/opt/rocm/lib", os.environ.get("LD_LIBRARY_PATH", "")) if p))
    r = subprocess.run([str(exe), str(tmp_path / "vecs.f32"), str(tmp_path / "qvecs.f32"), str(n), str(d), str(nq), str(k),
                        str(tmp_path / "gallery.bin")], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, (r.returncode, r.stderr[-2000:])
    vals = [ln.split() for ln in r.stdout.strip().splitlines()]
    idx = np.array([int(v[0]) for v in vals], dtype=np.int64).reshape(nq, k)
    sc = np.array([float(v[1]) for v in vals]).reshape(nq, k)
    exact = oracle.exact_scores_f64(rows, q)
    assert oracle.check_topk_parity(idx, exact, k, 1e-6) == []
    assert np.abs(sc - np.take_along_axis(exact, idx, axis=1)).max() < 1e-6
