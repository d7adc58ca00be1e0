"""CPU: the C-ABI library loads and exports every symbol include/mi355_retrieval.h declares
(no compute calls without a GPU)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built_lib():
    import __graft_entry__
    __graft_entry__.build()
    from isehr_amd import _lib
    return _lib.load(), _lib


def test_header_symbols_are_exported(built_lib):
    lib, _lib = built_lib
    hdr = open(os.path.join(ROOT, "include", "mi355_retrieval.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(mi_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    for name in sorted(declared):
        assert hasattr(lib, name), "library does not export " + name
    # the ctypes table binds exactly the declared functions
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)


def test_error_reporting_without_gpu(built_lib):
    lib, _lib = built_lib
    import ctypes as C
    rc = lib.mi_set_option(None, b"x", 1.0)
    assert rc == 1 and b"null" in lib.mi_last_error()
    with pytest.raises(RuntimeError):
        _lib.check(rc)


def test_header_is_plain_c_and_a_c_program_links_against_the_library(built_lib, tmp_path):
    """The boundary is a C ABI: a C99 translation unit (gcc, no C++, no torch) that includes the header compiles with -Wall
    -Werror -pedantic, links against the in-tree library and runs -- error paths only, nothing that needs a GPU."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    src = tmp_path / "consumer.c"
    src.write_text(r"""
#include <stdio.h>
#include <string.h>
#include "mi355_retrieval.h"
int main(void) {
  mi_gallery* g = NULL;
  mi_search_stats st;
  float rows[4] = {1.f, 0.f, 0.f, 1.f};
  memset(&st, 0, sizeof st);
  /* argument checks answer before any device is touched */
  if (mi_gallery_create(rows, 0, 2, MI_F32, 2, 1, MI_HOST, MI_NORM_L2, 0, 0, &g) == MI_OK) return 2;
  if (strlen(mi_last_error()) == 0) return 3;
  if (mi_set_option(NULL, "speculative", 1.0) == MI_OK) return 4;
  if (mi_set_global_option("no_such_option", 1.0) == MI_OK) return 5;
  if (mi_gallery_destroy(NULL) != MI_OK) return 6;
  printf("%s\n", mi_last_error());
  return 0;
}
""")
    libdir = os.path.join(ROOT, "image-search-engine-for-historical-research_amd")
    exe = tmp_path / "consumer"
    cmd = ["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
           "-L", libdir, "-l:libmi355_retrieval.so", "-Wl,-rpath," + libdir, "-Wl,--allow-shlib-undefined"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-3000:]
    env = dict(os.environ, LD_LIBRARY_PATH=os.pathsep.join(
        [p for p in (os.path.join(os.path.dirname(__import__("torch").__file__), "lib"), "/opt/rocm/lib",
                     os.environ.get("LD_LIBRARY_PATH", "")) if p]))
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr[-2000:])
    assert "unknown global option" in r.stdout
