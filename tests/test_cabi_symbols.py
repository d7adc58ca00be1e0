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
