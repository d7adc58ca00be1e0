"""GPU: the prepared gallery is the same BITS whatever layout and entry point its rows arrive through.

tests/golden/ingest_checksums.json holds the device-side section checksums (f32 rows | 16-bit image | rounding norms; MI355GAL
header) and the norm maxima of seeded galleries as the round-4 ingest kernel produced them from row-major device rows
(scripts/make_ingest_checksums.py, run before the round-5 rewrite).  The round-5 kernels -- one wave per row for rows that are
contiguous in memory, a one-pass panel kernel for the reference's [D, N] layout (callers hand over `vecs.T`,
src/test_rOP1m.py:155-157, src/online.py:96,132) -- and the block / append paths must all reproduce them: same summation
order for every norm, hence bit-identical galleries, hence identical answers by construction."""
import importlib.util
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIXTURE = json.load(open(os.path.join(ROOT, "tests", "golden", "ingest_checksums.json")))


def _maker():
    spec = importlib.util.spec_from_file_location("_make_ingest_checksums", os.path.join(ROOT, "scripts", "make_ingest_checksums.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


MAKER = _maker()
CASES = {c[0]: c for c in MAKER.CASES}
LAYOUTS = ["rowmajor_device", "rowmajor_host", "colmajor_host", "colmajor_device", "column_blocks", "append_device"]


def _build(layout, rows, norm, dtype):
    import torch
    from isehr_amd import _lib
    n, d = rows.shape
    code = _lib.MI_F64 if dtype == "f64" else _lib.MI_F32
    keep = None
    if layout == "rowmajor_device":
        keep = torch.from_numpy(rows).cuda()
        g = _lib.Gallery.from_device_ptr(keep.data_ptr(), n, d, norm_mode=norm, dtype=code)
    elif layout == "rowmajor_host":
        g = _lib.Gallery.from_host(rows, norm_mode=norm)
    elif layout == "colmajor_host":
        g = _lib.Gallery.from_host(np.ascontiguousarray(rows.T).T, norm_mode=norm)       # `vecs.T` of a [D, N] array
    elif layout == "colmajor_device":
        keep = torch.from_numpy(np.ascontiguousarray(rows.T)).cuda()                       # [D, N] on the device
        g = _lib.Gallery.from_device_ptr(keep.data_ptr(), n, d, norm_mode=norm, dtype=code, row_stride=1, col_stride=n)
    elif layout == "column_blocks":
        a = np.ascontiguousarray(rows.T)
        cut = max(1, (n * 2) // 5)
        g = _lib.Gallery.from_blocks([a[:, :cut], a[:, cut:]] if cut < n else [a], norm_mode=norm, chunk_rows=1000)
    else:
        g = _lib.Gallery.empty(n, d, norm_mode=norm)
        keep = torch.from_numpy(rows.astype(np.float32)).cuda()
        cut = (n // 3) if n >= 3 else 0
        s = torch.cuda.current_stream().cuda_stream
        if cut:
            g.append_device(keep.data_ptr(), cut, s)
        g.append_device(keep[cut:].data_ptr(), n - cut, s)
        torch.cuda.synchronize()
    return g, keep


@pytest.mark.parametrize("layout", LAYOUTS)
@pytest.mark.parametrize("name", sorted(FIXTURE))
def test_gallery_bits_are_layout_independent(name, layout):
    from isehr_amd import _lib
    _, seed, n, d, dtype, norm, f16, special = CASES[name]
    want = FIXTURE[name]
    if layout == "append_device" and dtype == "f64":
        pytest.skip("append_device takes float32 rows")
    if layout in ("column_blocks", "append_device") and norm == 0:
        pytest.skip("appendable raw galleries pick their image type after the fact (Gallery.from_blocks)")
    rows = MAKER.case_rows(seed, n, d, dtype, special)
    _lib.set_global_option("image_dtype", f16)
    try:
        g, keep = _build(layout, rows, norm, dtype)
        try:
            got = MAKER.gallery_sums(g)
        finally:
            g.close()
    finally:
        _lib.set_global_option("image_dtype", 1)
    del keep
    assert (got["n"], got["npad"], got["d"], got["dp"], got["img_f16"]) == (want["n"], want["npad"], want["d"], want["dp"],
                                                                            want["img_f16"])
    assert got["section_sums"][0] == want["section_sums"][0], "f32 rows differ"
    assert got["section_sums"][1] == want["section_sums"][1], "16-bit image differs"
    assert got["section_sums"][2] == want["section_sums"][2], "rounding norms differ"
    assert got["gstat3_hex"] == want["gstat3_hex"]
