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


@pytest.mark.parametrize("seed", range(12))
def test_layouts_blocks_and_odd_shapes_give_the_same_rows(seed):
    """Seeded sweep over what the fixture cases do not pin: row counts around the panel / run / block boundaries, widths other
    than 2048 (the scratch-block path), rows with a stride (a [N, D'] array with D' > D), float64, raw (un-normalised) rows,
    column blocks of odd lengths (appends whose first row is no multiple of 4), host and device sources.  The gallery of every
    variant must hold the same stored rows, the same norm maxima and give the same answers as the row-major device gallery."""
    import torch
    from isehr_amd import _lib
    from isehr_amd.synth import synth_rows
    seed += 100 * int(os.environ.get("ISEHR_SWEEP_SEED", "0"))           # further seeds: ISEHR_SWEEP_SEED=n python -m pytest ...
    rng = np.random.default_rng(1000 + seed)
    d = int(rng.choice([2048, 2048, 2048, 64, 100, 320, 2500]))
    n = int(rng.choice([1, 3, 15, 16, 17, 255, 256, 257, 1000, 4099, 16385, 40001]))
    if d != 2048:
        n = min(n, 4099)
    f64 = bool(rng.integers(0, 2)) and d <= 2048
    norm = int(rng.choice([_lib.NORM_L2, _lib.NORM_L2, _lib.NORM_L2_EPS, _lib.NORM_NONE]))
    rows = synth_rows(2000 + seed, 0, n, d)
    if norm == _lib.NORM_NONE:
        rows = (rows / np.linalg.norm(rows, axis=1, keepdims=True)).astype(np.float32)      # unit rows: fp16 image like the others
    rows = rows.astype(np.float64) if f64 else rows
    code = _lib.MI_F64 if f64 else _lib.MI_F32
    q = synth_rows(3000 + seed, 0, 5, d)
    k = min(n, 10)

    def snapshot(g):
        try:
            idx, sc, _ = g.search(q, k)
            return g.get_rows(0, n).tobytes(), tuple(g.norm_bounds()), idx.tobytes(), sc.tobytes()
        finally:
            g.close()
    t = torch.from_numpy(rows).cuda()
    ref = snapshot(_lib.Gallery.from_device_ptr(t.data_ptr(), n, d, norm_mode=norm, dtype=code))
    variants = {}
    variants["host rows"] = _lib.Gallery.from_host(rows, norm_mode=norm)
    variants["host [D,N].T"] = _lib.Gallery.from_host(np.ascontiguousarray(rows.T).T, norm_mode=norm)
    wide = np.zeros((n, d + 24), dtype=rows.dtype)
    wide[:, :d] = rows
    variants["host rows with a stride"] = _lib.Gallery.from_host(wide[:, :d], norm_mode=norm)
    tt = torch.from_numpy(np.ascontiguousarray(rows.T)).cuda()
    variants["device [D,N]"] = _lib.Gallery.from_device_ptr(tt.data_ptr(), n, d, norm_mode=norm, dtype=code, row_stride=1,
                                                            col_stride=n)
    tw = torch.from_numpy(wide).cuda()
    variants["device rows with a stride"] = _lib.Gallery.from_device_ptr(tw.data_ptr(), n, d, norm_mode=norm, dtype=code,
                                                                         row_stride=d + 24, col_stride=1)
    if norm != _lib.NORM_NONE and n >= 3:
        a = np.ascontiguousarray(rows.T)
        c1 = max(1, n // 3) | 1                                        # odd: the next block starts at a row that is no multiple of 4
        variants["odd column blocks"] = _lib.Gallery.from_blocks([a[:, :c1], a[:, c1:]], norm_mode=norm, chunk_rows=999)
    for mode in (0, 1):
        _lib.set_global_option("host_ingest", mode)
        try:
            variants["host [D,N].T, host_ingest %d" % mode] = _lib.Gallery.from_host(np.ascontiguousarray(rows.T).T, norm_mode=norm)
        finally:
            _lib.set_global_option("host_ingest", 1)
    for name, g in variants.items():
        got = snapshot(g)
        assert got[0] == ref[0], (name, n, d, f64, norm, "stored rows differ")
        assert got[1] == ref[1], (name, n, d, f64, norm, "norm maxima differ")
        assert got[2] == ref[2] and got[3] == ref[3], (name, n, d, f64, norm, "answers differ")


def test_gallery_file_round_trip_small_and_mapped(tmp_path):
    """The loader maps the file and lets the runtime copy from the mapping; a file of a few KB, one with an odd width, and the
    checks of a corrupted payload all go through it."""
    from isehr_amd import _lib
    from isehr_amd.synth import synth_rows
    for n, d in ((1, 64), (257, 100), (3000, 2048)):
        rows = synth_rows(77, 0, n, d)
        g = _lib.Gallery.from_host(rows)
        path = str(tmp_path / ("g_%d_%d.bin" % (n, d)))
        try:
            want = MAKER.gallery_sums(g)
            g.save(path)
        finally:
            g.close()
        L = _lib.Gallery.load(path)
        try:
            assert MAKER.gallery_sums(L)["section_sums"] == want["section_sums"]
        finally:
            L.close()
        raw = bytearray(open(path, "rb").read())
        raw[len(raw) // 2] ^= 0x40                                    # one flipped bit in the payload
        open(path, "wb").write(raw)
        with pytest.raises(RuntimeError, match="checksum"):
            _lib.Gallery.load(path)
