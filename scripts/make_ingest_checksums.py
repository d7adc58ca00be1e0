#!/usr/bin/env python3
"""Records the section checksums of prepared galleries (MI355GAL header: f32 rows | 16-bit image | rounding norms, computed
on the device by csrc/ingest.hip checksum_kernel) for a list of seeded shapes, as tests/golden/ingest_checksums.json.

Run on a GPU box with the library whose galleries are to be pinned:

    python scripts/make_ingest_checksums.py > tests/golden/ingest_checksums.json

Round 5 rewrote the gallery ingest kernels (one wave per row; a one-pass kernel for the reference's [D, N] layout); this file
was generated with the round-4 kernels BEFORE the rewrite, from row-major device rows (the persistent one-row-per-workgroup
kernel of rounds 3-4, whose summation order is the canonical one).  tests/test_gpu_ingest_bits.py requires every layout and
every entry point of the new kernels to reproduce these sums: bit-identical galleries, not just equal answers.
"""
import json
import os
import struct
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# (name, seed, rows, d, source dtype, norm mode, image dtype f16?, special rows)
CASES = [
    ("n70001_d2048_f32_l2_f16", 11, 70001, 2048, "f32", 1, 1, False),
    ("n70001_d2048_f32_l2_bf16", 11, 70001, 2048, "f32", 1, 0, False),
    ("n5000_d2048_f64_l2_f16", 12, 5000, 2048, "f64", 1, 1, False),
    ("n3000_d320_f32_l2eps_f16", 13, 3000, 320, "f32", 2, 1, False),
    ("n3000_d100_f32_l2_f16", 14, 3000, 100, "f32", 1, 1, False),
    ("n2049_d64_f32_l2_f16", 15, 2049, 64, "f32", 1, 1, False),
    ("n1500_d2500_f32_l2_f16", 16, 1500, 2500, "f32", 1, 1, False),
    ("n1200_d4096_f64_l2eps_bf16", 17, 1200, 4096, "f64", 2, 0, False),
    ("n4000_d2048_f32_none_bf16", 18, 4000, 2048, "f32", 0, 0, False),
    ("n4000_d512_f32_l2_f16_special", 19, 4000, 512, "f32", 1, 1, True),
    ("n257_d2048_f32_l2_f16", 20, 257, 2048, "f32", 1, 1, False),
    ("n1_d2048_f32_l2_f16", 21, 1, 2048, "f32", 1, 1, False),
]

HEADER = struct.Struct("<8s4q4i3fI3QQ")   # csrc/api_file.hip FileHeader


def case_rows(seed, n, d, dtype, special):
    import numpy as np
    from isehr_amd.synth import synth_rows
    g = synth_rows(seed, 0, n, d)
    if special:
        g[7] = 0.0                                   # zero row: NaN after the normalisation, like the reference
        g[8] = g[9]                                  # exact duplicate
        g[10] *= 1e-20                               # tiny row: denormal squares
        g[11] *= 1e18                                # huge row
        g[12, ::2] = 0.0
        g[13] = -g[13]
    return g.astype(np.float64) if dtype == "f64" else g


def file_sums(path):
    with open(path, "rb") as f:
        h = HEADER.unpack(f.read(HEADER.size))
    magic, version, n, npad, row_offset, d, dp, norm_mode, img_f16 = h[:9]
    gstat3 = h[9:12]
    sums = h[13:16]
    assert magic == b"MI355GAL" and version == 2
    return {"n": n, "npad": npad, "d": d, "dp": dp, "norm_mode": norm_mode, "img_f16": img_f16,
            "gstat3_hex": [struct.pack("<f", v).hex() for v in gstat3], "section_sums": ["%016x" % s for s in sums]}


def gallery_sums(g):
    fd, path = tempfile.mkstemp(suffix=".mi355gal")
    os.close(fd)
    try:
        g.save(path)
        return file_sums(path)
    finally:
        os.unlink(path)


def main():
    import torch
    import isehr_amd  # noqa: F401
    from isehr_amd import _lib
    out = {}
    for name, seed, n, d, dtype, norm, f16, special in CASES:
        rows = case_rows(seed, n, d, dtype, special)
        _lib.set_global_option("image_dtype", f16)
        t = torch.from_numpy(rows).cuda()
        g = _lib.Gallery.from_device_ptr(t.data_ptr(), n, d, norm_mode=norm, device=0,
                                         dtype=_lib.MI_F64 if dtype == "f64" else _lib.MI_F32)
        try:
            rec = gallery_sums(g)
        finally:
            g.close()
        rec.update(seed=seed, dtype=dtype, special=special)
        out[name] = rec
    _lib.set_global_option("image_dtype", 1)
    json.dump(out, sys.stdout, indent=1, sort_keys=True)
    sys.stdout.write("\n")


if __name__ == "__main__":
    main()
