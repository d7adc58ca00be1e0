"""ctypes binding of libmi355_retrieval.so (include/mi355_retrieval.h).

There is no CPU fallback: if the shared library is missing or a call fails, a RuntimeError is
raised.  `load()` never builds; build with `python image-search-engine-for-historical-research_amd/build.py`
or `__graft_entry__.build()`.
"""
import ctypes as C
import os
import threading

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libmi355_retrieval.so")

MI_F32, MI_F64 = 0, 1
MI_HOST, MI_DEVICE = 0, 1
NORM_NONE, NORM_L2, NORM_L2_EPS = 0, 1, 2

c_i64p = C.POINTER(C.c_int64)
c_f32p = C.POINTER(C.c_float)
c_f64p = C.POINTER(C.c_double)


class SearchStats(C.Structure):
    _fields_ = [("searches", C.c_int64), ("queries", C.c_int64), ("overflow_batches", C.c_int64),
                ("survivors", C.c_int64), ("candidates", C.c_int64), ("gemm_ms", C.c_double),
                ("gemm_launches", C.c_int64), ("gemm_flops", C.c_double), ("gemm_bytes", C.c_double),
                ("kernel_clock_mhz", C.c_double), ("spec_retries", C.c_int64), ("inkernel_repairs", C.c_int64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


# every symbol include/mi355_retrieval.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "mi_last_error": (C.c_char_p, []),
    "mi_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "mi_gallery_create": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int, C.c_int64, C.c_int64, C.c_int,
                                    C.c_int, C.c_int, C.c_int64, C.POINTER(C.c_void_p)]),
    "mi_gallery_create_empty": (C.c_int, [C.c_int64, C.c_int32, C.c_int, C.c_int, C.c_int64, C.POINTER(C.c_void_p)]),
    "mi_gallery_append_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "mi_desc_tail_device": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_void_p,
                                      C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mi_desc_ms_accumulate_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_int, C.c_void_p]),
    "mi_desc_ms_finish_device": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p]),
    "mi_gallery_append": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int64, C.c_int64, C.c_int]),
    "mi_gallery_destroy": (C.c_int, [C.c_void_p]),
    "mi_gallery_info": (C.c_int, [C.c_void_p, c_i64p, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                  C.POINTER(C.c_int32), c_i64p, c_i64p]),
    "mi_gallery_save": (C.c_int, [C.c_void_p, C.c_char_p]),
    "mi_gallery_load": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]),
    "mi_gallery_get_rows": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]),
    "mi_knn_search": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int64, C.c_int64, C.c_int32,
                                C.c_void_p, C.c_void_p, c_f64p]),
    "mi_knn_search_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p]),
    "mi_knn_phase1_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    "mi_kth_of_gathered_device": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    "mi_knn_phase2_device": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p]),
    "mi_topk_merge_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_void_p,
                                       C.c_void_p, C.c_void_p]),
    "mi_topk_merge_strided_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.c_int32,
                                               C.c_void_p, C.c_void_p, C.c_void_p]),
    "mi_aqe_partial_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int32,
                                        C.c_double, C.c_void_p, C.c_void_p]),
    "mi_aqe_rows_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_void_p,
                                     C.c_void_p]),
    "mi_aqe_combine_device": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_double, C.c_void_p, C.c_void_p]),
    "mi_aqe_finish_device": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_double, C.c_void_p, C.c_void_p,
                                       C.c_void_p]),
    "mi_aqe_search": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_double,
                                C.c_double, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, c_f64p]),
    "mi_knn_dense_search": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int64, C.c_int64, C.c_int32,
                                      C.c_void_p, C.c_void_p, c_f64p]),
    "mi_knn_dense64_search": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int64, C.c_int64, C.c_int32,
                                        C.c_void_p, C.c_void_p, C.c_void_p, c_f64p]),
    "mi_rank_all": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int64, C.c_int64, C.c_int, C.c_void_p,
                              C.c_void_p, c_f64p]),
    "mi_rank_prefix": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int64, C.c_int64, C.c_int, C.c_int64,
                                 C.c_void_p, C.c_void_p, c_f64p]),
    "mi_rank_positions": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int64, C.c_int64, C.c_int,
                                    C.c_void_p, C.c_int32, C.c_void_p]),
    "mi_kr_rerank": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int64,
                               C.c_int32, C.c_int, C.c_int32, C.c_int32, C.c_double, C.c_int, C.c_void_p, C.c_void_p]),
    "mi_diffusion_offline": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_double, C.c_int32, C.c_int32, C.c_double,
                                       C.c_void_p, C.c_void_p, C.c_void_p]),
    "mi_diffusion_offline_nodes": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_double, C.c_int32, C.c_int32, C.c_double,
                                             C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mi_diffusion_set_offline": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]),
    "mi_diffusion_online": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int64, C.c_int64, C.c_int32,
                                      C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "mi_gather_weighted": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_void_p,
                                     C.c_void_p]),
    "mi_column_sum": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int, C.c_int64, C.c_int64, C.c_int, C.c_void_p]),
    "mi_whiten_apply": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int, C.c_int64, C.c_int64, C.c_void_p,
                                  C.c_void_p, C.c_int32, C.c_double, C.c_int, C.c_void_p]),
    "mi_whiten_apply_device": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int, C.c_int64, C.c_int64, C.c_void_p,
                                         C.c_void_p, C.c_int32, C.c_double, C.c_void_p, C.c_void_p]),
    "mi_gallery_append_whitened_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int, C.c_int64,
                                                    C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mi_gallery_calibrate": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "mi_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "mi_search_status": (C.c_int, [C.c_void_p, C.POINTER(SearchStats), C.c_int]),
    "mi_profile_launch_ms": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]),
    "mi_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_double]),
    "mi_get_option": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_double)]),
    "mi_search_flags": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32)]),
    "mi_search_join": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mi_gallery_norm_bounds": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_int]),
    "mi_gallery_set_image_dtype": (C.c_int, [C.c_void_p, C.c_int]),
    "mi_debug_read_cycles": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "mi_debug_xcc_shares": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]),
    "mi_debug_sample_source_row": (C.c_int64, [C.c_int64, C.c_int64, C.c_int64]),
    "mi_set_global_option": (C.c_int, [C.c_char_p, C.c_double]),
    "mi_get_global_option": (C.c_int, [C.c_char_p, C.POINTER(C.c_double)]),
    "mi_online_create": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_int32, C.c_int32,
                                   C.POINTER(C.c_void_p)]),
    "mi_online_query": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mi_online_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "mi_online_destroy": (C.c_int, [C.c_void_p]),
    "mi_debug_online_clients": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                          C.POINTER(C.c_double)]),
    "mi_synth_fill_device": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int64, C.c_int64, C.c_int32, C.c_void_p]),
}

_lib = None
_lock = threading.Lock()


def load():
    """Loads the HIP library; raises RuntimeError (never falls back) if it is not built."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libmi355_retrieval.so is not built (%s). Run `python __graft_entry__.py build`; "
                "this package has no CPU fallback." % LIB_PATH)
        # PyTorch-ROCm bundles its own HIP runtime (torch/lib/libamdhip64.so, soname libamdhip64.so.7).
        # Two HIP runtimes in one process cannot both own the device, and torch tensors / streams /
        # RCCL buffers are handed to this library as raw pointers, so torch is imported first: the
        # dynamic linker then resolves our DT_NEEDED libamdhip64.so.7 to the already-loaded copy.
        import torch  # noqa: F401
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)        # AttributeError here = header/library mismatch
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return lib


def check(rc):
    if rc != 0:
        msg = load().mi_last_error()
        raise RuntimeError("mi355_retrieval error %d: %s" % (rc, msg.decode() if msg else "?"))


def _strided(a):
    """(pointer, dtype code, row stride, col stride) of a 2-D float32/float64 array, in elements.
    Arrays whose strides are not whole, non-negative element multiples are copied once."""
    a = np.asarray(a)
    if a.ndim != 2:
        raise ValueError("expected a 2-D array")
    if a.dtype not in (np.float32, np.float64):
        a = a.astype(np.float64 if a.dtype.itemsize > 4 else np.float32)
    isz = a.dtype.itemsize
    if any(s % isz or s < 0 for s in a.strides):
        a = np.ascontiguousarray(a)
    code = MI_F32 if a.dtype == np.float32 else MI_F64
    return a, code, a.strides[0] // isz, a.strides[1] // isz


def _base_pointer(a):
    """Pointer to element (0,0) plus the lowest address of the strided block."""
    return a.ctypes.data


class Gallery:
    """One gallery row shard resident on one MI355X (a `mi_gallery` handle)."""

    def __init__(self, handle):
        self._h = C.c_void_p(handle)
        self._lock = threading.Lock()      # the handle is not re-entrant (Flask threads in online.py)
        n, d = C.c_int64(), C.c_int32()
        nm, dev = C.c_int32(), C.c_int32()
        off, hb = C.c_int64(), C.c_int64()
        check(load().mi_gallery_info(self._h, n, d, nm, dev, off, hb))
        self.n, self.d, self.norm_mode, self.device = n.value, d.value, nm.value, dev.value
        self.row_offset, self.hbm_bytes = off.value, hb.value

    # ---- construction
    @classmethod
    def from_host(cls, rows, norm_mode=NORM_L2, device=0, row_offset=0):
        """rows: [N, D] float32/float64, any strides (e.g. `vecs.T` of the reference's [D, N])."""
        a, code, rs, cs = _strided(rows)
        h = C.c_void_p()
        check(load().mi_gallery_create(C.c_void_p(_base_pointer(a)), a.shape[0], a.shape[1], code, rs, cs, MI_HOST,
                                       norm_mode, device, row_offset, C.byref(h)))
        return cls(h.value)

    @classmethod
    def from_device_ptr(cls, ptr, n, d, norm_mode=NORM_L2, device=0, row_offset=0, dtype=MI_F32, row_stride=None,
                        col_stride=1):
        """Device rows -> gallery, synchronous.  The ingest runs on a stream of the handle's own: the producer of `ptr` must
        have completed (torch.cuda.synchronize() / stream.synchronize()) before the call; `ptr` may be freed after it."""
        h = C.c_void_p()
        check(load().mi_gallery_create(C.c_void_p(ptr), n, d, dtype, d if row_stride is None else row_stride,
                                       col_stride, MI_DEVICE, norm_mode, device, row_offset, C.byref(h)))
        return cls(h.value)

    @classmethod
    def empty(cls, capacity, d, norm_mode=NORM_L2, device=0, row_offset=0):
        """Appendable gallery: `capacity` rows allocated, filled by append_device()."""
        h = C.c_void_p()
        check(load().mi_gallery_create_empty(capacity, d, norm_mode, device, row_offset, C.byref(h)))
        return cls(h.value)

    def append_device(self, rows_ptr, m, stream=None):
        """rows_ptr: device pointer to [m, d] float32 (C order)."""
        with self._lock:
            check(load().mi_gallery_append_device(self._h, C.c_void_p(rows_ptr), m, C.c_void_p(stream)))
            self.n += m

    def append_whitened_device(self, x_ptr, m, d, mean_ptr, p_ptr, dtype=MI_F32, row_stride=None, col_stride=1, stream=None):
        """m device rows [m, d] (strided; f32 | f64) -> P[:self.d] (x - mean), normalised by the gallery's own mode, appended.
        mean_ptr f64 [d], p_ptr f64 row-major [>= self.d, d], both on the device.  Synchronises `stream`."""
        with self._lock:
            check(load().mi_gallery_append_whitened_device(self._h, C.c_void_p(x_ptr), m, d, dtype,
                                                           d if row_stride is None else row_stride, col_stride,
                                                           C.c_void_p(mean_ptr), C.c_void_p(p_ptr), C.c_void_p(stream)))
            self.n += m

    def append(self, rows):
        """rows: [m, D] float32/float64 host array, any strides -- e.g. a column block `vecs[:, a:b].T` of the
        reference's [D, N] layout, which moves as one 2-D copy (no host transpose, no float64 promotion)."""
        a, code, rs, cs = _strided(rows)
        if a.shape[1] != self.d:
            raise ValueError("row dimension %d != gallery dimension %d" % (a.shape[1], self.d))
        with self._lock:
            check(load().mi_gallery_append(self._h, C.c_void_p(_base_pointer(a)), a.shape[0], code, rs, cs, MI_HOST))
            self.n += a.shape[0]

    @classmethod
    def from_blocks(cls, blocks, norm_mode=NORM_L2, device=0, row_offset=0, chunk_rows=131072):
        """Gallery of the concatenation `np.concatenate(blocks, axis=1).T` (src/test_rOP1m.py:136-139: [rOxford | 1M
        distractors]) WITHOUT building it on the host: every block is a [D, N_i] array (numpy, memory-mapped or a CPU
        torch tensor) and is appended in column chunks."""
        blocks = [b.numpy() if hasattr(b, "numpy") and not isinstance(b, np.ndarray) else np.asarray(b) for b in blocks]
        d = blocks[0].shape[0]
        if any(b.ndim != 2 or b.shape[0] != d for b in blocks):
            raise ValueError("blocks must be [D, N_i] arrays with one D")
        g = cls.empty(sum(b.shape[1] for b in blocks), d, norm_mode, device, row_offset)
        try:
            for b in blocks:
                for c0 in range(0, b.shape[1], chunk_rows):
                    g.append(b[:, c0:c0 + chunk_rows].T)
            if norm_mode == NORM_NONE and g.norm_bounds()[0] <= 4.0:
                g.set_image_dtype(_default_image_f16[0])      # raw rows turned out to sit inside fp16's comfortable range
        except Exception:
            g.close()
            raise
        return g

    @classmethod
    def load(cls, path, device=0):
        h = C.c_void_p()
        check(load().mi_gallery_load(os.fsencode(path), device, C.byref(h)))
        return cls(h.value)

    def save(self, path):
        check(load().mi_gallery_save(self._h, os.fsencode(path)))

    def close(self):
        """mi_gallery_destroy.  Refused (RuntimeError, the handle stays valid) while an online chain is built on the gallery."""
        if self._h is not None and self._h.value:
            rc = load().mi_gallery_destroy(self._h)
            if rc:
                check(rc)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- host API
    def search(self, queries, k):
        """-> (idx int64 [Q,k], scores float32 [Q,k], seconds)."""
        a, code, rs, cs = _strided(queries)
        if a.shape[1] != self.d:
            raise ValueError("query dimension %d != gallery dimension %d" % (a.shape[1], self.d))
        nq = a.shape[0]
        idx = np.empty((nq, k), dtype=np.int64)
        sc = np.empty((nq, k), dtype=np.float32)
        secs = C.c_double()
        with self._lock:
            check(load().mi_knn_search(self._h, C.c_void_p(_base_pointer(a)), nq, code, rs, cs, k,
                                       idx.ctypes.data_as(C.c_void_p), sc.ctypes.data_as(C.c_void_p), C.byref(secs)))
        return idx, sc, secs.value

    def aqe_search(self, ranks, k_qe, w, k, eps=1e-6, return_qexp=False):
        """ranks [K_in, Q] int64 (any strides) -> (idx [Q,k], scores [Q,k], qexp [Q,D] f64 | None, seconds)."""
        r = np.asarray(ranks)
        if r.dtype != np.int64:
            r = r.astype(np.int64)
        if any(s % 8 or s < 0 for s in r.strides):
            r = np.ascontiguousarray(r)
        nq = r.shape[1]
        idx = np.empty((nq, k), dtype=np.int64)
        sc = np.empty((nq, k), dtype=np.float32)
        qx = np.empty((nq, self.d), dtype=np.float64) if return_qexp else None
        secs = C.c_double()
        with self._lock:
            check(load().mi_aqe_search(self._h, r.ctypes.data_as(C.c_void_p), r.strides[0] // 8, r.strides[1] // 8,
                                       nq, k_qe, float(w), float(eps), k, idx.ctypes.data_as(C.c_void_p),
                                       sc.ctypes.data_as(C.c_void_p),
                                       qx.ctypes.data_as(C.c_void_p) if return_qexp else None, C.byref(secs)))
        return idx, sc, qx, secs.value

    def dense64_search(self, queries, k):
        """Every score of the gallery in float64 + exact top-k (no threshold logic): the independent checker of the filtered
        search.  -> (idx int64 [Q,k], scores float32 [Q,k], scores float64 [Q,k], seconds)."""
        a, code, rs, cs = _strided(queries)
        if a.shape[1] != self.d:
            raise ValueError("query dimension %d != gallery dimension %d" % (a.shape[1], self.d))
        nq = a.shape[0]
        idx = np.empty((nq, k), dtype=np.int64)
        sc = np.empty((nq, k), dtype=np.float32)
        sc64 = np.empty((nq, k), dtype=np.float64)
        secs = C.c_double()
        with self._lock:
            check(load().mi_knn_dense64_search(self._h, C.c_void_p(_base_pointer(a)), nq, code, rs, cs, k,
                                               idx.ctypes.data_as(C.c_void_p), sc.ctypes.data_as(C.c_void_p),
                                               sc64.ctypes.data_as(C.c_void_p), C.byref(secs)))
        return idx, sc, sc64, secs.value

    def dense_search(self, queries, k):
        """Exact f32 inner-product top-k for large k (k <= 4096): -> (idx [Q,k], scores [Q,k], seconds)."""
        a, code, rs, cs = _strided(queries)
        if a.shape[1] != self.d:
            raise ValueError("query dimension %d != gallery dimension %d" % (a.shape[1], self.d))
        nq = a.shape[0]
        idx = np.empty((nq, k), dtype=np.int64)
        sc = np.empty((nq, k), dtype=np.float32)
        secs = C.c_double()
        with self._lock:
            check(load().mi_knn_dense_search(self._h, C.c_void_p(_base_pointer(a)), nq, code, rs, cs, k,
                                             idx.ctypes.data_as(C.c_void_p), sc.ctypes.data_as(C.c_void_p),
                                             C.byref(secs)))
        return idx, sc, secs.value

    def rank_all(self, queries, query_norm=-1, return_scores=False):
        """Full-length ranking: -> (idx int64 [Q,N][, scores float32 [Q,N]], seconds)."""
        a, code, rs, cs = _strided(queries)
        if a.shape[1] != self.d:
            raise ValueError("query dimension %d != gallery dimension %d" % (a.shape[1], self.d))
        nq = a.shape[0]
        idx = np.empty((nq, self.n), dtype=np.int64)
        sc = np.empty((nq, self.n), dtype=np.float32) if return_scores else None
        secs = C.c_double()
        with self._lock:
            check(load().mi_rank_all(self._h, C.c_void_p(_base_pointer(a)), nq, code, rs, cs, query_norm,
                                     idx.ctypes.data_as(C.c_void_p),
                                     sc.ctypes.data_as(C.c_void_p) if return_scores else None, C.byref(secs)))
        return (idx, sc, secs.value) if return_scores else (idx, secs.value)

    def rank_prefix(self, queries, keep, query_norm=-1, return_scores=False):
        """The first `keep` columns of the full-length ranking: -> (idx int64 [Q,keep][, scores float32 [Q,keep]], seconds)."""
        a, code, rs, cs = _strided(queries)
        if a.shape[1] != self.d:
            raise ValueError("query dimension %d != gallery dimension %d" % (a.shape[1], self.d))
        nq, keep = a.shape[0], int(keep)
        idx = np.empty((nq, keep), dtype=np.int64)
        sc = np.empty((nq, keep), dtype=np.float32) if return_scores else None
        secs = C.c_double()
        with self._lock:
            check(load().mi_rank_prefix(self._h, C.c_void_p(_base_pointer(a)), nq, code, rs, cs, query_norm, keep,
                                        idx.ctypes.data_as(C.c_void_p),
                                        sc.ctypes.data_as(C.c_void_p) if return_scores else None, C.byref(secs)))
        return (idx, sc, secs.value) if return_scores else (idx, secs.value)

    def rank_positions(self, queries, row_ids, query_norm=-1):
        """Zero-based positions of the listed rows in every query's full ranking (score desc, idx asc), counted on the
        device.  row_ids int64 [Q, m] of global ids, -1 = padding -> position -1.  Returns int64 [Q, m]."""
        a, code, rs, cs = _strided(queries)
        if a.shape[1] != self.d:
            raise ValueError("query dimension %d != gallery dimension %d" % (a.shape[1], self.d))
        ids = np.ascontiguousarray(row_ids, dtype=np.int64)
        if ids.ndim != 2 or ids.shape[0] != a.shape[0]:
            raise ValueError("row_ids must be [Q, m]")
        out = np.empty(ids.shape, dtype=np.int64)
        with self._lock:
            check(load().mi_rank_positions(self._h, C.c_void_p(_base_pointer(a)), a.shape[0], code, rs, cs, query_norm,
                                           ids.ctypes.data_as(C.c_void_p), ids.shape[1], out.ctypes.data_as(C.c_void_p)))
        return out

    def diffusion_offline(self, n_trunc, kd, alpha=0.99, gamma=3, maxiter=20, tol=1e-6, return_sims=False):
        """-> (ids int64 [N,n_trunc], vals float32 [N,n_trunc][, knn sims float32 [N,n_trunc]])."""
        ids = np.empty((self.n, n_trunc), dtype=np.int64)
        vals = np.empty((self.n, n_trunc), dtype=np.float32)
        sims = np.empty((self.n, n_trunc), dtype=np.float32) if return_sims else None
        with self._lock:
            check(load().mi_diffusion_offline(self._h, n_trunc, kd, float(alpha), gamma, maxiter, float(tol),
                                              ids.ctypes.data_as(C.c_void_p), vals.ctypes.data_as(C.c_void_p),
                                              sims.ctypes.data_as(C.c_void_p) if return_sims else None))
        return (ids, vals, sims) if return_sims else (ids, vals)

    def diffusion_offline_nodes(self, n_trunc, kd, node0, node1, alpha=0.99, gamma=3, maxiter=20, tol=1e-6):
        """The offline rows of the nodes [node0, node1) only: -> (ids int64 [N,n_trunc], vals float32 [node1-node0,n_trunc])."""
        ids = np.empty((self.n, n_trunc), dtype=np.int64)
        vals = np.empty((max(0, node1 - node0), n_trunc), dtype=np.float32)
        with self._lock:
            check(load().mi_diffusion_offline_nodes(self._h, n_trunc, kd, float(alpha), gamma, maxiter, float(tol),
                                                    int(node0), int(node1), ids.ctypes.data_as(C.c_void_p),
                                                    vals.ctypes.data_as(C.c_void_p), None))
        return ids, vals

    def diffusion_set_offline(self, ids, vals):
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        vals = np.ascontiguousarray(vals, dtype=np.float32)
        with self._lock:
            check(load().mi_diffusion_set_offline(self._h, ids.ctypes.data_as(C.c_void_p),
                                                  vals.ctypes.data_as(C.c_void_p), ids.shape[1]))

    def diffusion_online(self, queries, k_query=3, gamma=3, trunc=2000):
        """-> (ranks int64 [Q,trunc], scores float32 [Q,trunc])."""
        a, code, rs, cs = _strided(queries)
        nq = a.shape[0]
        ranks = np.empty((nq, trunc), dtype=np.int64)
        sc = np.empty((nq, trunc), dtype=np.float32)
        with self._lock:
            check(load().mi_diffusion_online(self._h, C.c_void_p(_base_pointer(a)), nq, code, rs, cs, k_query, gamma,
                                             trunc, ranks.ctypes.data_as(C.c_void_p), sc.ctypes.data_as(C.c_void_p)))
        return ranks, sc

    def gather_weighted(self, ranks, weights):
        """ranks int64 [k, Q] (global ids), weights [k] -> float64 [Q, D] = sum_j weights[j] * row(ranks[j, q])."""
        r = np.ascontiguousarray(ranks, dtype=np.int64)
        w = np.ascontiguousarray(weights, dtype=np.float64)
        out = np.empty((r.shape[1], self.d), dtype=np.float64)
        with self._lock:
            check(load().mi_gather_weighted(self._h, r.ctypes.data_as(C.c_void_p), r.strides[0] // 8, r.strides[1] // 8,
                                            r.shape[1], r.shape[0], w.ctypes.data_as(C.c_void_p),
                                            out.ctypes.data_as(C.c_void_p)))
        return out

    def get_rows(self, row0, nrows):
        out = np.empty((nrows, self.d), dtype=np.float32)
        check(load().mi_gallery_get_rows(self._h, row0, nrows, out.ctypes.data_as(C.c_void_p)))
        return out

    # ---- device API (raw pointers; used by bench.py and the sharded path with torch tensors)
    def search_device(self, q_ptr, nq, k, idx_ptr, score_ptr=None, score64_ptr=None, stream=None):
        check(load().mi_knn_search_device(self._h, C.c_void_p(q_ptr), nq, k, C.c_void_p(idx_ptr),
                                          C.c_void_p(score_ptr), C.c_void_p(score64_ptr), C.c_void_p(stream)))

    def phase1_device(self, q_ptr, nq, k, approx_ptr, stream=None):
        check(load().mi_knn_phase1_device(self._h, C.c_void_p(q_ptr), nq, k, C.c_void_p(approx_ptr),
                                          C.c_void_p(stream)))

    def phase2_device(self, nq, k, L_ptr, idx_ptr, score_ptr, score64_ptr, stream=None):
        check(load().mi_knn_phase2_device(self._h, nq, k, C.c_void_p(L_ptr), C.c_void_p(idx_ptr),
                                          C.c_void_p(score_ptr), C.c_void_p(score64_ptr), C.c_void_p(stream)))

    def aqe_rows_device(self, ranks_ptr, stride_j, stride_q, nq, k_qe, rows_ptr, stream=None):
        check(load().mi_aqe_rows_device(self._h, C.c_void_p(ranks_ptr), stride_j, stride_q, nq, k_qe, C.c_void_p(rows_ptr),
                                        C.c_void_p(stream)))

    def aqe_partial_device(self, ranks_ptr, stride_j, stride_q, nq, k_qe, w, sum_ptr, stream=None):
        check(load().mi_aqe_partial_device(self._h, C.c_void_p(ranks_ptr), stride_j, stride_q, nq, k_qe, float(w),
                                           C.c_void_p(sum_ptr), C.c_void_p(stream)))

    # ---- instrumentation
    def set_option(self, name, value):
        check(load().mi_set_option(self._h, name.encode(), float(value)))

    def xcc_shares(self):
        """-> (float32 [8] shares of the XCD labels, launches that have updated them; -1 = no workspace yet)."""
        w = np.empty(8, dtype=np.float32)
        n = C.c_int32()
        check(load().mi_debug_xcc_shares(self._h, w.ctypes.data_as(C.c_void_p), C.byref(n)))
        return w, n.value

    def norm_bounds(self, raise_to=None):
        """{max ||g||, max ||g_hat||, max ||g_hat - g||} of this shard; raise_to: three floats -> the bounds become
        max(own, given) (agreement across the shards of one gallery, include/mi355_retrieval.h)."""
        b = (C.c_float * 3)(*(raise_to if raise_to is not None else (0.0, 0.0, 0.0)))
        check(load().mi_gallery_norm_bounds(self._h, b, 1 if raise_to is not None else 0))
        return [float(b[0]), float(b[1]), float(b[2])]

    def set_image_dtype(self, f16):
        check(load().mi_gallery_set_image_dtype(self._h, 1 if f16 else 0))

    def join(self, stream=None):
        """Asynchronous-tail mode: make `stream` wait for the re-score + sort of every search_device call made so far."""
        check(load().mi_search_join(self._h, C.c_void_p(stream)))

    def get_option(self, name):
        v = C.c_double(0.0)
        check(load().mi_get_option(self._h, name.encode(), C.byref(v)))
        return v.value

    def flags(self):
        """Synchronises and returns (then clears) the sticky device flags of the asynchronous entry points."""
        f = C.c_uint32(0)
        check(load().mi_search_flags(self._h, C.byref(f)))
        return f.value

    def calibrate(self, launches=8, stream=None):
        """Converges the tile kernel's per-XCD shares of the gallery before the first real search (mi_gallery_calibrate)."""
        check(load().mi_gallery_calibrate(self._h, launches, C.c_void_p(stream)))

    def profile(self, on=True):
        check(load().mi_profile_enable(self._h, 1 if on else 0))

    def debug_cycles(self, nseg=2048):
        out = np.zeros((nseg, 8), dtype=np.uint64)
        check(load().mi_debug_read_cycles(self._h, out.ctypes.data_as(C.c_void_p), out.size))
        return out

    def launch_ms(self, cap=65536):
        """Durations (ms) of the timed scoring launches since the last status(reset=True), in launch order."""
        out = np.zeros(cap, dtype=np.float32)
        n = C.c_int64(0)
        check(load().mi_profile_launch_ms(self._h, out.ctypes.data_as(C.c_void_p), cap, C.byref(n)))
        return out[:min(cap, n.value)].copy()

    def status(self, reset=False):
        st = SearchStats()
        check(load().mi_search_status(self._h, C.byref(st), 1 if reset else 0))
        return st.as_dict()


def kth_of_gathered_device(gathered_ptr, nshards, nq, k, out_ptr, stream=None):
    check(load().mi_kth_of_gathered_device(C.c_void_p(gathered_ptr), nshards, nq, k, C.c_void_p(out_ptr),
                                           C.c_void_p(stream)))


def topk_merge_strided_device(score64_ptr, idx_ptr, shard_stride, nshards, nq, k, out_idx_ptr, out_score_ptr, stream=None):
    check(load().mi_topk_merge_strided_device(C.c_void_p(score64_ptr), C.c_void_p(idx_ptr), shard_stride, nshards, nq, k,
                                              C.c_void_p(out_idx_ptr), C.c_void_p(out_score_ptr), C.c_void_p(stream)))


def topk_merge_device(score64_ptr, idx_ptr, nshards, nq, k, out_idx_ptr, out_score_ptr, stream=None):
    check(load().mi_topk_merge_device(C.c_void_p(score64_ptr), C.c_void_p(idx_ptr), nshards, nq, k,
                                      C.c_void_p(out_idx_ptr), C.c_void_p(out_score_ptr), C.c_void_p(stream)))


def aqe_combine_device(rows_ptr, nq, d, k_qe, w, sum_ptr, stream=None):
    check(load().mi_aqe_combine_device(C.c_void_p(rows_ptr), nq, d, k_qe, float(w), C.c_void_p(sum_ptr), C.c_void_p(stream)))


def aqe_finish_device(sum_ptr, nq, d, eps, out_q_ptr, out_q64_ptr=None, stream=None):
    check(load().mi_aqe_finish_device(C.c_void_p(sum_ptr), nq, d, float(eps), C.c_void_p(out_q_ptr),
                                      C.c_void_p(out_q64_ptr), C.c_void_p(stream)))


def synth_fill_device(dst_ptr, seed, row0, nrows, d, stream=None):
    check(load().mi_synth_fill_device(C.c_void_p(dst_ptr), seed, row0, nrows, d, C.c_void_p(stream)))


def column_sum(rows, device=0):
    """float64 column sums of a 2-D float array (any strides), computed on the device."""
    a, code, rs, cs = _strided(rows)
    out = np.empty(a.shape[1], dtype=np.float64)
    check(load().mi_column_sum(C.c_void_p(_base_pointer(a)), a.shape[0], a.shape[1], code, rs, cs, device,
                               out.ctypes.data_as(C.c_void_p)))
    return out


def whiten_apply(rows, m, P, dims, eps=1e-6, device=0):
    """rows [N,D] (any strides) -> float64 [N,dims] = rows of P[:dims] (x - m) / (||.|| + eps)."""
    a, code, rs, cs = _strided(rows)
    m = np.ascontiguousarray(np.asarray(m, dtype=np.float64).reshape(-1))
    Pd = np.ascontiguousarray(np.asarray(P, dtype=np.float64)[:dims, :])
    out = np.empty((a.shape[0], dims), dtype=np.float64)
    check(load().mi_whiten_apply(C.c_void_p(_base_pointer(a)), a.shape[0], a.shape[1], code, rs, cs,
                                 m.ctypes.data_as(C.c_void_p), Pd.ctypes.data_as(C.c_void_p), dims, float(eps), device,
                                 out.ctypes.data_as(C.c_void_p)))
    return out


def whiten_apply_device(x_ptr, n, d, m_ptr, p_ptr, dims, out_ptr, eps=1e-6, dtype=MI_F32, row_stride=None, col_stride=1,
                        stream=None):
    """Device-resident whitenapply (no synchronisation): out f64 [n, dims] = rows of P[:dims] (x - m) / (||.|| + eps);
    eps < 0 leaves the rows un-normalised."""
    check(load().mi_whiten_apply_device(C.c_void_p(x_ptr), n, d, dtype, d if row_stride is None else row_stride, col_stride,
                                        C.c_void_p(m_ptr), C.c_void_p(p_ptr), dims, float(eps), C.c_void_p(out_ptr),
                                        C.c_void_p(stream)))


def desc_tail_device(feat_ptr, b, c, hw, p, eps, w_ptr, b_ptr, c_out, scratch_ptr, out_ptr, stream=None):
    check(load().mi_desc_tail_device(C.c_void_p(feat_ptr), b, c, hw, float(p), float(eps), C.c_void_p(w_ptr),
                                     C.c_void_p(b_ptr), c_out, C.c_void_p(scratch_ptr), C.c_void_p(out_ptr),
                                     C.c_void_p(stream)))


def desc_ms_accumulate_device(acc_ptr, desc_ptr, count, msp, first, stream=None):
    check(load().mi_desc_ms_accumulate_device(C.c_void_p(acc_ptr), C.c_void_p(desc_ptr), count, float(msp),
                                              1 if first else 0, C.c_void_p(stream)))


def desc_ms_finish_device(acc_ptr, b, d, nscales, msp, stream=None):
    check(load().mi_desc_ms_finish_device(C.c_void_p(acc_ptr), b, d, nscales, float(msp), C.c_void_p(stream)))


def kr_rerank(queries, gallery, k1=20, k2=6, lambda_value=0.3, device=0, return_dist=False):
    """queries [Q, D], gallery [N, D] (any float strides) -> indices int64 [Q, N][, final distances float32 [Q, N]]."""
    q, code, qrs, qcs = _strided(queries)
    g, gcode, grs, gcs = _strided(gallery)
    if gcode != code:
        g = g.astype(q.dtype)
        g, gcode, grs, gcs = _strided(g)
    if q.shape[1] != g.shape[1]:
        raise ValueError("query dimension %d != gallery dimension %d" % (q.shape[1], g.shape[1]))
    idx = np.empty((q.shape[0], g.shape[0]), dtype=np.int64)
    dist = np.empty((q.shape[0], g.shape[0]), dtype=np.float32) if return_dist else None
    check(load().mi_kr_rerank(C.c_void_p(_base_pointer(q)), q.shape[0], qrs, qcs, C.c_void_p(_base_pointer(g)), g.shape[0],
                              grs, gcs, q.shape[1], code, int(k1), int(k2), float(lambda_value), int(device),
                              idx.ctypes.data_as(C.c_void_p), dist.ctypes.data_as(C.c_void_p) if return_dist else None))
    return (idx, dist) if return_dist else idx


_default_image_f16 = [1]


def set_global_option(name, value):
    if name == "image_dtype":
        _default_image_f16[0] = 1 if value else 0
    check(load().mi_set_global_option(name.encode(), float(value)))


def get_global_option(name):
    """"image_dtype", "host_ingest", "keep_buffers", "spare_bytes" (device memory held in the spare slots right now)."""
    v = C.c_double(0.0)
    check(load().mi_get_global_option(name.encode(), C.byref(v)))
    return v.value


def sample_source_rows(n, n_s=8192):
    """Diagnostics: gallery rows the bootstrap sample image of an n-row shard is drawn from."""
    import numpy as np
    f = load().mi_debug_sample_source_row
    return np.array([f(i, n, n_s) for i in range(n_s)], dtype=np.int64)


def device_count():
    n = C.c_int()
    check(load().mi_device_count(C.byref(n)))
    return n.value


class OnlineChain:
    """mi_online_*: search -> qge1 expansion -> re-search (src/online.py:108-152) behind one coalescing worker thread of the
    LIBRARY.  `query` blocks inside the C call -- ctypes releases the interpreter lock -- so the request threads of a Python
    server cost the interpreter one foreign call each.  g_rows None: the plain search.  The galleries must outlive it."""

    def __init__(self, g_search, g_rows, k, k_qe=3, w=4.0, eps=1e-6, max_batch=128, max_wait_us=500):
        h = C.c_void_p()
        check(load().mi_online_create(g_search._h, None if g_rows is None else g_rows._h, k, k_qe, w, eps, max_batch,
                                      max_wait_us, C.byref(h)))
        self._h, self.k, self.d = h.value, int(k), g_search.d
        self._keep = (g_search, g_rows)
        self._query = load().mi_online_query

    def query(self, ptr, nq, memspace, pending=False, producer_stream=None, scores=False):
        """ptr: [nq, d] float32 rows (host or device address); pending: device rows still being produced on producer_stream
        (None = the null stream).  -> idx int64 [nq, k] (, score float32 [nq, k]), host arrays."""
        if self._keep[0]._h is None or (self._keep[1] is not None and self._keep[1]._h is None):
            raise RuntimeError("the galleries of this online chain have been closed")
        idx = np.empty((nq, self.k), dtype=np.int64)
        sc = np.empty((nq, self.k), dtype=np.float32) if scores else None
        rc = self._query(self._h, ptr, nq, memspace, 1 if pending else 0, producer_stream,
                         idx.__array_interface__["data"][0], None if sc is None else sc.__array_interface__["data"][0])
        if rc:
            check(rc)
        return (idx, sc) if scores else idx

    def native_clients(self, desc_ptr, n_desc, threads, per_thread):
        """Diagnostics: `threads` request threads of the library in a closed loop (mi_debug_online_clients).
        -> (seconds, last answer of every thread int64 [threads, k])."""
        last = np.empty((threads, self.k), dtype=np.int64)
        sec = C.c_double()
        check(load().mi_debug_online_clients(self._h, desc_ptr, n_desc, threads, per_thread, last.ctypes.data, C.byref(sec)))
        return sec.value, last

    def stats(self):
        a, b = C.c_int64(), C.c_int64()
        check(load().mi_online_stats(self._h, C.byref(a), C.byref(b)))
        return {"chains": a.value, "requests": b.value}

    def close(self):
        if self._h:
            check(load().mi_online_destroy(self._h))
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
