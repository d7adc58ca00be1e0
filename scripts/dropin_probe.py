#!/usr/bin/env python3
"""What the reference's callers get: matching_HIP(K, vecs.T, qvecs.T[, dataset]) on a HOST [D, N] array at BASELINE
size (src/test_rOP1m.py:136-139,155-157), next to the rates of the box that bound it: pinned and pageable host-to-device
copies and the host's own memcpy rate at 1..16 threads.  Writes one JSON object to stdout.

    python scripts/dropin_probe.py [--rows 1005994] [--f64-cols 131072]
"""
import argparse
import json
import os
import shutil
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def host_rates(torch, np):
    out = {}
    n = 1 << 28                                   # 1 GiB of float32
    pinned = torch.empty(n, dtype=torch.float32, pin_memory=True)
    pinned.fill_(1.0)
    dev = torch.empty(n, dtype=torch.float32, device="cuda")
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dev.copy_(pinned, non_blocking=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    out["pinned_h2d_GBps"] = n * 4 / dt / 1e9
    pageable = np.ones(n, dtype=np.float32)
    tp = torch.from_numpy(pageable)
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dev.copy_(tp)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    out["pageable_h2d_GBps"] = n * 4 / dt / 1e9
    dst = pinned.numpy()
    rates = {}
    for nt in (1, 2, 4, 8, 16):
        step = n // nt

        def work(i):
            np.copyto(dst[i * step:(i + 1) * step], pageable[i * step:(i + 1) * step])
        best = None
        for _ in range(3):
            th = [threading.Thread(target=work, args=(i,)) for i in range(nt)]
            t0 = time.perf_counter()
            for t in th:
                t.start()
            for t in th:
                t.join()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        rates[str(nt)] = n * 4 / best / 1e9
    out["host_memcpy_pageable_to_pinned_GBps_by_threads"] = rates
    out["host_cpus"] = os.cpu_count()
    try:
        out["sched_affinity"] = len(os.sched_getaffinity(0))
    except Exception:
        pass
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1005994)
    ap.add_argument("--f64-cols", type=int, default=131072)
    ap.add_argument("--queries", type=int, default=70)
    ap.add_argument("--no-rates", action="store_true")
    args = ap.parse_args()
    import numpy as np
    import torch
    import isehr_amd  # noqa: F401
    from isehr_amd import _lib, nnsearch
    out = {"rows": args.rows, "queries": args.queries}
    if not args.no_rates:
        out["rates"] = host_rates(torch, np)
    d, n, nq, k = 2048, args.rows, args.queries, 100
    stream = torch.cuda.current_stream().cuda_stream
    # host [D, N] float32 like np.concatenate([vecs, vecs_1m], axis=1): generated on the device row by row, transposed there
    vecs = np.empty((d, n), dtype=np.float32)
    blk = 131072
    for r0 in range(0, n, blk):
        m = min(blk, n - r0)
        raw = torch.empty((m, d), dtype=torch.float32, device="cuda")
        _lib.synth_fill_device(raw.data_ptr(), 1234, r0, m, d, stream)
        vecs[:, r0:r0 + m] = raw.t().contiguous().cpu().numpy()
    qraw = torch.empty((nq, d), dtype=torch.float32, device="cuda")
    _lib.synth_fill_device(qraw.data_ptr(), 1235, 0, nq, d, stream)
    qvecs = np.ascontiguousarray(qraw.t().cpu().numpy())
    del raw, qraw
    torch.cuda.empty_cache()
    nnsearch.matching_HIP(k, vecs[:, :4096].T, qvecs.T)                        # library initialised
    calls = {}
    t0 = time.time()
    idx0, tpq = nnsearch.matching_HIP(k, vecs.T, qvecs.T)
    calls["stateless"] = {"time_per_query_s": tpq, "wall_s": time.time() - t0}
    t0 = time.time()
    idx0b, tpq = nnsearch.matching_HIP(k, vecs.T, qvecs.T)
    calls["stateless_again"] = {"time_per_query_s": tpq, "wall_s": time.time() - t0}
    shutil.rmtree(os.path.join("outputs", "dropin_probe"), ignore_errors=True)
    t0 = time.time()
    idx1, tpq = nnsearch.matching_HIP(k, vecs.T, qvecs.T, dataset="dropin_probe", ifgenerate=True)
    calls["dataset_first"] = {"time_per_query_s": tpq, "wall_s": time.time() - t0}
    t0 = time.time()
    idx2, tpq = nnsearch.matching_HIP(k, vecs.T, qvecs.T, dataset="dropin_probe")
    calls["dataset_cached"] = {"time_per_query_s": tpq, "wall_s": time.time() - t0}
    nnsearch.drop_cached_galleries()
    t0 = time.time()
    idx3, tpq = nnsearch.matching_HIP(k, vecs.T, qvecs.T, dataset="dropin_probe")
    calls["dataset_from_file"] = {"time_per_query_s": tpq, "wall_s": time.time() - t0}
    nnsearch.drop_cached_galleries()
    shutil.rmtree(os.path.join("outputs", "dropin_probe"), ignore_errors=True)
    out["same_answers"] = bool((idx0 == idx1).all() and (idx0 == idx2).all() and (idx0 == idx3).all() and (idx0 == idx0b).all())
    out["f32_host_DxN"] = calls
    out["bytes_f32"] = int(vecs.nbytes)
    # the float64 case of src/online.py:96 (np.empty((2048, 0)) promotes the concatenation)
    v64 = vecs[:, :args.f64_cols].astype(np.float64)
    t0 = time.time()
    i64, tpq = nnsearch.matching_HIP(k, v64.T, qvecs.T.astype(np.float64))
    out["f64_host_DxN"] = {"cols": args.f64_cols, "time_per_query_s": tpq, "wall_s": time.time() - t0,
                           "bytes": int(v64.nbytes)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
