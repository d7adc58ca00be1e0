#!/usr/bin/env python3
"""Headline benchmark: exhaustive top-100 retrieval, queries/sec (BASELINE.json metric).

Workload (BASELINE.json configs[2]): synthetic rOxford5k + 1M distractors gallery,
N = 1,005,994 x D = 2048 f32 descriptors (seeded counter-based generator, L2-normalised by the
ingest kernel), one step = one batch of 1024 queries -> exact top-100 per query.  Inputs are
resident in HBM when the timed region starts.  With --gpus N the SAME gallery is row-sharded over N
ranks (strong scaling; one process per GPU, RCCL all-gathers of the per-shard top-K).

Prints ONE JSON line on rank 0 (contract in the task description): metric/value/unit, `roofline`
(dominant kernel = the bf16 MFMA scoring kernel, HIP-event timed inside the timed region) and
`cpu_baseline` (the oracle's matching_L2 restatement timed on a bounded sample on the host).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (gallery rows, description)
    "roxford5k+1m": (1005994, "rOxford5k+1M distractors (synthetic 2048-d), HIP MFMA QxG^T + top-100"),
    "roxford5k": (4993, "rOxford5k-sized synthetic gallery"),
    "rparis6k+1m": (1007323, "rParis6k+1M distractors (synthetic 2048-d): BASELINE configs[4] with --with-aqe"),
    "10m": (10000000, "synthetic 10Mx2048 gallery"),
}
MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense bf16 = dense f16, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="roxford5k+1m", choices=sorted(WORKLOADS))
    ap.add_argument("--rows", type=int, default=0, help="override gallery rows")
    ap.add_argument("--dim", type=int, default=2048)
    ap.add_argument("--queries", type=int, default=1024)
    ap.add_argument("--topk", type=int, default=100)
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-rows", type=int, default=131072)
    ap.add_argument("--cpu-sample-queries", type=int, default=16)
    ap.add_argument("--option", action="append", default=[], help="name=value passed to mi_set_option")
    ap.add_argument("--image-dtype", default="f16", choices=["f16", "bf16"],
                    help="element type of the streamed 16-bit operand image (MFMA input type)")
    ap.add_argument("--with-aqe", action="store_true",
                    help="BASELINE configs[4]: every step = search + alpha-QE (k=3, w=4) re-search of the expanded queries")
    ap.add_argument("--async-tail", type=int, nargs="?", const=1, default=0, choices=[0, 1, 2],
                    help="single GPU: re-score + sort of batch i on the handle's second stream.  1: beside the scoring launch "
                         "of batch i+1 (measured: no gain -- the board is at its power cap and the scoring launch slows "
                         "down by what the overlap hides).  2: beside the query ingest + bootstrap of batch i+1 only; its "
                         "scoring launch waits for the tail.  Results of every batch are joined inside the timed region")
    ap.add_argument("--force-protocol", action="store_true",
                    help="one GPU: run the sharded two-phase protocol with its RCCL collectives on a group of ONE rank "
                         "(what a rank of a multi-GPU run executes, collectives included); diagnostic, not the headline")
    ap.add_argument("--layout", default="auto",
                    help="N GPUs as QGxRS: QG query groups (each answers 1/QG of every batch, no exchange between groups) times "
                         "RS row shards of the gallery per group (two-phase protocol with all-gathers inside the group); "
                         "isehr_amd.sharded.job_layout.  'auto': 2 x N/2 for an even N and batches of >= 512 queries, else "
                         "1 x N.  Measured per-rank steps of every layout: scripts/layout_model.sh")
    ap.add_argument("--pipeline", action="store_true",
                    help="sharded runs (--gpus N > 1 or --force-protocol): ShardedGallery.search_stream -- asynchronous "
                         "all-gathers, three batches in flight -- instead of one synchronous search per step")
    ap.add_argument("--diagnostic", action="store_true", help="skip result checks (ablation builds; number is NOT a result)")
    return ap.parse_args()


def cpu_baseline(gallery, q_host, n_total, args):
    """Oracle (numpy restatement of matching_L2, src/utils/nnsearch.py:687-706) on a bounded sample:
    the first `cpu_sample_rows` gallery rows and `cpu_sample_queries` queries; its cost is linear in
    the number of gallery rows, so queries/s at the full gallery = measured * sample_rows / N."""
    import numpy as np
    import oracle
    ns = min(args.cpu_sample_rows, gallery.n)
    nqs = min(args.cpu_sample_queries, q_host.shape[0])
    g = gallery.get_rows(0, ns)
    t0 = time.time()
    oracle.matching_l2(args.topk, g, q_host[:nqs])
    dt = time.time() - t0
    qps_sample = nqs / dt
    try:
        import threadpoolctl
        blas_threads = max([p.get("num_threads", 1) for p in threadpoolctl.threadpool_info()] or [1])
    except Exception:
        blas_threads = None
    # SURVEY.md 8d's second CPU baseline: the strongest fair exhaustive CPU path (multi-threaded sgemm + partial
    # selection = what faiss-CPU IndexFlatIP does; faiss itself is not installable offline), at the BLAS width numpy
    # has on this host, on the same row sample with more queries (a GEMM needs a batch to be fair to the CPU)
    nqb = min(1024, q_host.shape[0])
    gn = g / np.maximum(np.linalg.norm(g, axis=1, keepdims=True), 1e-30)
    qn = q_host[:nqb] / np.maximum(np.linalg.norm(q_host[:nqb], axis=1, keepdims=True), 1e-30)
    oracle.knn_flat_ip_blas(gn[:4096], qn, args.topk)          # BLAS warm-up
    t0 = time.time()
    oracle.knn_flat_ip_blas(gn, qn, args.topk)
    dtb = time.time() - t0
    blas = {"value": nqb / dtb * ns / n_total, "unit": "queries/s", "cores": blas_threads,
            "sample": "oracle.knn_flat_ip_blas (numpy sgemm + argpartition, f32) on the first %d gallery rows x %d "
                      "queries, K=%d: %.2f s; scaled by rows" % (ns, nqb, args.topk, dtb)}
    return {
        "value": qps_sample * ns / n_total, "unit": "queries/s", "cores": 1, "kind": "port", "blas": blas,
        "sample": "oracle.matching_l2 (numpy, f32, single thread like the reference) on the first %d of %d gallery "
                  "rows x %d queries, K=%d: %.2f s; scaled by rows (cost is linear in N)" % (ns, n_total, nqs,
                                                                                            args.topk, dt),
        "host_cpus": os.cpu_count(), "blas_threads": blas_threads, "numpy": np.__version__,
    }


def main():
    args = parse()
    # stdout carries the ONE JSON line and nothing else: libraries write banners to file descriptor 1 (RCCL prints its
    # version block there when a communicator is created, on every rank), so fd 1 is pointed at stderr for the whole run
    # and the JSON line goes to the saved descriptor at the end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist
    import isehr_amd  # noqa: F401
    from isehr_amd import _lib
    from isehr_amd.sharded import ShardedGallery, shard_bounds, job_layout, layout_groups

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # ISEHR_DIST_BACKEND=gloo + ISEHR_SHARE_GPU=1: rehearsal of the multi-rank flow with all ranks on one GPU
    # (RCCL refuses two ranks on one device); the real run is one rank per GPU over RCCL.
    backend = os.environ.get("ISEHR_DIST_BACKEND", "nccl")
    dev_index = 0 if os.environ.get("ISEHR_SHARE_GPU") == "1" else local_rank
    torch.cuda.set_device(dev_index)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
    if world == 1 and args.force_protocol:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29571")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", dev_index))
    dev = torch.device("cuda", dev_index)
    n_total = args.rows or WORKLOADS[args.workload][0]
    d, nq, k = args.dim, args.queries, args.topk
    # ---- layout of the job: gq query groups x gs row shards (gq * gs = world).  Every batch is split into gq slices of
    # queries; the gs ranks of a group shard the gallery rows among themselves and run the two-phase protocol inside the
    # group.  Groups never exchange anything: a query's answer lives with the group that computed it.
    try:
        gq, gs, qgroup, shard = job_layout(world, rank, nq, args.layout)
    except ValueError as e:
        raise SystemExit(str(e))
    group = layout_groups(gq, gs)[qgroup] if world > 1 else None
    nq_job, nq = nq, nq // gq                     # nq: queries THIS rank answers per step
    q_lo = qgroup * nq
    lo, hi = shard_bounds(n_total, gs, shard)
    stream = torch.cuda.current_stream().cuda_stream

    # ---- synthetic shard, generated on device (rows are a pure function of (seed, row))
    raw = torch.empty((hi - lo, d), dtype=torch.float32, device=dev)
    _lib.synth_fill_device(raw.data_ptr(), args.seed, lo, hi - lo, d, stream)
    torch.cuda.synchronize()
    _lib.set_global_option("image_dtype", 1 if args.image_dtype == "f16" else 0)
    t0 = time.time()
    gal = _lib.Gallery.from_device_ptr(raw.data_ptr(), hi - lo, d, norm_mode=_lib.NORM_L2, device=dev_index,
                                       row_offset=lo)
    torch.cuda.synchronize()
    ingest_s = time.time() - t0
    del raw
    torch.cuda.empty_cache()
    for opt in args.option:
        name, val = opt.split("=")
        gal.set_option(name, float(val))
    sg = ShardedGallery(gal, group=group, force_protocol=args.force_protocol)

    # a small pool of query batches (seeded), cycled over the steps
    pool = []
    for i in range(4):
        qb = torch.empty((nq, d), dtype=torch.float32, device=dev)        # this query group's slice of job batch i
        _lib.synth_fill_device(qb.data_ptr(), args.seed + 1 + i, q_lo, nq, d, stream)
        pool.append(qb)
    torch.cuda.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # single GPU: the exact re-score + sort of a batch runs on the handle's second stream beside the scoring launch of the
    # next batch (mi_set_option "async_tail"); every batch's results are complete after gal.join(), inside the timed region
    pipelined = world == 1 and args.async_tail > 0
    if pipelined:
        gal.set_option("async_tail", args.async_tail)

    def one_step(qb):
        idx_, sc_ = sg.search(qb, k, join=not pipelined or args.with_aqe)
        one_step.last_queries = qb
        if args.with_aqe:
            # ranks[K,Q] view of the [Q,K] result, like `ranks = match_idx.T` (src/test_rOP1m.py:157) -> QGE (N >= 120000)
            idx_, sc_, one_step.last_queries = sg.aqe_search(idx_.t(), 3, 4.0, k, join=not pipelined)
        return idx_, sc_

    use_stream = args.pipeline and sg._protocol and not args.with_aqe

    def run_steps(count):
        """`count` steps; returns the results of the last one"""
        if not use_stream:
            out_ = None
            for i in range(count):
                out_ = one_step(pool[i % len(pool)])
            return out_
        out_ = None
        for out_ in sg.search_stream((pool[i % len(pool)] for i in range(count)), k):
            pass
        one_step.last_queries = pool[(count - 1) % len(pool)]
        return out_

    if args.warmup:
        run_steps(args.warmup)
    if pipelined:
        gal.join(stream)
    barrier()
    gal.status(reset=True)
    gal.profile(True)
    barrier()
    t0 = time.perf_counter()
    idx, sc = run_steps(args.steps)
    if pipelined:
        gal.join(stream)
    barrier()
    elapsed = time.perf_counter() - t0
    gal.profile(False)
    st = gal.status(reset=True)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        ov = torch.tensor([st["overflow_batches"]], dtype=torch.int64, device=dev)
        dist.all_reduce(ov, op=dist.ReduceOp.SUM)
        overflow = int(ov.item())
    else:
        overflow = st["overflow_batches"]

    # result sanity on the last batch (size-independent properties; the oracle cannot run at this size)
    sc_h = sc.cpu().numpy()
    idx_h = idx.cpu().numpy()
    import numpy as np
    if not args.diagnostic:
        assert (np.diff(sc_h, axis=1) <= 0).all(), "scores not sorted"
        assert all(len(set(r)) == k for r in idx_h), "duplicate indices"
        assert idx_h.min() >= 0 and idx_h.max() < n_total
        # the returned scores are the exact cosines: recompute those of 16 queries in float64 from the stored rows
        # (each rank checks the rows of its own shard); with --with-aqe the last search ran on the expanded queries
        q_last = one_step.last_queries.double()
        if not args.with_aqe:
            q_last = q_last / q_last.norm(dim=1, keepdim=True)
        q_last = q_last.cpu().numpy()
        worst = 0.0
        for qi in range(0, nq, max(1, nq // 16)):
            mine = np.flatnonzero((idx_h[qi] >= lo) & (idx_h[qi] < hi))
            if len(mine):
                rows = np.stack([gal.get_rows(int(r) - lo, 1)[0] for r in idx_h[qi, mine]]).astype(np.float64)
                worst = max(worst, float(np.abs(rows @ q_last[qi] - sc_h[qi, mine]).max()))
        assert worst < 3e-7 * max(1.0, float(np.abs(sc_h).max())), "returned scores differ from the float64 re-computation: %g" % worst
    if overflow and not args.diagnostic:
        raise SystemExit("bench invalid: %d batches overflowed the candidate buffers" % overflow)

    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        gemm_s = st["gemm_ms"] * 1e-3
        # the scoring launch: MFMA-bound on the 256 x 256-tile kernel, HBM-bound (2*D bytes per gallery row) when the
        # batch is small enough for the streaming kernel (<= 128 queries, csrc/stream_select.hip)
        hbm_bound = nq <= 128
        if gemm_s > 0:
            achieved = st["gemm_bytes"] / gemm_s / 1e9 if hbm_bound else st["gemm_flops"] / gemm_s / 1e12
        else:
            achieved = None
        peak = HBM_PEAK_GBS if hbm_bound else MFMA_BF16_PEAK_TFLOPS
        traffic, traffic_src = None, None
        tp = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tp) and not hbm_bound and args.workload == "roxford5k+1m" and not args.rows and world == 1:
            try:
                tj = json.load(open(tp))
                traffic = tj.get("gemm_select_hbm_bytes_per_launch")
                traffic_src = "profiles/pmc_traffic.json (%s): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this " \
                              "command, not measured by this run" % tj.get("source", "?")
            except Exception:
                traffic = None
        clock = st.get("kernel_clock_mhz") or None
        roof = {"bound": "hbm" if hbm_bound else "mfma", "achieved": achieved, "peak": peak,
                "unit": "GB/s" if hbm_bound else "TFLOP/s", "frac": (achieved / peak) if achieved else None,
                "traffic": traffic, "traffic_source": traffic_src,
                "kernel": "stream_select_kernel" if hbm_bound else "gemm_tile_kernel", "launches": st["gemm_launches"],
                "avg_launch_ms": st["gemm_ms"] / max(1, st["gemm_launches"]),
                "kernel_share_of_step": gemm_s / elapsed}
        if clock and not hbm_bound:
            # the chip lowers its shader clock under MFMA load (DVFS): the dense peak it offers at the clock measured
            # INSIDE the timed launches (s_memtime / s_memrealtime, median over waves) next to the nominal 2.4 GHz peak
            roof["in_kernel_clock_mhz"] = clock
            roof["peak_at_clock"] = MFMA_BF16_PEAK_TFLOPS * clock / 2400.0
            roof["frac_at_clock"] = achieved / roof["peak_at_clock"] if achieved else None
        out = {
            "metric": "queries/sec", "value": nq_job * args.steps / elapsed, "unit": "queries/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": args.image_dtype,
            "data": "synthetic",
            "config": {"workload": WORKLOADS[args.workload][1] if not args.rows else "synthetic gallery",
                       "gallery_rows": n_total, "dim": d, "queries_per_step": nq_job, "topk": k,
                       "parallelism": ("row-shard x%d" % world if gq == 1 else
                                       "%d query groups (%d queries of every batch each, no exchange between groups) x %d "
                                       "row shards per group" % (gq, nq, gs)) +
                                      (" (two-phase protocol over RCCL forced on one rank)"
                                       if args.force_protocol and world == 1 else ""),
                       "alpha_qe": bool(args.with_aqe),
                       "collectives": ("asynchronous, three batches in flight (search_stream)" if use_stream else
                                       "synchronous per batch") if sg._protocol else None,
                       "tail": ("re-score + sort of batch i on a second stream beside the %s of batch i+1; joined inside "
                                "the timed region" % ("scoring launch" if args.async_tail == 1 else
                                                      "query ingest + bootstrap (not the scoring launch)"))
                               if pipelined else "same stream", "exact": "%s MFMA filter + f64 re-score certificate" % args.image_dtype,
                       "ingest_s": round(ingest_s, 3),
                       # matching_L2's own timer spans the normalisation of the gallery too (src/utils/nnsearch.py:688-705):
                       # the rate of ONE call that prepares the resident gallery and answers one batch
                       "queries_per_s_incl_gallery_ingest": nq / (ms_step * 1e-3 + ingest_s),
                       "candidates_per_query": st["candidates"] / max(1, st["queries"]),
                       "survivors_per_query": st["survivors"] / max(1, st["queries"]),
                       "score_check": "16 queries x top-%d re-computed in float64: max |d| %.2e" % (k, worst)
                                      if not args.diagnostic else None},
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(gal, pool[(args.steps - 1) % len(pool)].cpu().numpy(), n_total, args)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if world > 1 or (args.force_protocol and dist.is_initialized()):
        dist.barrier()
        dist.destroy_process_group()
    gal.close()


if __name__ == "__main__":
    main()
