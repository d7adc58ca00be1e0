#!/usr/bin/env python3
"""Headline benchmark: exhaustive top-100 retrieval, queries/sec (BASELINE.json metric).

Workload (BASELINE.json configs[2]): synthetic rOxford5k + 1M distractors gallery,
N = 1,005,994 x D = 2048 f32 descriptors (seeded counter-based generator, L2-normalised by the
ingest kernel), one step = one batch of 1024 queries -> exact top-100 per query.  Inputs are
resident in HBM when the timed region starts.

`--gpus N` is the number of ranks of the job, one process per GPU over RCCL.  Started bare
(`python bench.py --gpus N`, no WORLD_SIZE in the environment) this file starts the N rank processes
itself -- `python -m torch.distributed.run --nproc-per-node N bench.py <same arguments>` as a CHILD
process, before torch is imported or the GPU touched -- relays their output and exits with their
code.  Started by a launcher (WORLD_SIZE set) it is one rank; WORLD_SIZE != --gpus is an error.
The SAME gallery is sharded over the ranks (strong scaling).

Prints ONE JSON line on rank 0 (contract in the task description): metric/value/unit, `roofline`
(dominant kernel = the 16-bit MFMA scoring kernel, HIP-event timed inside the timed region),
`cpu_baseline` (the oracle's matching_L2 restatement timed on a bounded sample on the host) and
`scale_10m`: BASELINE configs[3] -- a synthetic 10,000,000 x 2048 gallery with a bf16 image,
row-sharded 1 x N over the same ranks with the RCCL all-gathers of the per-shard top-K -- measured
in the same process after the headline (10 steps).
"""
import argparse
import json
import os
import subprocess
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
HBM_ACHIEVABLE_GBS = 6300.0      # measured float4 copy, same guide
PG_TIMEOUT_S = 90                # process-group timeout: a rank that never arrives fails the job after this, not after 10 min
# seconds a block may take before rank 0 prints the line with {"error": "timeout"} for it and the job exits non-zero
# (ISEHR_BENCH_DEADLINE_S overrides every one of them: tests)
DEADLINES = {"headline": 420, "synchronous": 120, "cpu_baseline": 240, "row_shard_1xN": 150, "batch_replicas": 150,
             "scale_10m": 300, "map": 120, "q1": 60, "q70": 60, "aqe_rparis_1m": 150, "qge_small": 120, "dropin": 240,
             "online": 60, "pipelined_collectives": 150, "shutdown": 60, "qsweep": 90, "hard_data": 240, "whiten": 120,
             "online_concurrent": 90}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1,
                    help="ranks of the job (one per GPU).  Without WORLD_SIZE in the environment and N > 1 this process "
                         "starts the N ranks itself (torch.distributed.run as a child process)")
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
    ap.add_argument("--async-tail", type=int, nargs="?", const=1, default=None, choices=[0, 1, 2, 3],
                    help="single GPU: where the exact re-score + final order of batch i run.  0: on the caller's stream "
                         "(synchronous).  3: deferred -- enqueued by the call of batch i+1 right before its scoring launch, "
                         "on the handle's own stream beside that launch (the throughput mode; +2..7 %% queries/s, the scoring "
                         "launch slows down by ~0.1 ms).  1: from the end of phase 1 of batch i.  2: beside the query ingest + "
                         "bootstrap of batch i+1 only.  Results of every batch are joined inside the timed region.  Default: "
                         "3 for the headline of a one-GPU run with batches of > 128 queries (the line then also carries a "
                         "`synchronous` block, and `roofline` is quoted on ITS undisturbed launches), else 0")
    ap.add_argument("--calibrate", type=int, default=16,
                    help="scoring launches of mi_gallery_calibrate after the ingest (the XCD shares converge within ~4; the rest "
                         "carries the chip through the 15-20 slow launches that follow the light-load ingest phase, "
                         "scripts/gap_probe.py; 0 = off)")
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
    ap.add_argument("--graph", action="store_true",
                    help="one GPU, diagnostic: capture the launch sequence of one search per query batch in a hipGraph "
                         "(torch.cuda.CUDAGraph over the library's launches on the capture stream) and time graph replays "
                         "instead of eager launches.  No per-launch events exist inside a graph: `roofline.achieved` is null")
    ap.add_argument("--scale-10m", default="auto", choices=["auto", "on", "off"],
                    help="secondary block `scale_10m` (BASELINE configs[3]: 10 M x 2048 rows, bf16 image, row-sharded 1 x N "
                         "over the job's ranks), measured after the headline.  auto: on for the default workload without "
                         "--rows / --queries / --dim overrides and without diagnostic modes")
    ap.add_argument("--multi-gpu-blocks", default="auto", choices=["auto", "on", "off"],
                    help="N > 1: the `row_shard_1xN` block (north_star's partition of the benchmarked gallery: row shards only, "
                         "synchronous and with pipelined collectives, per-stage timings).  auto: on for the default workload")
    ap.add_argument("--scale-10m-steps", type=int, default=10)
    ap.add_argument("--scale-10m-rows", type=int, default=10000000)
    ap.add_argument("--extra-blocks", default="auto", choices=["auto", "off"],
                    help="one GPU, default workload: the blocks that give every BASELINE config a number in the line -- `q70`, `q1` "
                         "(HBM-bound small batches on the resident gallery), `aqe_rparis_1m` (configs[4]), `qge_small` (the "
                         "N < 120 000 diffusion branch at rOxford5k size), `dropin` (matching_HIP on a host [D, N] array at "
                         "BASELINE size: what the reference's entry points call)")
    ap.add_argument("--blocks", default="",
                    help="comma-separated names: run only these blocks behind the headline (q70, q1, qsweep, online, "
                         "online_concurrent, hard_data, whiten, cpu_baseline, scale_10m, map, aqe_rparis_1m, qge_small, dropin); "
                         "default: all of them.  For iterating on one block on a GPU box")
    ap.add_argument("--diagnostic", action="store_true", help="skip result checks (ablation builds; number is NOT a result)")
    return ap.parse_args()


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (one per GPU) as a CHILD process group
    -- this process has not imported torch nor touched the GPU, and it does not exec -- pass their output through (rank 0
    prints the one JSON line) and return their exit code."""
    import uuid
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    # c10d rendezvous on 127.0.0.1 with port 0: the store binds a free port ITSELF (what --standalone does, minus the host
    # name lookup) -- no window between picking a port and using it, so concurrent bench runs cannot collide
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--rdzv-backend", "c10d", "--rdzv-endpoint", "127.0.0.1:0", "--rdzv-id", uuid.uuid4().hex,
           "--local-addr", "127.0.0.1", os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return None


def cpu_baseline(gallery, q_host, n_total, args):
    """Oracle (numpy restatement of matching_L2, src/utils/nnsearch.py:687-706) on a bounded sample:
    the first `cpu_sample_rows` gallery rows and `cpu_sample_queries` queries; its cost is linear in
    the number of gallery rows, so queries/s at the full gallery = measured * sample_rows / N.
    Beside it the BLAS exhaustive baseline -- sgemm + partial selection, the stand-in for
    faiss-CPU IndexFlatIP -- at Q in {1, 70, 1024}, at numpy's full BLAS width and on one thread."""
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
        threadpoolctl, blas_threads = None, None
    gn = g / np.maximum(np.linalg.norm(g, axis=1, keepdims=True), 1e-30)
    qall = q_host / np.maximum(np.linalg.norm(q_host, axis=1, keepdims=True), 1e-30)
    oracle.knn_flat_ip_blas(gn[:4096], qall, args.topk)          # BLAS warm-up

    def blas_point(nqb, rows, threads):
        nqb = min(nqb, qall.shape[0])
        qn = np.ascontiguousarray(qall[:nqb])
        gs = gn[:rows]
        reps = 3 if nqb <= 128 else 1
        best = None
        for _ in range(reps):
            t0_ = time.time()
            if threads is not None and threadpoolctl is not None:
                with threadpoolctl.threadpool_limits(limits=threads):
                    oracle.knn_flat_ip_blas(gs, qn, args.topk)
            else:
                oracle.knn_flat_ip_blas(gs, qn, args.topk)
            dt_ = time.time() - t0_
            best = dt_ if best is None else min(best, dt_)
        return {"queries": nqb, "threads": threads if threads is not None else blas_threads, "sample_rows": int(rows),
                "seconds": round(best, 4), "value": nqb / best * rows / n_total, "unit": "queries/s"}

    points = [blas_point(1024, ns, None), blas_point(70, ns, None), blas_point(1, ns, None)]
    if threadpoolctl is not None:
        # one thread: a quarter of the rows for the 1024-query GEMM keeps the leg at a few seconds
        points += [blas_point(1024, max(4096, ns // 4), 1), blas_point(70, ns, 1), blas_point(1, ns, 1)]
    full = points[0]
    # B4: alpha-QE on the CPU = the expansion of feature_enhancement (src/utils/Reranking.py:195-204:
    # weighted sum of the top-3 rows, w = 4, eps-normalised; float64 like the reference's promotion) followed by B2 on the
    # expanded queries -- what one alpha-QE step costs on top of the first search, at numpy's full BLAS width
    _, ids1 = oracle.knn_flat_ip_blas(gn, qall, 3)
    t0 = time.time()
    wts = oracle.qe_weights(3, 4.0)
    qx = (gn[ids1].astype(np.float64) * wts[None, :, None]).sum(axis=1)
    qx /= (np.linalg.norm(qx, axis=1, keepdims=True) + 1e-6)
    oracle.knn_flat_ip_blas(gn, qx.astype(np.float32), args.topk)
    dt_aqe = time.time() - t0
    aqe = {"value": qall.shape[0] / dt_aqe * ns / n_total, "unit": "queries/s", "cores": blas_threads,
           "sample": "alpha-QE step (k=3, w=4: expansion + BLAS re-search, oracle.qe_weights / knn_flat_ip_blas) of %d queries on "
                     "the first %d gallery rows: %.2f s; scaled by rows" % (qall.shape[0], ns, dt_aqe)}
    blas = {"value": full["value"], "unit": "queries/s", "cores": blas_threads,
            "sample": "oracle.knn_flat_ip_blas (numpy sgemm + argpartition, f32) on the first %d gallery rows x %d "
                      "queries, K=%d: %.2f s; scaled by rows" % (ns, full["queries"], args.topk, full["seconds"]),
            "points": points}
    try:
        import faiss  # noqa: F401
        faiss_note = "importable on this box but not timed (the reference pins no version: requirements.txt:4 is commented out)"
    except Exception as e:
        faiss_note = "unavailable (%s: BASELINE.md section 3 B3 cannot run; B2 = sgemm + argpartition below is its stand-in)" \
                     % type(e).__name__
    return {
        "value": qps_sample * ns / n_total, "unit": "queries/s", "cores": 1, "kind": "port", "blas": blas, "aqe": aqe,
        # the strongest CPU exhaustive baseline beside the one-thread port of the reference's path
        "value_blas": blas["value"], "cores_blas": blas_threads, "faiss": faiss_note,
        "sample": "oracle.matching_l2 (numpy, f32, single thread like the reference) on the first %d of %d gallery "
                  "rows x %d queries, K=%d: %.2f s; scaled by rows (cost is linear in N)" % (ns, n_total, nqs,
                                                                                            args.topk, dt),
        "cpu_model": _cpu_model(), "host_cpus": os.cpu_count(), "blas_threads": blas_threads, "numpy": np.__version__,
    }


class Emitter:
    """Rank 0's JSON line and the deadlines of the blocks that fill it.

    The line is written as soon as the headline is measured (`"complete": false`) and again, complete, at the end -- the driver
    parses the LAST line, so a secondary block that hangs or dies can no longer cost the headline.  Every block runs under a
    deadline: when it expires, rank 0 writes the line with `{"error": "timeout"}` for that block and EVERY rank leaves with
    os._exit(3) from its own watchdog thread (the main thread may be inside a collective that will never complete; a launcher
    sees a non-zero child and ends the others).  Nothing here re-executes a process that touched the GPU."""

    def __init__(self, fd, rank):
        import threading
        self.fd, self.rank, self.out = fd, rank, None
        self._threading = threading
        self._timer = None
        self._lock = threading.Lock()
        self.exit_code = 0

    def write(self, complete):
        if self.rank != 0 or self.out is None:
            return
        with self._lock:
            self.out["complete"] = bool(complete)
            os.write(self.fd, (json.dumps(self.out) + "\n").encode())

    def _expire(self, name, seconds):
        sys.stderr.write("bench.py: block %r exceeded its deadline of %d s on rank %d\n" % (name, seconds, self.rank))
        sys.stderr.flush()
        if self.rank == 0:
            if self.out is None:
                self.out = {"metric": "queries/sec", "value": None, "unit": "queries/s"}
            self.out[name if name != "headline" else "headline_error"] = {"error": "timeout", "deadline_s": seconds}
            self.out["incomplete_because"] = "block %r hit its deadline" % name
            self.write(False)
        os._exit(3)

    def deadline(self, name):
        emitter = self
        seconds = int(os.environ.get("ISEHR_BENCH_DEADLINE_S", DEADLINES.get(name, 120)))

        class _Ctx:
            def __enter__(self_):
                t = emitter._threading.Timer(seconds, emitter._expire, (name, seconds))
                t.daemon = True
                emitter._timer = t
                t.start()
                return self_

            def __exit__(self_, *exc):
                emitter._timer.cancel()
                return False
        return _Ctx()

    def block(self, name, fn, world, fatal_types=()):
        """Runs fn() -> value under the block's deadline.  An exception becomes {"error": ...} in the line; with several ranks
        it also ends the job (exit code 4 after the line is written: a rank that failed alone would leave the others in a
        collective), with one rank the remaining blocks still run."""
        try:
            with self.deadline(name):
                return fn()
        except (RuntimeError, AssertionError, MemoryError, SystemExit, ValueError, OSError) as e:
            sys.stderr.write("bench.py: block %r failed on rank %d: %s: %s\n" % (name, self.rank, type(e).__name__, e))
            if self.out is not None:
                self.out[name] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
            if world > 1:
                self.write(False)
                os._exit(4)
            return None


class Job:
    """The process-wide state of one rank: device, process group, stream."""

    def __init__(self, args):
        import torch
        import torch.distributed as dist
        self.args = args
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        # ISEHR_DIST_BACKEND=gloo + ISEHR_SHARE_GPU=1: rehearsal of the multi-rank flow with all ranks on one GPU
        # (RCCL refuses two ranks on one device); the real run is one rank per GPU over RCCL.
        self.backend = os.environ.get("ISEHR_DIST_BACKEND", "nccl")
        self.dev_index = 0 if os.environ.get("ISEHR_SHARE_GPU") == "1" else local_rank
        from datetime import timedelta
        torch.cuda.set_device(self.dev_index)
        pg_timeout = timedelta(seconds=int(os.environ.get("ISEHR_BENCH_PG_TIMEOUT_S", PG_TIMEOUT_S)))
        if self.world > 1:
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            if self.backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", self.dev_index), timeout=pg_timeout)
            else:
                dist.init_process_group(self.backend, timeout=pg_timeout)
        if self.world == 1 and args.force_protocol:
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29571")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", self.dev_index),
                                    timeout=pg_timeout)
        self.dev = torch.device("cuda", self.dev_index)
        self.stream = torch.cuda.current_stream().cuda_stream
        # what the communicator itself says (the answer to "did RCCL see N ranks")
        self.comm_ranks = dist.get_world_size() if dist.is_initialized() else 1
        self.comm_backend = dist.get_backend() if dist.is_initialized() else None
        # ... and which device every rank really runs on: (host, torch.cuda.current_device(), PCI bus id) of all ranks
        props = torch.cuda.get_device_properties(self.dev_index)
        me = (os.uname().nodename, int(torch.cuda.current_device()),
              str(getattr(props, "pci_bus_id", "")) + ":" + str(getattr(props, "pci_device_id", "")))
        self.rank_devices = [me]
        if self.world > 1:
            got = [None] * self.world
            dist.all_gather_object(got, me)
            self.rank_devices = [tuple(x) for x in got]
        self.devices_distinct = len(set(self.rank_devices)) == len(self.rank_devices)
        if self.world > 1 and self.backend == "nccl" and os.environ.get("ISEHR_SHARE_GPU") != "1":
            if self.comm_ranks != args.gpus or not self.devices_distinct:
                sys.stderr.write("bench.py: %d ranks in the communicator for --gpus %d, devices %r\n"
                                 % (self.comm_ranks, args.gpus, self.rank_devices))
                sys.exit(2)

    def barrier(self):
        import torch
        import torch.distributed as dist
        if self.world > 1:
            dist.barrier()
        torch.cuda.synchronize()


def run_workload(job, n_total, nq_job, image_dtype, steps, warmup, layout, with_aqe=False, async_tail=0, pipeline=False,
                 options=(), check=True, warm_ingest=False, keep=False, graph=False, also_stream=False, phases=0,
                 defer_extras=False):
    """Ingests this rank's shard of an n_total-row synthetic gallery and times `steps` steps of nq_job queries.
    Returns a dict of measurements (+ the gallery and the last query batch when keep=True).
    also_stream: sharded runs -- after the synchronous timed region, time the same number of steps through
    ShardedGallery.search_stream (asynchronous all-gathers, three batches in flight) on the same gallery: res["stream"].
    phases: sharded runs -- that many searches with an event behind every stage of the protocol (search_timed); the means,
    maximised over the ranks, in res["phases"]."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from isehr_amd import _lib
    from isehr_amd.sharded import ShardedGallery, shard_bounds, job_layout, layout_groups
    args, world, rank, dev, stream = job.args, job.world, job.rank, job.dev, job.stream
    d, k = args.dim, args.topk
    # ---- layout of the job: gq query groups x gs row shards (gq * gs = world).  Every batch is split into gq slices of
    # queries; the gs ranks of a group shard the gallery rows among themselves and run the two-phase protocol inside the
    # group.  Groups never exchange anything: a query's answer lives with the group that computed it.
    # layout "replicas": every rank holds the WHOLE gallery (288 GB of HBM per GPU: both BASELINE galleries fit one GPU) and the
    # job's steps are dealt to the ranks -- rank r answers batches r, r + N, ... with the one-GPU pipeline, nothing is exchanged
    replicas = layout == "replicas" and world > 1
    job_steps = steps
    try:
        gq, gs, qgroup, shard = (1, 1, 0, 0) if replicas else job_layout(world, rank, nq_job, layout)
    except ValueError as e:
        raise SystemExit(str(e))
    if replicas:
        group = layout_groups(world, 1)[rank]      # one-rank groups: no collective anywhere on the search path
        steps = steps // world + (1 if rank < steps % world else 0)
    else:
        group = layout_groups(gq, gs)[qgroup] if world > 1 else None
    nq = nq_job // gq                             # queries THIS rank answers per step
    q_lo = qgroup * nq
    lo, hi = shard_bounds(n_total, gs, shard)

    # ---- synthetic shard, generated on device (rows are a pure function of (seed, row))
    raw = torch.empty((hi - lo, d), dtype=torch.float32, device=dev)
    _lib.synth_fill_device(raw.data_ptr(), args.seed, lo, hi - lo, d, stream)
    torch.cuda.synchronize()
    _lib.set_global_option("image_dtype", 1 if image_dtype == "f16" else 0)
    alloc_s = None
    ingest_kernel_s = None
    ingest_launch_ms = None
    stateless_call_s = None
    t0 = time.time()
    if warm_ingest or (hi - lo) * d * 4 <= 16 << 30:
        gal = _lib.Gallery.from_device_ptr(raw.data_ptr(), hi - lo, d, norm_mode=_lib.NORM_L2, device=job.dev_index,
                                           row_offset=lo)
        torch.cuda.synchronize()
    else:
        # large shards (the 10 M-row block: 82 GB of rows -> 123 GB of gallery): allocation and ingest timed apart.  The
        # allocation is first-touch page-table work for ~123 GB (seconds, differs between boxes by a factor of ten); the
        # ingest kernel is the same persistent kernel as above
        gal = _lib.Gallery.empty(hi - lo, d, norm_mode=_lib.NORM_L2, device=job.dev_index, row_offset=lo)
        torch.cuda.synchronize()
        alloc_s = time.time() - t0
        t1 = time.time()
        gal.append_device(raw.data_ptr(), hi - lo, stream)
        torch.cuda.synchronize()
        ingest_kernel_s = time.time() - t1
    ingest_first_s = time.time() - t0
    ingest_s = ingest_first_s
    if warm_ingest:
        # the first ingest of a process also initialises the library (code objects, workspaces): the reference-style
        # per-call normalisation (matching_L2 normalises the gallery inside its timer, src/utils/nnsearch.py:688-705) is
        # what a SECOND ingest of the same rows costs
        # (twice: the first of the two maps 12 GB of fresh device memory, ~5 ms of page-table work; a caller that prepares
        # a gallery per call -- create, search, destroy -- gets the block of the previous call back, which is the second)
        for _ in range(2):
            t0 = time.time()
            g2 = _lib.Gallery.from_device_ptr(raw.data_ptr(), hi - lo, d, norm_mode=_lib.NORM_L2, device=job.dev_index,
                                              row_offset=lo)
            torch.cuda.synchronize()
            ingest_s = min(ingest_s, time.time() - t0)
            g2.close()
        # and the device work alone: the same rows appended to a gallery whose buffers exist already (what hipMalloc costs
        # differs between boxes and states of the driver by a factor of ten; the kernels do not)
        # -- what a stateless caller's timer spans (matching_L2 normalises the gallery inside every call): create + ONE batch +
        # destroy on the resident rows, wall clock, device-synchronised, best of three
        qtmp = torch.empty((nq_job // gq, d), dtype=torch.float32, device=dev)
        _lib.synth_fill_device(qtmp.data_ptr(), args.seed + 1, 0, qtmp.shape[0], d, stream)
        itmp = torch.empty((qtmp.shape[0], args.topk), dtype=torch.int64, device=dev)
        stmp = torch.empty((qtmp.shape[0], args.topk), dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        for _ in range(3):
            t0 = time.time()
            g2 = _lib.Gallery.from_device_ptr(raw.data_ptr(), hi - lo, d, norm_mode=_lib.NORM_L2, device=job.dev_index,
                                              row_offset=lo)
            g2.search_device(qtmp.data_ptr(), qtmp.shape[0], args.topk, itmp.data_ptr(), stmp.data_ptr(), None, stream)
            torch.cuda.synchronize()
            g2.close()
            stateless_call_s = min(stateless_call_s or 1e9, time.time() - t0)
        del qtmp, itmp, stmp
        # -- the same rows appended to three galleries back to back, each launch between HIP events on the launch stream.  The
        # same launch takes 3.5-4.1 ms on one box depending on where the driver put the buffers (scripts/ingest_context_probe.py):
        # the three destinations are three placements; `ingest_kernel_s` is their median
        g3 = [_lib.Gallery.empty(hi - lo, d, norm_mode=_lib.NORM_L2, device=job.dev_index, row_offset=lo) for _ in range(3)]
        torch.cuda.synchronize()
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        evs[0].record()
        for i in range(3):
            g3[i].append_device(raw.data_ptr(), hi - lo, stream)
            evs[i + 1].record()
        torch.cuda.synchronize()
        ingest_launch_ms = [evs[i].elapsed_time(evs[i + 1]) for i in range(3)]
        ingest_kernel_s = sorted(ingest_launch_ms)[1] * 1e-3
        for g in g3:
            g.close()
    del raw
    torch.cuda.empty_cache()
    for opt in options:
        name, val = opt.split("=")
        gal.set_option(name, float(val))
    sg = ShardedGallery(gal, group=group, force_protocol=args.force_protocol)
    for opt in options:                           # (the constructor sizes the re-score launch for ITS group: --option wins)
        name, val = opt.split("=")
        gal.set_option(name, float(val))

    # a small pool of query batches (seeded), cycled over the steps
    pool = []
    for i in range(4):
        qb = torch.empty((nq, d), dtype=torch.float32, device=dev)        # this query group's slice of job batch i
        _lib.synth_fill_device(qb.data_ptr(), args.seed + 1 + i, q_lo, nq, d, stream)
        pool.append(qb)
    torch.cuda.synchronize()

    # single GPU: the exact re-score + sort of a batch runs on the handle's second stream beside the scoring launch of the
    # next batch (mi_set_option "async_tail"); every batch's results are complete after gal.join(), inside the timed region
    pipelined = (world == 1 or replicas) and async_tail > 0
    if pipelined:
        gal.set_option("async_tail", async_tail)
    last = {}

    def one_step(qb):
        idx_, sc_ = sg.search(qb, k, join=not pipelined or with_aqe)
        last["q"] = qb
        if with_aqe:
            # ranks[K,Q] view of the [Q,K] result, like `ranks = match_idx.T` (src/test_rOP1m.py:157) -> QGE (N >= 120000)
            idx_, sc_, last["q"] = sg.aqe_search(idx_.t(), 3, 4.0, k, join=not pipelined)
        return idx_, sc_

    use_stream = pipeline and sg._protocol and not with_aqe

    def run_eager(count):
        out_ = None
        for i in range(count):
            out_ = one_step(pool[i % len(pool)])
        return out_

    def run_stream(count):
        out_ = None
        for out_ in sg.search_stream((pool[i % len(pool)] for i in range(count)), k):
            pass
        last["q"] = pool[(count - 1) % len(pool)]
        return out_

    run_steps = run_stream if use_stream else run_eager

    # XCD shares of the tile kernel converged before the first search (mi_gallery_calibrate: a product call, one-off
    # `calibrate` x one batch).  Its launches are also the first sustained load after the (light) ingest, i.e. they take the
    # power-management transient of the chip that follows every idle gap of more than a few ms -- the first ~10 launches
    # after one run 3..25 % slow (scripts/gap_probe.py, profiles/r04a_gap_probe.txt)
    if args.calibrate:
        gal.calibrate(args.calibrate, stream)

    def timed_region(fn, count, warm, profile_it=True):
        """`warm` untimed steps, then EXACTLY `count` steps between barrier + synchronize on both sides.  Nothing but the
        barrier sits between the warm-up and the timed steps: the statistics are reset BEFORE the warm-up (they then cover
        it too: only ratios are read from them) and the launch timing is a host-side switch."""
        gal.status(reset=True)
        out_w = fn(warm) if warm else None
        if pipelined:
            gal.join(stream)
        job.barrier()
        gal.profile(profile_it)
        t0_ = time.perf_counter()
        out_ = fn(count) if count else out_w          # (a replica rank may have no step of a very short job)
        if pipelined:
            gal.join(stream)
        job.barrier()
        el = time.perf_counter() - t0_
        gal.profile(False)
        lms = gal.launch_ms()
        st_ = gal.status(reset=True)
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return out_, el, st_, lms

    graphs = None
    if graph:
        # one hipGraph per pooled query batch: the whole launch sequence of a search (query ingest ... emit), captured on a
        # side stream and replayed; outputs land in the ShardedGallery's (static) result buffers like the eager calls'
        if world > 1 or with_aqe or pipelined or use_stream:
            raise SystemExit("--graph: single-GPU plain search only")
        run_eager(2)
        graphs = []
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for qb in pool:
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr, stream=side):
                    out_g = sg.search(qb, k)
                graphs.append((gr, out_g, qb))
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()

        def run_steps(count):                      # noqa: F811  (graph replays instead of eager launches)
            out_ = None
            for i in range(count):
                gr, out_, qb = graphs[i % len(graphs)]
                gr.replay()
                last["q"] = qb
            return out_

    (idx, sc), elapsed, st, launch_ms = timed_region(run_steps, steps, max(1, warmup) if replicas else warmup,
                                                     profile_it=not graph)
    if world > 1:
        ov = torch.tensor([st["overflow_batches"] + st["spec_retries"]], dtype=torch.int64, device=dev)
        dist.all_reduce(ov, op=dist.ReduceOp.SUM)
        overflow = int(ov.item())
    else:
        overflow = st["overflow_batches"] + st["spec_retries"]

    # result sanity on the last batch (size-independent properties here; the oracle itself checks the full size in tests/:
    # test_gpu_full_size.py scores 8 queries against all rows in float64 on the host)
    worst = None
    dense_check = None
    if check:
        sc_h = sc.cpu().numpy()
        idx_h = idx.cpu().numpy()
        assert (np.diff(sc_h, axis=1) <= 0).all(), "scores not sorted"
        assert all(len(set(r)) == k for r in idx_h), "duplicate indices"
        assert idx_h.min() >= 0 and idx_h.max() < n_total
        # the returned scores are the exact cosines: recompute those of 16 queries in float64 from the stored rows
        # (each rank checks the rows of its own shard); with alpha-QE the last search ran on the expanded queries
        q_last = last["q"].double()
        if not with_aqe:
            q_last = q_last / q_last.norm(dim=1, keepdim=True)
        q_last = q_last.cpu().numpy()
        worst = 0.0
        picks = list(range(0, nq, max(1, nq // 16)))[:16]
        for qi in picks:
            mine = np.flatnonzero((idx_h[qi] >= lo) & (idx_h[qi] < hi))
            if len(mine):
                rows = np.stack([gal.get_rows(int(r) - lo, 1)[0] for r in idx_h[qi, mine]]).astype(np.float64)
                worst = max(worst, float(np.abs(rows @ q_last[qi] - sc_h[qi, mine]).max()))
        assert worst < 3e-7 * max(1.0, float(np.abs(sc_h).max())), \
            "returned scores differ from the float64 re-computation: %g" % worst
        if overflow:
            raise SystemExit("bench invalid: %d batches raised a sticky flag (buffer overflow or failed threshold)" % overflow)
        # COMPLETENESS by an independent path (no sample, no thresholds, no survivor / candidate buffers): every score of
        # this rank's shard for the same 16 queries in float64 + exact top-k of the dense rows (mi_knn_dense64_search).  The
        # job's answer restricted to my rows must be exactly the head of my dense list, and no row of my dense list that
        # beats the job's K-th score may be missing from the answer (src/utils/nnsearch.py:701-703: full argsort = nothing
        # is missing).  The last search ran on `last["q"]` (expanded queries under --with-aqe: used as they are).
        if (hi - lo) >= k and k <= 4096:
            qh = last["q"][picks].cpu().numpy()
            didx, dsc, dsc64, _ = gal.dense64_search(qh, k)
            # (under --with-aqe the expanded queries have norm 1 - 1e-6 / ||sum|| and the host entry point normalises them
            # once more: the order is untouched, the scores move by ~1e-6 relative)
            tol = 3e-6 if with_aqe else 6e-8
            for j, qi in enumerate(picks):
                mine = (idx_h[qi] >= lo) & (idx_h[qi] < hi)
                m = int(mine.sum())
                assert np.array_equal(idx_h[qi][mine], didx[j, :m]), "dense f64 search disagrees (query %d)" % qi
                assert m == 0 or np.abs(sc_h[qi][mine] - dsc[j, :m]).max() <= tol
                # my best row that is NOT in the answer must not beat the job's K-th score
                assert m == k or dsc64[j, m] <= float(sc_h[qi, k - 1]) + tol, \
                    "a row of this shard is missing from the answer (query %d)" % qi
            dense_check = "%d queries x %d rows: dense float64 scores + exact top-%d (mi_knn_dense64_search) equal the answer" \
                          % (len(picks), hi - lo, k)

    res = dict(n_total=n_total, nq_job=nq_job, nq=nq, gq=gq, gs=gs, lo=lo, hi=hi, elapsed=elapsed, steps=steps, st=st,
               job_steps=job_steps,
               ingest_s=ingest_s, ingest_first_s=ingest_first_s, ingest_kernel_s=ingest_kernel_s, ingest_launch_ms=ingest_launch_ms, stateless_call_s=stateless_call_s, alloc_s=alloc_s, worst=worst,
               use_stream=use_stream, launch_ms=[float(v) for v in launch_ms], dense_check=dense_check,
               protocol=sg._protocol, pipelined=pipelined, image_dtype=image_dtype, graph=bool(graph))

    ref_i, ref_s = (idx.clone(), sc.clone()) if (also_stream and sg._protocol) else (None, None)

    def extras():
        """The pipelined collectives (search_stream) and the per-stage timings of the same gallery: secondary figures.  With
        defer_extras they run AFTER the caller has written the headline line, under a deadline of their own -- the first
        execution of the asynchronous all-gathers on real hardware must not be able to cost the synchronous headline."""
        ex = {}
        if also_stream and sg._protocol and not use_stream and not with_aqe:
            (si, ss), el2, st2, _ = timed_region(run_stream, steps, max(3, warmup))
            ex["stream"] = {"ms_per_step": el2 / steps * 1e3, "value": nq_job * steps / el2, "unit": "queries/s", "steps": steps,
                            "equals_synchronous_answer": bool(torch.equal(si, ref_i) and torch.equal(ss, ref_s)),
                            "scoring_share_of_step": st2["gemm_ms"] * 1e-3 / el2}
        if phases and sg._protocol:
            acc = {}
            sg.search_timed(pool[0], k)
            for i in range(phases):
                _, ms = sg.search_timed(pool[i % len(pool)], k)
                for name, v in ms.items():
                    acc[name] = acc.get(name, 0.0) + v / phases
            names = sorted(acc)
            t = torch.tensor([acc[n_] for n_ in names], dtype=torch.float64, device=dev)
            if world > 1:
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ex["phases"] = {n_: round(float(v), 4) for n_, v in zip(names, t.tolist())}
            gal.status(reset=True)
        return ex

    if defer_extras and keep:
        res["later"] = extras
    else:
        res.update(extras())

    if keep:
        res["gal"], res["q_last_pool"] = gal, pool[(steps - 1) % len(pool)]
    else:
        gal.close()
        del sg, gal, pool
        torch.cuda.empty_cache()
    return res


def synchronous_block(job, gal, args, steps):
    """One GPU, same gallery, behind a headline measured in the deferred-tail mode: the SAME number of steps with the tail on
    the caller's stream.  The scoring launch then runs undisturbed, which is the launch `roofline` is quoted on, and the
    record shows both modes of the same box.  Twenty untimed steps first: the result checks of the headline left the device
    idle for seconds, and the first 10-20 launches after an idle gap run slow (power management, scripts/gap_probe.py)."""
    import torch
    from isehr_amd import _lib
    from isehr_amd.sharded import ShardedGallery
    d, k, nq, dev, stream = args.dim, args.topk, args.queries, job.dev, job.stream
    sg = ShardedGallery(gal)
    pool = []
    for i in range(4):
        qb = torch.empty((nq, d), dtype=torch.float32, device=dev)
        _lib.synth_fill_device(qb.data_ptr(), args.seed + 1 + i, 0, nq, d, stream)
        pool.append(qb)
    gal.set_option("async_tail", 3)
    ref = tuple(t.clone() for t in sg.search(pool[(steps - 1) % 4], k))      # join=True: complete
    gal.set_option("async_tail", 0)
    gal.status(reset=True)
    out_ = None
    for i in range(20):
        out_ = sg.search(pool[i % 4], k)
    torch.cuda.synchronize()
    gal.profile(True)
    t0 = time.perf_counter()
    for i in range(steps):
        out_ = sg.search(pool[i % 4], k)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    gal.profile(False)
    lms = [float(v) for v in gal.launch_ms()]
    st = gal.status(reset=True)
    same = bool(torch.equal(out_[0], ref[0]) and torch.equal(out_[1], ref[1]))
    return dict(st=st, nq=nq, elapsed=elapsed, steps=steps, launch_ms=lms, equals_pipelined_answer=same)



def small_batch_block(job, gal, args, nq, steps):
    """BASELINE.md section 4's small shapes on the gallery already resident: Q = 70 (the reference's own test sets,
    src/test_rOP1m.py) and Q = 1 (one uploaded image, src/online.py:132-149), HBM-bound: the 16-bit image is streamed once per
    batch (N * D * 2 bytes) by stream_select_kernel.  Synchronous calls, every step complete when it returns."""
    import numpy as np
    import torch
    from isehr_amd import _lib
    from isehr_amd.sharded import ShardedGallery
    d, k, dev, stream = args.dim, args.topk, job.dev, job.stream
    sg = ShardedGallery(gal)
    pool = []
    for i in range(4):
        qb = torch.empty((nq, d), dtype=torch.float32, device=dev)
        _lib.synth_fill_device(qb.data_ptr(), args.seed + 11 + i, 0, nq, d, stream)
        pool.append(qb)
    gal.set_option("async_tail", 0)
    for i in range(10):
        sg.search(pool[i % 4], k)
    torch.cuda.synchronize()
    gal.status(reset=True)
    gal.profile(True)
    t0 = time.perf_counter()
    out_ = None
    for i in range(steps):
        out_ = sg.search(pool[i % 4], k)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    gal.profile(False)
    lms = [float(v) for v in gal.launch_ms()]
    st = gal.status(reset=True)
    m = min(nq, 16)
    idx_h, sc_h = out_[0][:m].cpu().numpy(), out_[1][:m].cpu().numpy()
    didx, dsc, _, _ = gal.dense64_search(pool[(steps - 1) % 4][:m].cpu().numpy(), k)
    assert np.array_equal(idx_h, didx) and float(np.abs(sc_h - dsc).max()) <= 6e-8, "q%d block: dense f64 search disagrees" % nq
    roof = roofline_of(dict(st=st, nq=nq, elapsed=elapsed, launch_ms=lms), args, 1, False)
    roof["frac_of_achievable"] = roof["achieved"] / HBM_ACHIEVABLE_GBS if roof["achieved"] else None
    roof["algorithmic_bytes_per_launch"] = st["gemm_bytes"] / max(1, st["gemm_launches"])
    whole = (gal.n * d * 2.0 + nq * d * 2.0) * steps / elapsed / 1e9
    return {"queries_per_step": nq, "steps": steps, "value": nq * steps / elapsed, "unit": "queries/s",
            "ms_per_step": elapsed / steps * 1e3, "roofline": roof, "whole_step_GBps": whole,
            "whole_step_frac_of_8TBps": whole / HBM_PEAK_GBS,
            "flagged_batches": st["overflow_batches"] + st["spec_retries"], "inkernel_repairs": st["inkernel_repairs"],
            "score_check": "%d queries x %d rows: dense float64 scores + exact top-%d equal the answer" % (m, gal.n, k)}


QSWEEP = (128, 129, 192, 256, 257, 384, 512, 768, 1024)


def qsweep_block(job, gal, args, steps=20):
    """The batch-size staircase on the resident gallery: per point the whole step, the scoring launch between
    HIP events, the roof of the shape = max(N * D * 2 B / 8 TB/s, 2 * Q * N * D / 2.5 PF) and both fractions of it.  The
    reference has no batch-size cliff (src/utils/nnsearch.py:699 loops over the queries)."""
    import torch
    from isehr_amd import _lib
    from isehr_amd.sharded import ShardedGallery
    d, k, dev, stream = args.dim, args.topk, job.dev, job.stream
    sg = ShardedGallery(gal)
    gal.set_option("async_tail", 0)
    qall = torch.empty((1024, d), dtype=torch.float32, device=dev)
    _lib.synth_fill_device(qall.data_ptr(), args.seed + 23, 0, 1024, d, stream)
    points = []
    for nq in QSWEEP:
        qb = qall[:nq]
        for _ in range(5):
            sg.search(qb, k)
        torch.cuda.synchronize()
        gal.status(reset=True)
        gal.profile(True)
        t0 = time.perf_counter()
        for _ in range(steps):
            sg.search(qb, k)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        gal.profile(False)
        lms = [float(v) for v in gal.launch_ms()]
        st = gal.status(reset=True)
        ms = el / steps * 1e3
        launch = sum(lms) / max(1, len(lms))
        roof_ms = max(gal.n * d * 2.0 / (HBM_PEAK_GBS * 1e9), 2.0 * nq * gal.n * d / (MFMA_BF16_PEAK_TFLOPS * 1e12)) * 1e3
        points.append({"queries": nq, "ms_per_step": round(ms, 4), "scoring_launch_ms": round(launch, 4),
                       "launches_per_step": len(lms) / steps, "roof_ms": round(roof_ms, 4),
                       "frac_whole_step": round(roof_ms / ms, 4), "frac_at_launch": round(roof_ms / launch, 4) if launch else None,
                       "value": nq / (ms * 1e-3),
                       "flagged_batches": st["overflow_batches"] + st["spec_retries"]})
    by = {p["queries"]: p for p in points}
    return {"gallery_rows": gal.n, "steps": steps, "unit": "queries/s", "points": points,
            "roof": "max(N*D*2 B / 8 TB/s, 2*Q*N*D / 2.5 PFLOP/s)",
            "ms_257_over_256": round(by[257]["ms_per_step"] / by[256]["ms_per_step"], 4),
            "ms_129_over_128": round(by[129]["ms_per_step"] / by[128]["ms_per_step"], 4),
            "min_frac_whole_step": min(p["frac_whole_step"] for p in points),
            "min_frac_at_launch": min(p["frac_at_launch"] for p in points)}


def _hard_rows(kind, n, d, dev, seed):
    """Structured galleries of VERDICT r05 #1, generated on the device with torch's generator (plumbing: the data, not the
    product).  Returns (raw rows [n, d] f32, queries(nq) -> [nq, d] f32, description)."""
    import torch
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    raw = torch.empty((n, d), dtype=torch.float32, device=dev)
    blk = 65536

    def randn(*shape):
        return torch.randn(shape, generator=g, device=dev, dtype=torch.float32)

    def unit(x):
        return x / x.norm(dim=1, keepdim=True)
    if kind == "nonneg":
        for r0 in range(0, n, blk):
            m = min(blk, n - r0)
            raw[r0:r0 + m] = randn(m, d).abs_()

        def queries(nq):
            return randn(nq, d).abs_()
        desc = "|N(0,1)| rows and queries (pre-whitening GeM descriptors are non-negative): mean pairwise cosine 2/pi = 0.64"
    elif kind == "clustered":
        per = 500
        ncl = max(16, (n - 5994) // per)                               # 2000 clusters at the benchmarked 1 005 994 rows
        centres = unit(randn(ncl, d))
        for r0 in range(0, n, blk):
            m = min(blk, n - r0)
            ids = torch.arange(r0, r0 + m, device=dev)
            cl = torch.clamp(ids // per, max=ncl - 1)
            cos = 0.90 + 0.09 * torch.rand((m, 1), generator=g, device=dev)
            rows = cos * centres[cl] + torch.sqrt(1 - cos * cos) * unit(randn(m, d))
            back = (ids >= ncl * per).unsqueeze(1)
            raw[r0:r0 + m] = torch.where(back, randn(m, d), rows)

        def queries(nq):
            # five queries per landmark like rOxford / rParis (70 queries of ~13 landmarks), cosine 0.95 to the centre: the
            # K-th (100th) best score lies INSIDE a cluster of 500 rows at cosine 0.855 .. 0.94 to the query
            cl = (torch.arange(nq, device=dev) // 5 * 37) % ncl
            return 0.95 * centres[cl] + (1 - 0.95 ** 2) ** 0.5 * unit(randn(nq, d))
        desc = "%d clusters x 500 rows at cosine 0.90-0.99 to their centre (+ %d Gaussian rows); queries at cosine 0.95 to a " \
               "centre, five per cluster" % (ncl, n - ncl * per)
    elif kind == "near_duplicates":
        nbase, copies = 125000, 8
        base = unit(randn(nbase, d))
        exact = unit(randn(16, d))
        n_dup = nbase * copies
        for r0 in range(0, n, blk):
            m = min(blk, n - r0)
            ids = torch.arange(r0, r0 + m, device=dev)
            rows = base[torch.clamp(ids // copies, max=nbase - 1)] + 1e-4 * unit(randn(m, d))
            ex = exact[torch.clamp((ids - n_dup) // 64, min=0, max=15)]
            rows = torch.where(((ids >= n_dup) & (ids < n_dup + 1024)).unsqueeze(1), ex, rows)
            raw[r0:r0 + m] = torch.where((ids >= n_dup + 1024).unsqueeze(1), randn(m, d), rows)

        def queries(nq):
            # query i < 16: near one of the 16 rows stored 64 times (its best 64 scores are exact ties); the others near a
            # base row (their best 8 scores differ by ~2e-6, far inside the certificate's margin)
            q = unit(base[(torch.arange(nq, device=dev) * 97) % nbase] + 0.5 * unit(randn(nq, d)))
            m = min(nq, 16)
            q[:m] = unit(exact[:m] + 0.5 * unit(randn(m, d)))
            return q
        desc = "125 000 base rows x 8 copies with 1e-4 relative noise + 64 exact copies of each of 16 rows (+ %d Gaussian rows)" \
               % (n - n_dup - 1024)
    else:
        raise ValueError(kind)
    return raw, queries, desc


def hard_data_block(job, args, gaussian_qps, steps=10):
    """The headline's shape on data that is NOT i.i.d. Gaussian: the filter's work per query -- candidates,
    survivors, repairs, fallbacks -- depends on the data, the reference's does not (src/utils/nnsearch.py:693-703: every row,
    full argsort).  Per gallery: 1024 / 70 / 1 queries, K = 100, `steps` synchronous steps each through the device entry
    point; a case that raised a sticky flag is timed again WITH the fallbacks of the host entry point (verify=True) and that
    figure is the value.  Every case ends with the dense float64 completeness check of 16 queries."""
    import numpy as np
    import torch
    from isehr_amd import _lib
    from isehr_amd.sharded import ShardedGallery
    n, d, k, dev, stream = WORKLOADS["roxford5k+1m"][0], args.dim, args.topk, job.dev, job.stream
    out = {"gallery_rows": n, "topk": k, "steps": steps, "unit": "queries/s", "cases": {}}
    _lib.set_global_option("image_dtype", 1 if args.image_dtype == "f16" else 0)
    for ci, kind in enumerate(("nonneg", "clustered", "near_duplicates")):
        raw, queries, desc = _hard_rows(kind, n, d, dev, args.seed + 500 + ci)
        torch.cuda.synchronize()
        gal = _lib.Gallery.from_device_ptr(raw.data_ptr(), n, d, norm_mode=_lib.NORM_L2, device=job.dev_index)
        del raw
        torch.cuda.empty_cache()
        rec = {"data": desc}
        try:
            sg = ShardedGallery(gal)
            gal.calibrate(8, stream)
            for nq in (1024, 70, 1):
                pool = [queries(nq).contiguous() for _ in range(2)]

                def run(count, verify):
                    o = None
                    for i in range(count):
                        o = sg.search(pool[i % 2], k, verify=verify)
                    torch.cuda.synchronize()
                    return o
                run(3, False)
                gal.flags()
                gal.status(reset=True)
                gal.profile(True)
                t0 = time.perf_counter()
                o = run(steps, False)
                el = time.perf_counter() - t0
                gal.profile(False)
                lms = [float(v) for v in gal.launch_ms()]
                flagged_now = bool(gal.flags())
                st = gal.status(reset=True)
                flagged = st["overflow_batches"] + st["spec_retries"]
                case = {"queries_per_step": nq, "ms_per_step": el / steps * 1e3, "value": nq * steps / el,
                        "scoring_launch_ms": sum(lms) / max(1, len(lms)),
                        "candidates_per_query": st["candidates"] / max(1, st["queries"]),
                        "survivors_per_query": st["survivors"] / max(1, st["queries"]),
                        "flagged": flagged_now or flagged > 0, "inkernel_repairs": st["inkernel_repairs"]}
                if case["flagged"]:
                    # answered again by the rigorous schedule / the f32 scorer where a flag was raised: the honest figure
                    run(1, True)
                    gal.status(reset=True)
                    t0 = time.perf_counter()
                    o = run(steps, True)
                    el = time.perf_counter() - t0
                    st = gal.status(reset=True)
                    case.update({"ms_per_step_unverified": case["ms_per_step"], "ms_per_step": el / steps * 1e3,
                                 "value": nq * steps / el, "mode": "verify=True: flags of every batch read, fallbacks run",
                                 "overflow_batches": st["overflow_batches"], "spec_retries": st["spec_retries"]})
                m = min(nq, 16)
                idx_h, sc_h = o[0][:m].cpu().numpy(), o[1][:m].cpu().numpy()
                didx, dsc, dsc64, dsec = gal.dense64_search(pool[(steps - 1) % 2][:m].cpu().numpy(), k)
                if nq == 1024:
                    # what the LAST resort of the fallback chain costs on this gallery (every score in float64, no thresholds):
                    # the price of a batch that leaves the filtered paths altogether
                    case["dense_f64_fallback_ms_per_query"] = dsec / m * 1e3
                # (exact ties may be cut anywhere among EQUAL float64 scores by either path: compare scores, and ids where
                # the float64 scores differ from their neighbours)
                same_ids = bool(np.array_equal(idx_h, didx))
                case["score_check"] = {"queries": m, "ids_equal_dense_f64": same_ids,
                                       "max_abs_score_diff": float(np.abs(sc_h - dsc).max())}
                assert case["score_check"]["max_abs_score_diff"] <= 6e-8, "%s q%d: scores differ from the dense f64 path" % (kind, nq)
                assert same_ids, "%s q%d: ids differ from the dense f64 path" % (kind, nq)
                if nq == 1024 and gaussian_qps:
                    case["vs_gaussian_headline"] = case["value"] / gaussian_qps
                rec["q%d" % nq] = case
        finally:
            gal.close()
            torch.cuda.empty_cache()
        out["cases"][kind] = rec
    out["min_vs_gaussian_headline"] = min(c["q1024"].get("vs_gaussian_headline", 1.0) for c in out["cases"].values())
    return out


F64_MATRIX_PEAK_TFLOPS = 78.6    # AMD's MI355X data sheet (FP64 matrix = FP64 vector); the guide lists no f64 matrix figure


def whiten_block(job, args):
    """Learned whitening (north_star; src/utils/whiten.py:4-12, applied to the whole database at src/main_train.py:711-712) at
    BASELINE size: 1 005 994 x 2048 float32 descriptors on the device, float64 mean and a 2048 x 2048 float64 P.  (i)
    mi_whiten_apply_device -> float64 [N, dims] for dims = 2048 and 128 (f64 MFMA GEMM + one-pass normalisation), (ii)
    mi_gallery_append_whitened_device: whitened rows straight into a searchable gallery, the float64 matrix never exists."""
    import numpy as np
    import torch
    from isehr_amd import _lib
    n, d, dev, stream = WORKLOADS["roxford5k+1m"][0], args.dim, job.dev, job.stream
    raw = torch.empty((n, d), dtype=torch.float32, device=dev)
    _lib.synth_fill_device(raw.data_ptr(), args.seed + 60, 0, n, d, stream)
    raw /= raw.norm(dim=1, keepdim=True)
    g = torch.Generator(device=dev)
    g.manual_seed(args.seed + 61)
    P = torch.randn((d, d), dtype=torch.float64, device=dev, generator=g) / d ** 0.5
    m = raw[:65536].double().mean(dim=0).contiguous()
    out = torch.empty((n, d), dtype=torch.float64, device=dev)
    picks = list(range(0, n, n // 16))[:16]
    xh = raw[picks].double().cpu().numpy()
    rec = {"rows": n, "d": d, "peak_TFLOPs": F64_MATRIX_PEAK_TFLOPS,
           "peak_source": "AMD MI355X data sheet, FP64 matrix; /opt/skills/guides/MI355X_MICROARCH.md lists no f64 matrix rate",
           "kernel": "whiten_mfma_kernel: v_mfma_f64_16x16x4_f64, 128 x 128 outputs per workgroup, (x - m) applied on load"}
    try:
        for dims in (d, 128):
            best = {}
            for name, eps in (("gemm_plus_normalise", 1e-6), ("gemm_only", -1.0)):
                for _ in range(3):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    _lib.whiten_apply_device(raw.data_ptr(), n, d, m.data_ptr(), P.data_ptr(), dims, out.data_ptr(), eps=eps,
                                             stream=stream)
                    torch.cuda.synchronize()
                    dt = time.perf_counter() - t0
                    best[name] = min(best.get(name, 1e9), dt)
            _lib.whiten_apply_device(raw.data_ptr(), n, d, m.data_ptr(), P.data_ptr(), dims, out.data_ptr(), eps=1e-6, stream=stream)
            torch.cuda.synchronize()
            got = out.view(-1)[:n * dims].view(n, dims)[picks].cpu().numpy()
            ref = (xh - m.cpu().numpy()) @ P[:dims].cpu().numpy().T
            ref /= (np.linalg.norm(ref, axis=1, keepdims=True) + 1e-6)
            err = float(np.abs(got - ref).max())
            assert err < 1e-12, "whitening differs from the float64 recomputation: %g" % err
            flop = 2.0 * n * dims * d
            rec["dims_%d" % dims] = {
                "seconds": round(best["gemm_plus_normalise"], 5), "gemm_seconds": round(best["gemm_only"], 5),
                "TFLOPs_gemm": flop / best["gemm_only"] / 1e12, "frac_of_f64_matrix_peak": flop / best["gemm_only"] / 1e12 / F64_MATRIX_PEAK_TFLOPS,
                "TFLOPs_incl_normalise": flop / best["gemm_plus_normalise"] / 1e12,
                "bytes_algorithmic": n * d * 4 + n * dims * 8 * (1 + 2), "rows_per_s": n / best["gemm_plus_normalise"],
                "max_abs_diff_vs_float64_numpy_16_rows": err}
        # the layout whitenapply is CALLED with: X is [D, N], one image per column (src/utils/whiten.py:4-12; `vecs` of
        # src/main_train.py:711-712) -- the same rows, transposed in device memory, read with strides (row stride 1, column
        # stride N): the kernel's other loader (64 consecutive images of one k per wave instruction)
        xt = torch.empty((d, n), dtype=torch.float32, device=dev)
        for r0 in range(0, n, 131072):
            xt[:, r0:r0 + 131072] = raw[r0:r0 + 131072].t()
        best_t = 1e9
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _lib.whiten_apply_device(xt.data_ptr(), n, d, m.data_ptr(), P.data_ptr(), d, out.data_ptr(), eps=-1.0,
                                     row_stride=1, col_stride=n, stream=stream)
            torch.cuda.synchronize()
            best_t = min(best_t, time.perf_counter() - t0)
        _lib.whiten_apply_device(xt.data_ptr(), n, d, m.data_ptr(), P.data_ptr(), d, out.data_ptr(), eps=1e-6,
                                 row_stride=1, col_stride=n, stream=stream)
        torch.cuda.synchronize()
        got = out[picks].cpu().numpy()
        ref = (xh - m.cpu().numpy()) @ P.cpu().numpy().T
        ref /= (np.linalg.norm(ref, axis=1, keepdims=True) + 1e-6)
        err = float(np.abs(got - ref).max())
        assert err < 1e-12, "whitening of the [D, N] layout differs from the float64 recomputation: %g" % err
        rec["dims_2048_from_DxN"] = {"gemm_seconds": round(best_t, 5), "TFLOPs_gemm": 2.0 * n * d * d / best_t / 1e12,
                                     "frac_of_f64_matrix_peak": 2.0 * n * d * d / best_t / 1e12 / F64_MATRIX_PEAK_TFLOPS,
                                     "vs_row_major": best_t / rec["dims_%d" % d]["gemm_seconds"],
                                     "max_abs_diff_vs_float64_numpy_16_rows": err,
                                     "layout": "X [D, N] float32 (the reference's), row stride 1, column stride N"}
        del out, xt
        torch.cuda.empty_cache()
        # (ii) whitened rows straight into a gallery (MI_NORM_L2_EPS = whitenapply's tail): one call, 512 MiB of scratch
        gal = _lib.Gallery.empty(n, d, norm_mode=_lib.NORM_L2_EPS, device=job.dev_index)
        try:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            gal.append_whitened_device(raw.data_ptr(), n, d, m.data_ptr(), P.data_ptr(), stream=stream)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            ref = (xh - m.cpu().numpy()) @ P.cpu().numpy().T
            ref /= (np.linalg.norm(ref, axis=1, keepdims=True) + 1e-6)
            got = np.stack([gal.get_rows(int(r), 1)[0] for r in picks])
            err = float(np.abs(got - ref).max())
            assert err < 2e-7, "whitened gallery rows differ from the float64 recomputation: %g" % err
            rec["into_gallery"] = {"seconds": round(dt, 5), "rows_per_s": n / dt, "TFLOPs": 2.0 * n * d * d / dt / 1e12,
                                   "max_abs_diff_of_stored_f32_rows": err,
                                   "note": "whiten -> normalise -> f32 rows + 16-bit image + rounding norms, chunks of 32 768 rows; "
                                           "the [N, 2048] float64 matrix (16.5 GB) never exists"}
        finally:
            gal.close()
    finally:
        del raw
        torch.cuda.empty_cache()
    return rec


def online_block(job, gal, args):
    """src/online.py:121-152 for ONE uploaded image, kept on the device (entry/online.py Searcher.query_device): the descriptor as
    the extractor tail leaves it (a device row) -> search, K = 100, on the L2-normalised gallery -> qge1 (alpha-QE k = 3, w = 4:
    src/utils/Reranking.py:287-306) expansion from the rows AS STORED + re-search -> one device-to-host copy of K indices.  The
    second gallery holds the same rows un-renormalised (the reference's `vecs`: unit descriptors), like the `database_raw`
    gallery of the entry point."""
    import numpy as np
    import torch
    from isehr_amd import _lib
    from isehr_amd.sharded import ShardedGallery
    n, d, k, dev, stream = gal.n, args.dim, args.topk, job.dev, job.stream
    _lib.set_global_option("image_dtype", 1 if args.image_dtype == "f16" else 0)
    g_raw = _lib.Gallery.empty(n, d, norm_mode=_lib.NORM_NONE, device=job.dev_index)
    try:
        blk = 131072
        for r0 in range(0, n, blk):                               # unit rows, block by block (no second 8 GB tensor)
            m = min(blk, n - r0)
            raw = torch.empty((m, d), dtype=torch.float32, device=dev)
            _lib.synth_fill_device(raw.data_ptr(), args.seed, r0, m, d, stream)
            raw /= raw.norm(dim=1, keepdim=True)
            g_raw.append_device(raw.data_ptr(), m, stream)
        torch.cuda.synchronize()
        del raw
        if args.image_dtype == "f16" and g_raw.norm_bounds()[0] <= 4.0:
            g_raw.set_image_dtype(1)                              # unit rows sit inside fp16's range (appendable raw galleries start as bf16)
        sg1, sg2 = ShardedGallery(gal), ShardedGallery(g_raw)
        gal.set_option("async_tail", 0)
        qs = torch.empty((8, d), dtype=torch.float32, device=dev)
        _lib.synth_fill_device(qs.data_ptr(), args.seed + 77, 0, 8, d, stream)

        def chain(q):
            idx, _ = sg1.search(q, k)
            idx2, _, _ = sg2.aqe_search(idx.t(), 3, 4.0, k)
            return idx2.cpu().numpy()                             # the one D2H copy (synchronises)
        for i in range(5):
            chain(qs[i % 8:i % 8 + 1])
        steps = 100
        t0 = time.perf_counter()
        for i in range(steps):
            out = chain(qs[i % 8:i % 8 + 1])
        el = time.perf_counter() - t0
        # uploaded images do not arrive back to back: the same chain with 50 ms of idle GPU before each query (clocks down,
        # scripts/gap_probe.py) -- the latency a visitor of the web service sees
        idle = []
        for i in range(12):
            time.sleep(0.05)
            t1 = time.perf_counter()
            chain(qs[i % 8:i % 8 + 1])
            idle.append(time.perf_counter() - t1)
        idle_ms = sorted(idle)[len(idle) // 2] * 1e3
        # check: the chain's answer for the last query = search + host-API alpha-QE of the same gallery on the same top-3
        q_last = qs[(steps - 1) % 8:(steps - 1) % 8 + 1]
        idx_h, _, _ = gal.search(q_last.cpu().numpy(), k)
        ref_idx, _, _, _ = g_raw.aqe_search(np.ascontiguousarray(idx_h.T), 3, 4.0, k)
        assert np.array_equal(out, ref_idx), "online device chain differs from the host entry points"
        # concurrent clients (src/online.py:163: Flask's threaded server, one thread per request): 64 threads x 20 queries
        # through entry/online.py Searcher.query_device, whose worker coalesces the waiting descriptors into one chain
        import threading
        from isehr_amd.entry.online import Searcher
        nthr = 64
        qc = torch.empty((nthr, d), dtype=torch.float32, device=dev)
        _lib.synth_fill_device(qc.data_ptr(), args.seed + 78, 0, nthr, d, stream)
        torch.cuda.synchronize()
        rows_c = [qc[j] for j in range(nthr)]                         # the requests' descriptors exist before the requests do
        conc = {}
        # queries per thread: 20 / 84 / 148 (the same last query of every thread, 19 mod 64: the answers are compared), so
        # that each mode runs for a good fraction of a second
        for mode, coalesce, native, per in (("sequential_calls_same_threads", False, False, 20),
                                            ("coalesced_python_worker", True, False, 84), ("coalesced", True, True, 148)):
            kw = {"max_wait_s": float(os.environ["ISEHR_ONLINE_WAIT_US"]) * 1e-6} if "ISEHR_ONLINE_WAIT_US" in os.environ else {}
            srv = Searcher.from_galleries(gal, g_raw, k, device=job.dev_index, coalesce=coalesce, native=native, **kw)
            try:
                got = [None] * nthr
                errs = []

                def client(t):
                    try:
                        torch.cuda.set_device(job.dev_index)
                        for i in range(per):
                            got[t] = srv.query_device(rows_c[(t + i) % nthr], return_indices=True)
                    except Exception as e:                                # noqa: BLE001
                        errs.append(e)
                srv.query_device(qc[0], return_indices=True)                                        # warm
                ths = [threading.Thread(target=client, args=(t,)) for t in range(nthr)]
                t0 = time.perf_counter()
                for th in ths:
                    th.start()
                for th in ths:
                    th.join()
                elc = time.perf_counter() - t0
                if errs:
                    raise errs[0]
                chains, answered = srv.chain_stats if coalesce else (nthr * per, nthr * per)
                conc[mode] = {"value": nthr * per / elc, "unit": "queries/s", "seconds": elc, "queries_per_thread": per,
                              "worker": ("library (mi_online_*)" if native else "python thread") if coalesce else None,
                              "chains_launched": chains, "mean_requests_per_chain": answered / max(1, chains)}
                conc[mode + "_answers"] = np.concatenate(got)
                if native:
                    # the same front driven by 64 request threads of the library itself (mi_debug_online_clients): what is left
                    # when the clients' interpreter time is taken out -- the figure a compiled host, or a Python host whose
                    # request threads do nothing else under the interpreter lock, sees
                    sec, last = srv._native().native_clients(qc.data_ptr(), nthr, nthr, 148 + 128)
                    c0, a0 = chains, answered
                    chains, answered = srv.chain_stats
                    conc["coalesced_native_clients"] = {
                        "value": nthr * (148 + 128) / sec, "unit": "queries/s", "seconds": sec, "queries_per_thread": 148 + 128,
                        "worker": "library (mi_online_*)", "clients": "threads of the library (mi_debug_online_clients)",
                        "chains_launched": chains - c0, "mean_requests_per_chain": (answered - a0) / max(1, chains - c0),
                        "equals_sequential_answers": bool(np.array_equal(last, conc["sequential_calls_same_threads_answers"]))}
                    assert conc["coalesced_native_clients"]["equals_sequential_answers"]
            finally:
                srv.close()
        want_c = conc.pop("sequential_calls_same_threads_answers")
        same = bool(np.array_equal(want_c, conc.pop("coalesced_answers")) and
                    np.array_equal(want_c, conc.pop("coalesced_python_worker_answers")))
        assert same, "coalesced answers differ from the sequential calls'"
        conc.update({"client_threads": nthr, "equals_sequential_answers": same})
        return {"gallery_rows": n, "topk": k, "steps": steps, "online_query_ms": el / steps * 1e3, "value": steps / el,
                "online_query_ms_after_50ms_idle": idle_ms, "online_concurrent": conc,
                "online_concurrent_qps": conc["coalesced"]["value"],
                "online_concurrent_native_clients_qps": conc["coalesced_native_clients"]["value"],
                "unit": "queries/s", "stages": "descriptor on the device -> mi_knn_search_device (K) -> qge1 expansion (k = 3, w = 4) "
                                               "-> re-search -> one D2H of K indices",
                "score_check": "equals mi_knn_search + mi_aqe_search (host entry points) on the same galleries"}
    finally:
        g_raw.close()
        torch.cuda.empty_cache()


def aqe_rparis_block(job, args):
    """BASELINE configs[4]: rParis6k + 1M distractors (N = 1 007 323 x 2048), alpha-QE re-ranking as QGE's large-database branch
    runs it (k = 3, w = 4.0, one iteration: src/utils/Reranking.py:273-283 -> feature_enhancement :195-208): one step = search
    (K = 100) -> expansion of every query from its top-3 rows (float64 weighted sum, eps-normalised) -> re-search of the
    expanded queries.  1024-query batches (the tile kernel) and 70 (the reference's query set)."""
    import numpy as np
    import torch
    from isehr_amd import _lib
    from isehr_amd.sharded import ShardedGallery
    n, d, k, dev, stream = WORKLOADS["rparis6k+1m"][0], args.dim, args.topk, job.dev, job.stream
    raw = torch.empty((n, d), dtype=torch.float32, device=dev)
    _lib.synth_fill_device(raw.data_ptr(), args.seed + 40, 0, n, d, stream)
    torch.cuda.synchronize()
    _lib.set_global_option("image_dtype", 1 if args.image_dtype == "f16" else 0)
    gal = _lib.Gallery.from_device_ptr(raw.data_ptr(), n, d, norm_mode=_lib.NORM_L2, device=job.dev_index)
    del raw
    torch.cuda.empty_cache()
    out = {"gallery_rows": n, "k_qe": 3, "w": 4.0, "topk": k, "image": args.image_dtype}
    try:
        sg = ShardedGallery(gal)
        gal.calibrate(8, stream)
        wts = np.array([(3 - j) / 3.0 for j in range(3)], dtype=np.float64) ** 4.0
        for nq in (1024, 70):
            pool = []
            for i in range(2):
                qb = torch.empty((nq, d), dtype=torch.float32, device=dev)
                _lib.synth_fill_device(qb.data_ptr(), args.seed + 41 + i, 0, nq, d, stream)
                pool.append(qb)

            def step(qb):
                idx1, _ = sg.search(qb, k)
                return idx1, sg.aqe_search(idx1.t(), 3, 4.0, k)
            for i in range(3):
                step(pool[i % 2])
            torch.cuda.synchronize()
            gal.status(reset=True)
            steps = 10
            t0 = time.perf_counter()
            for i in range(steps):
                step(pool[i % 2])
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            st = gal.status(reset=True)
            # checks (one more, untimed step: the first search's result buffer is reused by the re-search, so its top-3 ids are
            # copied before the expansion): (i) the expansion of 4 queries recomputed in float64 from the stored rows, (ii) the
            # re-search of 8 expanded queries against every row in float64 (dense path: no thresholds)
            idx1, _ = sg.search(pool[(steps - 1) % 2], k)
            idx1_h = idx1[:4, :3].cpu().numpy()
            idx2, sc2, qx = sg.aqe_search(idx1.t(), 3, 4.0, k)
            torch.cuda.synchronize()
            qx_h = qx[:8].cpu().numpy()
            worst = 0.0
            for qi in range(4):
                rows = np.stack([gal.get_rows(int(r), 1)[0] for r in idx1_h[qi, :3]]).astype(np.float64)
                e = (rows * wts[:, None]).sum(axis=0)
                e /= (np.linalg.norm(e) + 1e-6)
                worst = max(worst, float(np.abs(e - qx_h[qi]).max()))
            assert worst < 1e-6, "alpha-QE expansion differs from the float64 recomputation: %g" % worst
            didx, dsc, _, _ = gal.dense64_search(qx_h, k)
            assert np.array_equal(idx2[:8].cpu().numpy(), didx), "alpha-QE re-search: dense f64 search disagrees"
            assert float(np.abs(sc2[:8].cpu().numpy() - dsc).max()) <= 3e-6
            out["q%d" % nq] = {"queries_per_step": nq, "steps": steps, "value": nq * steps / el, "unit": "queries/s",
                               "ms_per_step": el / steps * 1e3,
                               "flagged_batches": st["overflow_batches"] + st["spec_retries"],
                               "score_check": "expansion of 4 queries vs float64: max |d| %.1e; re-search of 8 expanded queries "
                                              "= dense float64 top-%d" % (worst, k)}
    finally:
        gal.close()
        torch.cuda.empty_cache()
    return out


def qge_small_block(args, device):
    """QGE's small-database branch (N < 120 000, src/utils/Reranking.py:212-264) at rOxford5k size (4993 x 2048, 70 queries,
    planted dataset of map_block): alpha-QE k = 10, w = 4 -> offline truncated diffusion (kNN graph top-2000, mutual-200
    affinity, Laplacian, 4993 CG solves; src/utils/diffusion.py:52-116) -> online combination with the expanded queries
    (k_query = 3, truncation 2000) -> mAP.  The reference caches the offline stage in offline.jbl; here it is timed."""
    import numpy as np
    from isehr_amd import evaluate
    from isehr_amd.diffusion import Diffusion
    from isehr_amd.nnsearch import matching_HIP
    from isehr_amd.reranking import feature_enhancement_hip
    n = 4993
    vecs, qv, gnd = _planted(args, n)
    idx, _ = matching_HIP(args.topk, vecs.T, qv.T, device=device)
    ranks = idx.T
    t0 = time.time()
    qx, ranks_aqe = feature_enhancement_hip(10, ranks, vecs, 4.0, 1000, None, False, device)
    aqe_s = time.time() - t0
    trunc, kd = 2000, 200
    t0 = time.time()
    dif = Diffusion(np.ascontiguousarray(vecs.T), None, device)
    try:
        build_s = time.time() - t0
        t0 = time.time()
        import contextlib
        import io
        with contextlib.redirect_stdout(io.StringIO()):
            dif.get_offline_results(trunc, kd)
        offline_s = time.time() - t0
        dif.search_online(qx.T, 3, trunc)                         # warm
        t0 = time.time()
        reps = 5
        for _ in range(reps):
            ranks_dfs, scores = dif.search_online(qx.T, 3, trunc)
        online_s = (time.time() - t0) / reps
    finally:
        dif.close()
    assert ranks_dfs.shape == (trunc, 70) and (np.diff(scores, axis=1) <= 0).all(), "diffusion scores not sorted"
    assert all(len(set(ranks_dfs[:, i])) == trunc for i in range(0, 70, 7)), "duplicate ids in a diffusion ranking"
    m0 = evaluate.compute_map_revisited(ranks, gnd)
    m1 = evaluate.compute_map_revisited(ranks_aqe[:args.topk], gnd)
    m2 = evaluate.compute_map_revisited(ranks_dfs, gnd)
    return {"gallery_rows": n, "queries": 70, "truncation": trunc, "k_gallery": kd, "k_query": 3,
            "alpha_qe_s": round(aqe_s, 4), "gallery_prepare_s": round(build_s, 4), "offline_diffusion_s": round(offline_s, 3),
            "online_s_per_70_queries": round(online_s, 5), "online_value": 70 / online_s, "unit": "queries/s",
            "map": {"search": {"E": m0[0], "M": m0[1], "H": m0[2]}, "alpha_qe": {"E": m1[0], "M": m1[1], "H": m1[2]},
                    "diffusion": {"E": m2[0], "M": m2[1], "H": m2[2]}},
            "_ranks_aqe": ranks_aqe[:args.topk],
            "score_check": "rankings sorted and duplicate-free; mAP of the alpha-QE ranks equals the oracle's "
                           "feature_enhancement (cpu_baseline leg)"}


def cpu_baseline_qge(args, rec):
    """cpu_baseline leg: the oracle's feature_enhancement (k = 10, w = 4) on the planted rOxford5k-sized dataset; the mAP of
    the HIP alpha-QE ranks must equal the mAP of the oracle's."""
    import oracle
    vecs, qv, gnd = _planted(args, 4993)
    ref = oracle.matching_l2(args.topk, vecs.T, qv.T)
    t0 = time.time()
    _, ranks_o = oracle.feature_enhancement(10, ref.T, vecs, 4.0)
    rec["alpha_qe_oracle_s"] = round(time.time() - t0, 3)
    mo = oracle.compute_map_revisited(ranks_o[:args.topk], gnd)
    got = rec.pop("_ranks_aqe")
    mh = oracle.compute_map_revisited(got, gnd)
    rec["max_abs_map_difference_alpha_qe"] = float(max(abs(a - b) for a, b in zip(mh, mo)))
    assert rec["max_abs_map_difference_alpha_qe"] <= 1e-6, "alpha-QE mAP differs from the oracle's"


def pinned_h2d_rate(torch, nbytes=1 << 30):
    """GB/s of a pinned host -> device copy on this box (the roof of any call that takes a host gallery)."""
    pinned = torch.empty(nbytes // 4, dtype=torch.float32, pin_memory=True)
    pinned.fill_(1.0)
    dev = torch.empty(nbytes // 4, dtype=torch.float32, device="cuda")
    best = None
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dev.copy_(pinned, non_blocking=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    del pinned, dev
    return nbytes / best / 1e9


def dropin_block(job, args):
    """The call the reference's entry points make (src/test_rOP1m.py:136-139,155-159; src/offline.py:107-118; src/online.py:95-102,
    132): `matching_<method>(K, vecs.T, qvecs.T[, dataset])` with `vecs` a HOST [D, N] array -- float32 at BASELINE size
    (2048 x 1 005 994: 8.2 GB over PCIe, then the [D, N]-layout ingest kernel), and the float64 case of src/online.py:96 at
    131 072 columns -- and the `time_per_query` it returns: stateless, stateful first call (build; the prepared-gallery file is
    written behind the call), stateful cached, and from the file in a fresh cache."""
    import shutil
    import numpy as np
    import torch
    from isehr_amd import _lib, nnsearch
    d, n, nq, k, stream = args.dim, WORKLOADS["roxford5k+1m"][0], 70, args.topk, job.stream
    rate = pinned_h2d_rate(torch)
    vecs = np.empty((d, n), dtype=np.float32)
    blk = 131072
    for r0 in range(0, n, blk):
        m = min(blk, n - r0)
        raw = torch.empty((m, d), dtype=torch.float32, device=job.dev)
        _lib.synth_fill_device(raw.data_ptr(), args.seed, r0, m, d, stream)
        vecs[:, r0:r0 + m] = raw.t().contiguous().cpu().numpy()
    qraw = torch.empty((nq, d), dtype=torch.float32, device=job.dev)
    _lib.synth_fill_device(qraw.data_ptr(), args.seed + 1, 0, nq, d, stream)
    qvecs = np.ascontiguousarray(qraw.t().cpu().numpy())
    del raw, qraw
    torch.cuda.empty_cache()
    _lib.set_global_option("image_dtype", 1 if args.image_dtype == "f16" else 0)
    ds = "bench_dropin"
    shutil.rmtree(os.path.join("outputs", ds), ignore_errors=True)
    calls = {}

    def call(name, **kw):
        t0 = time.time()
        idx, tpq = nnsearch.matching_HIP(k, vecs.T, qvecs.T, **kw)
        calls[name] = {"time_per_query_s": tpq, "wall_s": time.time() - t0, "gallery": dict(nnsearch.last_timing)}
        return idx
    try:
        i0 = call("stateless_first")
        i1 = call("stateless")
        i2 = call("dataset_first_call", dataset=ds, ifgenerate=True)
        i3 = call("dataset_cached", dataset=ds)
        t0 = time.time()
        nnsearch.wait_for_saves()
        save_wait = time.time() - t0
        calls["dataset_first_call"]["file_written_behind_the_call"] = dict(nnsearch.last_timing.get("save") or {},
                                                                         waited_s_after_the_calls=save_wait)
        nnsearch.drop_cached_galleries()
        i4 = call("dataset_from_file", dataset=ds)
        nnsearch.drop_cached_galleries()
        same = bool(all(np.array_equal(i0, x) for x in (i1, i2, i3, i4)))
        v64 = vecs[:, :131072].astype(np.float64)
        q64 = qvecs.astype(np.float64)
        nnsearch.matching_HIP(k, v64.T, q64.T)
        t0 = time.time()
        i64, tpq64 = nnsearch.matching_HIP(k, v64.T, q64.T)
        f64 = {"columns": 131072, "bytes": int(v64.nbytes), "time_per_query_s": tpq64, "wall_s": time.time() - t0,
               "h2d_equivalent_GBps": v64.nbytes / (tpq64 * nq) / 1e9,
               "equals_f32_answer_of_the_same_rows": bool(np.array_equal(
                   i64, nnsearch.matching_HIP(k, vecs[:, :131072].T, qvecs.T)[0]))}
        # the ingest kernels alone, both layouts resident on the device (warm: the second of two calls each): the one-pass kernel
        # of the [D, N] layout against the row-major one (VERDICT r04 #1 iii: <= 1.3 x)
        dn = torch.from_numpy(vecs).to(job.dev)                                    # [D, N] on the device
        dev_ingest = {}
        for name, kw in (("dn_layout", dict(row_stride=1, col_stride=n)), ("row_major", dict())):
            src = dn if name == "dn_layout" else dn.t().contiguous()
            best = None
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                gdev = _lib.Gallery.from_device_ptr(src.data_ptr(), n, d, norm_mode=_lib.NORM_L2, device=job.dev_index, **kw)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                gdev.close()
                best = dt if best is None else min(best, dt)
            dev_ingest[name + "_s"] = round(best, 5)
            del src
        del dn
        torch.cuda.empty_cache()
        dev_ingest["dn_over_row_major"] = round(dev_ingest["dn_layout_s"] / dev_ingest["row_major_s"], 3)
        dev_ingest["note"] = "Gallery.from_device_ptr incl. the allocation of the gallery, best of 3; N = %d is no multiple of 4: " \
                             "every column of the [D, N] array starts at another 8-byte residue" % n
    finally:
        nnsearch.drop_cached_galleries()
        shutil.rmtree(os.path.join("outputs", ds), ignore_errors=True)
    st = calls["stateless"]
    return {"layout": "host float32 [D, N] = [2048, %d], passed as vecs.T (row stride 1, column stride N)" % n,
            "queries": nq, "topk": k, "bytes": int(vecs.nbytes), "pinned_h2d_GBps": rate,
            "stateless_h2d_equivalent_GBps": vecs.nbytes / st["wall_s"] / 1e9,
            "stateless_frac_of_pinned_h2d": vecs.nbytes / st["wall_s"] / 1e9 / rate,
            "calls": calls, "same_answers": same, "float64_online_case": f64, "device_ingest_both_layouts": dev_ingest,
            "host_ingest": "row blocks of ~32 MiB copied by the runtime from the caller's pageable array, block i + 1 under the ingest "
                           "of block i, no staging the size of the gallery (mi_set_global_option host_ingest = 1)"}


MAP_DATASETS = (("roxford5k-sized (configs[0])", 4993), ("roxford5k+rparis6k-sized (configs[1])", 4993 + 6322))


def _planted(args, n):
    from isehr_amd.synth import planted_dataset
    # noise bands chosen so that the hard positives (cosine ~0.10) sit where the best distractors of a 5 k .. 11 k gallery do
    # (~0.08) and the junk band below them: mAP is then a non-trivial number that moves when a single rank moves
    return planted_dataset(args.seed, n, args.dim, 70, n_pos=(20, 60), sigmas=(2.0, 10.0, 16.0))


def map_block(args, device):
    """BASELINE.json's metric is queries/sec AND mAP: rOxford5k / +rParis6k-sized planted datasets (70 queries,
    clusters of positives labelled easy / hard / junk by their noise band; configs[0] and configs[1] sizes), ranked to K = 100
    by the HIP matcher exactly as src/test_rOP1m.py:155-159 does it (`matching_L2(K, vecs.T, qvecs.T)`, `ranks = idx.T`,
    compute_map_and_print).  The CPU restatement's mAP and time per query of the same datasets are added by the cpu_baseline
    leg (cpu_baseline_map), which also asserts that the two agree to 1e-6."""
    from isehr_amd import evaluate
    from isehr_amd.nnsearch import matching_HIP
    out = []
    for name, n in MAP_DATASETS:
        vecs, qv, gnd = _planted(args, n)
        matching_HIP(args.topk, vecs.T, qv.T, device=device)                        # warm (library, allocator)
        idx, tpq = matching_HIP(args.topk, vecs.T, qv.T, device=device)            # stateless call: ingest inside its timer
        m = evaluate.compute_map_revisited(idx.T, gnd)
        out.append({"dataset": name, "gallery_rows": n, "queries": 70, "topk": args.topk,
                    "map": {"E": m[0], "M": m[1], "H": m[2]}, "time_per_query_s": tpq})
    return out


def cpu_baseline_map(args, recs):
    """cpu_baseline leg, second half: the oracle's matching_l2 (the reference's numpy path, timer over normalisation + ranking
    like src/utils/nnsearch.py:688-705) on the planted datasets of map_block, its mAP by the oracle's evaluator, and the
    assertion that the HIP ranks' mAP equals it to 1e-6.  The oracle is the checker here, nothing else."""
    import oracle
    for rec, (_, n) in zip(recs, MAP_DATASETS):
        vecs, qv, gnd = _planted(args, n)
        t0 = time.time()
        ref = oracle.matching_l2(args.topk, vecs.T, qv.T)
        rec["time_per_query_oracle_s"] = (time.time() - t0) / 70
        mo = oracle.compute_map_revisited(ref.T, gnd)
        rec["map_oracle"] = {"E": mo[0], "M": mo[1], "H": mo[2]}
        m = (rec["map"]["E"], rec["map"]["M"], rec["map"]["H"])
        rec["max_abs_map_difference"] = float(max(abs(a - b) for a, b in zip(m, mo)))
        assert rec["max_abs_map_difference"] <= 1e-6, "mAP of the HIP ranks differs from the oracle's: %r" % rec


def roofline_of(res, args, world, with_traffic):
    st, nq, elapsed = res["st"], res["nq"], res["elapsed"]
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
    if with_traffic and os.path.exists(tp) and not hbm_bound:
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
    lms = res.get("launch_ms") or []
    if len(lms) >= 10:
        # is the timed region steady state?  (after an idle gap of a few ms the first ~10 launches run slow)
        roof["launch_ms_first5"] = sum(lms[:5]) / 5
        roof["launch_ms_last5"] = sum(lms[-5:]) / 5
    if clock and not hbm_bound:
        # the chip lowers its shader clock under MFMA load (DVFS): the dense peak it offers at the clock measured
        # INSIDE the timed launches (s_memtime / s_memrealtime, median over waves) next to the nominal 2.4 GHz peak
        roof["in_kernel_clock_mhz"] = clock
        roof["peak_at_clock"] = MFMA_BF16_PEAK_TFLOPS * clock / 2400.0
        roof["frac_at_clock"] = achieved / roof["peak_at_clock"] if achieved else None
    return roof


def model_rank_step(queries, rows):
    """The one-GPU MODEL of a rank's step for (queries per rank, gallery rows per rank): scripts/layout_model.sh measures every
    layout of a 2- / 4- / 8-GPU run of the benchmarked gallery on ONE GPU (protocol on a one-rank RCCL group) and leaves
    profiles/layout_model.json.  None when the shape is not in the table."""
    tp = os.path.join(ROOT, "profiles", "layout_model.json")
    try:
        tab = json.load(open(tp))
    except Exception:
        return None, None
    for e in tab.get("entries", []):
        if e["queries"] == queries and abs(e["rows"] - rows) <= 8:
            return e["ms"], tab.get("source")
    return None, tab.get("source")


def multi_gpu_summary(res, ms_per_step, layout_requested):
    """What the first hardware run of N ranks is read by: the layout `auto` chose, the measured step (max over
    ranks) against the one-GPU model of the same per-rank shape -- the model has no second rank to wait for, so the ratio is what
    the real collectives and the slowest peer cost -- and the two all-gathers, maxima over the ranks."""
    model_ms, src = model_rank_step(res["nq"], res["hi"] - res["lo"])
    ph = res.get("phases") or {}
    return {"layout": "%dx%d" % (res["gq"], res["gs"]), "layout_requested": layout_requested,
            "queries_per_rank": res["nq"], "rows_per_rank": res["hi"] - res["lo"],
            "rank_step_ms": ms_per_step, "model_rank_step_ms": model_ms, "model_source": src,
            "strong_scaling_vs_model": (model_ms / ms_per_step) if model_ms else None,
            "allgather1_ms_max_over_ranks": ph.get("allgather1"), "allgather2_ms_max_over_ranks": ph.get("allgather2"),
            "expectation": "DESIGN section 7: the one-GPU models put the 1 M-row gallery at ~0.6 of 8 x the one-GPU rate (the per-rank "
                           "shards leave the MFMA-bound regime) and only the 10 M-row gallery (scale_10m) near 0.9"}


def selftest_main(args, json_fd):
    """ISEHR_BENCH_SELFTEST=1 (tests/test_bench_cli_cpu.py, no GPU): the rank-process plumbing of main() -- process group with
    its timeout, the early headline line, a secondary block under its deadline, the exit codes -- over gloo on the CPU, with
    a stand-in workload (one all-reduce).  ISEHR_BENCH_TEST_HANG="<block>:<rank>" makes that rank sleep inside that block."""
    from datetime import timedelta
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    emitter = Emitter(json_fd, rank)
    with emitter.deadline("headline"):
        if world > 1:
            dist.init_process_group("gloo", timeout=timedelta(seconds=int(os.environ.get("ISEHR_BENCH_PG_TIMEOUT_S",
                                                                                           PG_TIMEOUT_S))))
        t = torch.ones(1)
        if world > 1:
            dist.all_reduce(t)
        out = {"metric": "queries/sec", "value": float(t.item()), "unit": "queries/s", "n_gpus": world, "selftest": True,
               "config": {"rccl_ranks": dist.get_world_size() if world > 1 else 1}} if rank == 0 else None
        emitter.out = out
        emitter.write(False)
    hang = os.environ.get("ISEHR_BENCH_TEST_HANG", "")

    def blk():
        if hang == "row_shard_1xN:%d" % rank:
            time.sleep(3600)
        if world > 1:
            dist.barrier()
        if rank == 0:
            out["row_shard_1xN"] = {"value": 1.0}
    emitter.block("row_shard_1xN", blk, world)
    emitter.write(True)
    if world > 1:
        with emitter.deadline("shutdown"):
            dist.barrier()
            dist.destroy_process_group()
    return 0


def main():
    args = parse()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        # no launcher: be the launcher.  Nothing below this line has run yet -- torch is not imported, the GPU untouched.
        sys.exit(launch_ranks(args.gpus))
    if env_world is not None and int(env_world) != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but the launcher started WORLD_SIZE=%s ranks\n" % (args.gpus, env_world))
        sys.exit(2)
    # stdout carries the ONE JSON line and nothing else: libraries write banners to file descriptor 1 (RCCL prints its
    # version block there when a communicator is created, on every rank), so fd 1 is pointed at stderr for the whole run
    # and the JSON line goes to the saved descriptor at the end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    # dmabuf IPC for RCCL / device-memory sharing between rank processes: the runtime reads this when it initialises, i.e.
    # before anything below touches the GPU (a launcher's environment normally carries it already)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if os.environ.get("ISEHR_BENCH_SELFTEST") == "1":
        return selftest_main(args, json_fd)
    import torch  # noqa: F401
    import torch.distributed as dist
    import isehr_amd  # noqa: F401

    emitter = Emitter(json_fd, int(os.environ.get("RANK", "0")))
    with emitter.deadline("headline"):
        job = Job(args)
    world, rank = job.world, job.rank
    n_total = args.rows or WORKLOADS[args.workload][0]
    d, k = args.dim, args.topk
    default_shape = (args.workload == "roxford5k+1m" and not args.rows and args.queries == 1024 and d == 2048)
    plain = not (args.diagnostic or args.force_protocol or args.with_aqe or args.option or args.graph or args.pipeline)
    scale_10m = args.scale_10m == "on" or (args.scale_10m == "auto" and default_shape and plain and not args.async_tail)
    # the headline's tail mode: one GPU and batches for the tile kernel -> the deferred tail (the throughput mode); `roofline`
    # then comes from a synchronous block of the same length on the same gallery (the undisturbed launch)
    auto_tail = args.async_tail is None
    async_tail = (3 if (world == 1 and args.queries > 128 and plain) else 0) if auto_tail else args.async_tail
    args.async_tail = async_tail
    # the blocks beyond the headline that make the default one-GPU line cover every BASELINE config
    extra_blocks = world == 1 and default_shape and plain and not args.diagnostic and args.extra_blocks != "off"
    only = set(x for x in args.blocks.split(",") if x)

    def want(name):
        return not only or name in only
    scale_10m = scale_10m and want("scale_10m")

    with emitter.deadline("headline"):
        res = run_workload(job, n_total, args.queries, args.image_dtype, args.steps, args.warmup, args.layout,
                           with_aqe=args.with_aqe, async_tail=async_tail, pipeline=args.pipeline, options=args.option,
                           check=not args.diagnostic, warm_ingest=(n_total * d * 4 <= 16 << 30), keep=True, graph=args.graph,
                           also_stream=(world > 1 and plain), phases=(10 if world > 1 or args.force_protocol else 0),
                           defer_extras=world > 1)
        gal = res.pop("gal")
        q_last_pool = res.pop("q_last_pool")
        out = None
        if rank == 0:
            elapsed, st, nq, gq, gs = res["elapsed"], res["st"], res["nq"], res["gq"], res["gs"]
            ms_step = elapsed / args.steps * 1e3
            roof = roofline_of(res, args, world, with_traffic=default_shape and world == 1)
            out = {
                "metric": "queries/sec", "value": args.queries * args.steps / elapsed, "unit": "queries/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
                "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": args.image_dtype,
                "data": "synthetic",
                "config": {"workload": WORKLOADS[args.workload][1] if not args.rows else "synthetic gallery",
                           "gallery_rows": n_total, "dim": d, "queries_per_step": args.queries, "topk": k,
                           "rccl_ranks": job.comm_ranks, "comm_backend": job.comm_backend,
                           "parallelism": ("row-shard x%d" % world if gq == 1 else
                                           "%d query groups (%d queries of every batch each, no exchange between groups) x %d "
                                           "row shards per group" % (gq, nq, gs)) +
                                          (" (two-phase protocol over RCCL forced on one rank)"
                                           if args.force_protocol and world == 1 else ""),
                           "alpha_qe": bool(args.with_aqe),
                           "launch": "hipGraph replay of the per-batch launch sequence" if res["graph"] else "eager",
                           "collectives": ("asynchronous, three batches in flight (search_stream)" if res["use_stream"] else
                                           "synchronous per batch") if res["protocol"] else None,
                           "value_mode": ("pipelined: deferred tail (async_tail 3), every result joined inside the timed region; "
                                          "`synchronous` holds the same steps with the tail on the caller's stream, and "
                                          "`roofline` is quoted on those undisturbed launches") if res["pipelined"] and auto_tail
                                         else ("pipelined (async_tail %d)" % async_tail if res["pipelined"] else "synchronous"),
                           "tail": ({1: "re-score + sort of batch i on the handle's own stream from the end of its phase 1: beside "
                                        "the query ingest, bootstrap AND scoring launch of batch i+1",
                                     2: "re-score + sort of batch i on the handle's own stream beside the query ingest + "
                                        "bootstrap of batch i+1 only; its scoring launch waits for the tail",
                                     3: "deferred: re-score + sort of batch i enqueued by the call of batch i+1 right before "
                                        "its scoring launch, and runs beside that launch only"}[async_tail] +
                                    "; every result joined inside the timed region") if res["pipelined"] else "same stream",
                           "exact": "%s MFMA filter + f64 re-score certificate" % args.image_dtype,
                           "calibrate_launches": args.calibrate,
                           # matching_L2's own timer spans the normalisation of the gallery too (src/utils/nnsearch.py:688-705):
                           # ingest_s = one normalisation + layout pass over the resident raw rows in a warm process (the second
                           # ingest of this run; the first one, which also initialises the library, is ingest_first_s), and the
                           # rate of ONE call that prepares the gallery and answers one batch
                           "ingest_s": round(res["ingest_s"], 4), "ingest_first_s": round(res["ingest_first_s"], 3),
                           "ingest_kernel_s": round(res["ingest_kernel_s"], 5) if res["ingest_kernel_s"] else None,
                           "ingest_launch_ms": ([round(x, 3) for x in res["ingest_launch_ms"]] if res["ingest_launch_ms"] else None),
                           "queries_per_s_incl_gallery_ingest_per_call_of_one_batch": nq / (ms_step * 1e-3 + res["ingest_s"]),
                           # measured, not composed: create + one batch (first search of the handle) + destroy, wall clock
                           "stateless_call_s": round(res["stateless_call_s"], 5) if res["stateless_call_s"] else None,
                           "queries_per_s_of_one_stateless_call": (nq / res["stateless_call_s"]) if res["stateless_call_s"] else None,
                           "candidates_per_query": st["candidates"] / max(1, st["queries"]),
                           "survivors_per_query": st["survivors"] / max(1, st["queries"]),
                           "score_check": ("16 queries x top-%d re-computed in float64: max |d| %.2e; completeness: %s"
                                           % (k, res["worst"], res["dense_check"])) if not args.diagnostic else None},
                "roofline": roof,
            }
        emitter.out = out
        # THE HEADLINE IS SAFE FROM HERE ON: the line goes out now and again, complete, at the end (the driver reads the last one)
        if out is not None:
            out["config"]["rank_devices"] = [list(x) for x in job.rank_devices]
            out["config"]["rank_devices_distinct"] = job.devices_distinct
            out["config"]["rccl_ranks_equal_n_gpus"] = job.comm_ranks == args.gpus
            out["blocks_after_the_headline"] = "each under a deadline; {\"error\": \"timeout\"} + a non-zero exit if one hangs"
        emitter.write(False)

    selftest_hang = os.environ.get("ISEHR_BENCH_TEST_HANG", "")      # "<block>:<rank>": that rank sleeps in that block (tests)

    def maybe_hang(name):
        if selftest_hang == "%s:%d" % (name, rank):
            time.sleep(3600)

    if res.get("later"):
        # (every rank runs them: they are collective; the headline line is out already)
        def extras_block():
            maybe_hang("pipelined_collectives")
            return res.pop("later")()
        ex = emitter.block("pipelined_collectives", extras_block, world)
        if ex:
            res.update(ex)
    if rank == 0:
        if res.get("stream"):
            out["pipelined_collectives"] = res["stream"]
        if res.get("phases"):
            out["protocol_phases_ms_max_over_ranks"] = res["phases"]
        if world > 1:
            out["multi_gpu"] = multi_gpu_summary(res, out["ms_per_step"], args.layout)
    if world == 1 and res["pipelined"] and auto_tail:
        # the same steps with the tail on the caller's stream: the undisturbed scoring launch (roofline) and what the
        # synchronous mode delivers on this box
        def sync_block():
            sb = synchronous_block(job, gal, args, args.steps)
            roof_sync = roofline_of(sb, args, world, with_traffic=default_shape)
            roof_sync["measured_on"] = ("`synchronous` block: %d steps on the same gallery with the tail on the caller's "
                                        "stream (the launch runs undisturbed); the headline's launches, which share the "
                                        "device with the deferred tail, are under `pipelined`" % sb["steps"])
            roof = out["roofline"]
            roof_sync["pipelined"] = {kk: roof[kk] for kk in ("achieved", "frac", "avg_launch_ms", "kernel_share_of_step",
                                                             "launches", "launch_ms_first5", "launch_ms_last5",
                                                             "in_kernel_clock_mhz", "frac_at_clock") if kk in roof}
            out["roofline"] = roof_sync
            out["synchronous"] = {"value": args.queries * sb["steps"] / sb["elapsed"], "unit": "queries/s",
                                  "steps": sb["steps"], "ms_per_step": sb["elapsed"] / sb["steps"] * 1e3,
                                  "avg_launch_ms": roof_sync["avg_launch_ms"],
                                  "kernel_share_of_step": roof_sync["kernel_share_of_step"],
                                  "equals_pipelined_answer": sb["equals_pipelined_answer"],
                                  "flagged_batches": sb["st"]["overflow_batches"] + sb["st"]["spec_retries"]}
            # like-for-like with the records of rounds 1-3, whose `value` was this mode
            out["value_synchronous"] = out["synchronous"]["value"]
            out["value_pipelined"] = out["value"]
            if out["value_synchronous"] > out["value"]:
                # both modes ran the same K steps between the same brackets on this box; under the board's power cap the
                # deferred tail is a wash and loses on some boxes: the headline is the faster of the two
                out["value"], out["ms_per_step"] = out["value_synchronous"], out["synchronous"]["ms_per_step"]
                out["config"]["value_mode"] = ("synchronous (faster than the deferred tail on this box: %.0f vs %.0f queries/s; "
                                               "`value_pipelined` holds the other)" % (out["value_synchronous"], out["value_pipelined"]))
        emitter.block("synchronous", sync_block, world)
    elif rank == 0 and not res["pipelined"]:
        out["value_synchronous"] = out["value"]
    if extra_blocks:
        for nq_small, name in ((70, "q70"), (1, "q1")):
            if not want(name):
                continue
            r = emitter.block(name, lambda nq_small=nq_small: small_batch_block(job, gal, args, nq_small, 100), world)
            if r is not None:
                out[name] = r
        if want("qsweep"):
            r = emitter.block("qsweep", lambda: qsweep_block(job, gal, args), world)
            if r is not None:
                out["qsweep"] = r
        if want("online"):
            r = emitter.block("online", lambda: online_block(job, gal, args), world)
            if r is not None:
                out["online"] = r
    if world == 1 and not args.no_cpu_baseline and want("cpu_baseline"):
        r = emitter.block("cpu_baseline", lambda: cpu_baseline(gal, q_last_pool.cpu().numpy(), n_total, args), world)
        if r is not None:
            out["cpu_baseline"] = r
            out["vs_cpu_baseline"] = {"port_one_thread": out["value"] / r["value"],
                                      "blas_full_width": out["value"] / r["value_blas"]}
    gal.close()
    del gal, q_last_pool
    torch.cuda.empty_cache()
    emitter.write(False)

    multi = world > 1 and plain and (args.multi_gpu_blocks == "on" or (args.multi_gpu_blocks == "auto" and default_shape))
    if multi:
        # north_star's literal partition on the benchmarked gallery: row shards only (1 x N), every rank scores all queries
        # against its rows, two all-gathers per batch -- synchronous and with the pipelined collectives, plus what every
        # stage of the protocol costs (events behind every stage, maxima over the ranks)
        def row_shards():
            maybe_hang("row_shard_1xN")
            r1 = run_workload(job, n_total, args.queries, args.image_dtype, args.steps, args.warmup, "1x%d" % world,
                              check=True, keep=False, also_stream=True, phases=10)
            if rank == 0:
                rf = roofline_of(r1, args, world, with_traffic=False)
                out["row_shard_1xN"] = {
                    "parallelism": "row-shard x%d (two-phase protocol, RCCL all-gathers of the per-shard top-K)" % world,
                    "value": args.queries * r1["steps"] / r1["elapsed"], "unit": "queries/s", "steps": r1["steps"],
                    "ms_per_step": r1["elapsed"] / r1["steps"] * 1e3, "scoring_frac_of_mfma_peak": rf["frac"],
                    "scoring_share_of_step": rf["kernel_share_of_step"], "pipelined_collectives": r1.get("stream"),
                    "protocol_phases_ms_max_over_ranks": r1.get("phases"), "score_check": r1["dense_check"]}
        emitter.block("row_shard_1xN", row_shards, world)
        emitter.write(False)

        # and the layout a deployment would pick for a gallery that fits one GPU (12 GB of 288): every rank holds ALL rows and
        # the job's batches are dealt to the ranks, each answered by the one-GPU pipeline (deferred tail), nothing exchanged
        def replicas():
            maybe_hang("batch_replicas")
            rr = run_workload(job, n_total, args.queries, args.image_dtype, args.steps, args.warmup, "replicas", check=True,
                              keep=False, async_tail=3 if args.queries > 128 else 0)
            if rank == 0:
                out["batch_replicas"] = {
                    "parallelism": "%d replicas of the whole gallery; batch i of the job is answered by rank i mod %d with the "
                                   "one-GPU pipeline; no collective on the search path" % (world, world),
                    "value": args.queries * rr["job_steps"] / rr["elapsed"], "unit": "queries/s", "steps": rr["job_steps"],
                    "steps_of_rank_0": rr["steps"], "ms_per_step": rr["elapsed"] / rr["job_steps"] * 1e3,
                    "score_check": rr["dense_check"]}
        emitter.block("batch_replicas", replicas, world)
        emitter.write(False)

    if scale_10m:
        # BASELINE configs[3]: 10 M x 2048 rows, bf16 image, 1024-query batches, row-sharded over the job's ranks
        # (1 x N: every rank scores all queries against its rows; the two all-gathers of the protocol per batch)
        def ten_million():
            maybe_hang("scale_10m")
            r10 = run_workload(job, args.scale_10m_rows, 1024, "bf16", args.scale_10m_steps, 2, "1x%d" % world,
                               check=True, keep=False, also_stream=world > 1, phases=(5 if world > 1 else 0))
            if rank == 0:
                roof10 = roofline_of(r10, args, world, with_traffic=False)
                out["scale_10m"] = {
                    "gallery_rows": r10["n_total"], "rows_per_rank": r10["hi"] - r10["lo"], "image": "bf16",
                    "queries_per_step": 1024, "steps": r10["steps"], "ms_per_step": r10["elapsed"] / r10["steps"] * 1e3,
                    "value": 1024 * r10["steps"] / r10["elapsed"], "unit": "queries/s", "n_gpus": world,
                    "rccl_ranks": job.comm_ranks, "parallelism": "row-shard x%d" % world,
                    "ingest_s": round(r10["ingest_first_s"], 3),
                    "ingest_s_is": "first-touch allocation of the gallery's buffers (~123 GB on one GPU: page-table work, "
                                   "`alloc_s`) + the ingest kernel over the resident rows (`ingest_kernel_s`)"
                                   if r10.get("alloc_s") is not None else "allocation + ingest kernel of this rank's shard",
                    "alloc_s": round(r10["alloc_s"], 3) if r10.get("alloc_s") is not None else None,
                    "ingest_kernel_s": round(r10["ingest_kernel_s"], 4) if r10.get("ingest_kernel_s") is not None else None,
                    "scoring_frac_of_mfma_peak": roof10["frac"], "scoring_share_of_step": roof10["kernel_share_of_step"],
                    "scoring_launches_per_step": r10["st"]["gemm_launches"] / max(1, r10["steps"]),
                    "pipelined_collectives": r10.get("stream"), "protocol_phases_ms_max_over_ranks": r10.get("phases"),
                    "score_check": "16 queries x top-%d re-computed in float64: max |d| %.2e; completeness: %s"
                                   % (k, r10["worst"], r10["dense_check"])}
        emitter.block("scale_10m", ten_million, world)
        torch.cuda.empty_cache()
        emitter.write(False)

    if rank == 0 and default_shape and plain and not args.diagnostic and want("map"):
        # the metric's other half: mAP (E / M / H) on planted rOxford5k- / +rParis6k-sized datasets, HIP ranks vs the oracle's
        def map_blk():
            out["map"] = map_block(args, job.dev_index)
            if world == 1 and not args.no_cpu_baseline:
                cpu_baseline_map(args, out["map"])
        emitter.block("map", map_blk, 1)

    if extra_blocks and want("whiten"):
        r = emitter.block("whiten", lambda: whiten_block(job, args), world)
        if r is not None:
            out["whiten"] = r
        emitter.write(False)
    if extra_blocks and want("hard_data"):
        r = emitter.block("hard_data", lambda: hard_data_block(job, args, out.get("value_synchronous") or out["value"]), world)
        if r is not None:
            out["hard_data"] = r
        emitter.write(False)
    if extra_blocks:
        if want("aqe_rparis_1m"):
            r = emitter.block("aqe_rparis_1m", lambda: aqe_rparis_block(job, args), world)
            if r is not None:
                out["aqe_rparis_1m"] = r

        def qge_blk():
            rec = qge_small_block(args, job.dev_index)
            if not args.no_cpu_baseline:
                cpu_baseline_qge(args, rec)
            rec.pop("_ranks_aqe", None)
            out["qge_small"] = rec
        if want("qge_small"):
            emitter.block("qge_small", qge_blk, world)
        emitter.write(False)
        if want("dropin"):
            r = emitter.block("dropin", lambda: dropin_block(job, args), world)
            if r is not None:
                out["dropin"] = r

    emitter.write(True)
    if dist.is_initialized():
        with emitter.deadline("shutdown"):
            dist.barrier()
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
