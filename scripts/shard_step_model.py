"""What ONE rank of an G-way row-sharded search does per batch, measured on one GPU without the collectives:
G shards of the 1 005 994-row synthetic gallery are built on this device, the two-phase protocol runs over all of them
(stacking stands in for the all-gathers), and shard 0's phase 1 / K-th of the gathered lists / phase 2 / merge are timed
with events.  Prints a model of the strong-scaling efficiency (per-rank GPU time only; RCCL latency comes on top).

    python scripts/shard_step_model.py [G=8] [rows=1005994] [option=value ...]
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import isehr_amd  # noqa: F401,E402
from isehr_amd import _lib  # noqa: E402
from isehr_amd.sharded import shard_bounds  # noqa: E402


def main():
    G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1005994
    d, nq, k, reps = 2048, 1024, 100, 20
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    stream = torch.cuda.current_stream().cuda_stream
    shards = []
    opts = [o for o in sys.argv[3:] if not o.startswith("image_dtype=")]
    for o in sys.argv[3:]:
        if o.startswith("image_dtype="):                 # process-wide default of the galleries created below: 0 = bf16
            _lib.set_global_option("image_dtype", float(o.split("=")[1]))
    for r in range(G):
        lo, hi = shard_bounds(n, G, r)
        raw = torch.empty((hi - lo, d), dtype=torch.float32, device=dev)
        _lib.synth_fill_device(raw.data_ptr(), 1234, lo, hi - lo, d, stream)
        torch.cuda.synchronize()
        shards.append(_lib.Gallery.from_device_ptr(raw.data_ptr(), hi - lo, d, device=0, row_offset=lo))
        for opt in opts:
            name, val = opt.split("=")
            shards[-1].set_option(name, float(val))
        torch.cuda.synchronize()
        del raw
    q = torch.empty((nq, d), dtype=torch.float32, device=dev)
    _lib.synth_fill_device(q.data_ptr(), 1235, 0, nq, d, stream)
    approx = torch.empty((G, nq, k), dtype=torch.float32, device=dev)
    L = torch.empty((nq,), dtype=torch.float32, device=dev)
    pack = torch.empty((G, 2, nq, k), dtype=torch.int64, device=dev)
    sc = torch.empty((G, nq, k), dtype=torch.float32, device=dev)
    oi = torch.empty((nq, k), dtype=torch.int64, device=dev)
    os_ = torch.empty((nq, k), dtype=torch.float32, device=dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
    acc = [0.0] * 4
    for it in range(reps + 3):
        for r in range(1, G):
            shards[r].phase1_device(q.data_ptr(), nq, k, approx[r].data_ptr(), stream)
        ev[0].record()
        shards[0].phase1_device(q.data_ptr(), nq, k, approx[0].data_ptr(), stream)
        ev[1].record()
        _lib.kth_of_gathered_device(approx.data_ptr(), G, nq, k, L.data_ptr(), stream)
        ev[2].record()
        shards[0].phase2_device(nq, k, L.data_ptr(), pack[0, 1].data_ptr(), sc[0].data_ptr(), pack[0, 0].data_ptr(), stream)
        ev[3].record()
        for r in range(1, G):
            shards[r].phase2_device(nq, k, L.data_ptr(), pack[r, 1].data_ptr(), sc[r].data_ptr(), pack[r, 0].data_ptr(),
                                    stream)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.topk_merge_strided_device(pack[0, 0].data_ptr(), pack[0, 1].data_ptr(), 2 * nq * k, G, nq, k, oi.data_ptr(),
                                       os_.data_ptr(), stream)
        e1.record()
        torch.cuda.synchronize()
        if it >= 3:
            acc[0] += ev[0].elapsed_time(ev[1])
            acc[1] += ev[1].elapsed_time(ev[2])
            acc[2] += ev[2].elapsed_time(ev[3])
            acc[3] += e0.elapsed_time(e1)
    flags = [s.flags() for s in shards]
    single = _lib.Gallery  # noqa: F841
    t = [a / reps for a in acc]
    total = sum(t)
    print("G=%d shards of %d rows, %d queries, K=%d; options %s; flags %s" % (G, n, nq, k, sys.argv[3:], flags))
    print("rank 0 per batch: phase1 %.3f ms  kth-of-gathered %.3f  phase2 %.3f  merge %.3f  = %.3f ms (no collectives)"
          % (t[0], t[1], t[2], t[3], total))
    for s in shards:
        s.close()
    return total


if __name__ == "__main__":
    main()
