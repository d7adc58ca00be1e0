"""Rank program of tests/test_gpu_sharded_ranks.py: started as fresh processes by `python -m torch.distributed.run`
(the launcher itself never touches the GPU).  All ranks share cuda:0 (ISEHR_SHARE_GPU=1) and talk over gloo, because
RCCL refuses two ranks on one device; the code path is the one bench.py --gpus N runs over RCCL on a whole node:
ShardedGallery.__init__ (agreement on norm bounds / image type), search(verify=True) and aqe_search with world > 1.

Every rank writes <out>.<rank>.npz with its answers; rank 0 also computes the single-shard answers of the same gallery."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--rows", type=int, default=60000)
    ap.add_argument("--dim", type=int, default=256)
    ap.add_argument("--queries", type=int, default=300)
    ap.add_argument("--topk", type=int, default=50)
    ap.add_argument("--hetero", action="store_true",
                    help="un-normalised gallery whose later rows are 60x larger: shards with different norm bounds, "
                         "the large-row shards fall back to a bf16 image on their own")
    ap.add_argument("--diffusion", action="store_true",
                    help="node-sharded offline diffusion (isehr_amd.diffusion.Diffusion under torch.distributed) instead")
    ap.add_argument("--layout", default="",
                    help="QGxRS: the job as query groups x row shards (isehr_amd.sharded.job_layout); every rank answers its "
                         "group's slice of the queries against its row shard")
    ap.add_argument("--rccl1", action="store_true",
                    help="ONE rank over RCCL (backend nccl): the two-phase protocol with its real collectives on a one-GPU box")
    a = ap.parse_args()
    if a.diffusion:
        return diffusion_main(a)
    if a.rccl1:
        return rccl1_main(a)
    if a.layout:
        return layout_main(a)
    import numpy as np
    import torch
    import torch.distributed as dist
    import isehr_amd  # noqa: F401
    from isehr_amd import _lib
    from isehr_amd.sharded import ShardedGallery, shard_bounds

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    n, d, nq, k = a.rows, a.dim, a.queries, a.topk
    raw = torch.empty((n, d), dtype=torch.float32, device=dev)
    _lib.synth_fill_device(raw.data_ptr(), 77, 0, n, d, stream)
    q = torch.empty((nq, d), dtype=torch.float32, device=dev)
    _lib.synth_fill_device(q.data_ptr(), 78, 0, nq, d, stream)
    torch.cuda.synchronize()
    norm = _lib.NORM_L2
    if a.hetero:
        norm = _lib.NORM_NONE
        raw *= 0.05
        raw[n // 3:] *= 60.0                      # rows of norm ~ 0.8 and ~ 48 in one gallery
        q *= 0.1
    lo, hi = shard_bounds(n, world, rank)
    shard = _lib.Gallery.from_device_ptr(raw[lo:hi].data_ptr(), hi - lo, d, norm_mode=norm, row_offset=lo)
    own_dtype = int(shard.get_option("image_dtype"))
    own_bounds = shard.norm_bounds()
    sg = ShardedGallery(shard)
    idx, sc = sg.search(q, k, verify=True)
    idx, sc = idx.clone(), sc.clone()
    aidx, asc, qx = sg.aqe_search(idx.t(), 3, 4.0, k)
    aidx, asc, qx = aidx.clone(), asc.clone(), qx.clone()
    # pipelined search (asynchronous all-gathers, two workspace slots): five different batches, answers of every one
    qs5 = []
    for i in range(5):
        t = torch.empty((nq if i % 2 == 0 else max(1, nq // 3), d), dtype=torch.float32, device=dev)
        _lib.synth_fill_device(t.data_ptr(), 200 + i, 0, t.shape[0], d, stream)
        if a.hetero:
            t *= 0.1
        qs5.append(t)
    seq = []
    for t in qs5:
        i_, s_ = sg.search(t, k)
        seq.append((i_.clone(), s_.clone()))
    pipe = [(i_.clone(), s_.clone()) for i_, s_ in sg.search_stream(qs5, k)]
    torch.cuda.synchronize()
    stream_ok = int(len(pipe) == 5 and all(torch.equal(p_[0], r_[0]) and torch.equal(p_[1], r_[1])
                                           for p_, r_ in zip(pipe, seq)))
    flagged = sg.any_flag()
    agreed_dtype = int(shard.get_option("image_dtype"))
    # what THIS shard's rows measure under the agreed image type (a shard that chose fp16 on its own was re-imaged as bf16
    # by the agreement, with larger rounding norms than `own_bounds`): a fresh handle of the same rows, never raised
    probe = _lib.Gallery.from_device_ptr(raw[lo:hi].data_ptr(), hi - lo, d, norm_mode=norm, row_offset=lo)
    if int(probe.get_option("image_dtype")) != agreed_dtype:
        probe.set_image_dtype(agreed_dtype)
    reimaged_bounds = probe.norm_bounds()
    probe.close()
    out = dict(idx=idx.cpu().numpy(), sc=sc.cpu().numpy(), aidx=aidx.cpu().numpy(), asc=asc.cpu().numpy(),
               qx=qx.cpu().numpy(), own_dtype=own_dtype, agreed_dtype=agreed_dtype,
               own_bounds=np.array(own_bounds), reimaged_bounds=np.array(reimaged_bounds),
               agreed_bounds=np.array(shard.norm_bounds()), flagged=int(flagged),
               lo=lo, hi=hi, stream_ok=stream_ok)
    if rank == 0:
        single = _lib.Gallery.from_device_ptr(raw.data_ptr(), n, d, norm_mode=norm)
        ridx = torch.empty((nq, k), dtype=torch.int64, device=dev)
        rsc = torch.empty((nq, k), dtype=torch.float32, device=dev)
        single.search_device(q.data_ptr(), nq, k, ridx.data_ptr(), rsc.data_ptr(), None, stream)
        torch.cuda.synchronize()
        assert single.flags() == 0
        ref_idx = ridx.cpu().numpy()
        r_aidx, r_asc, r_qx, _ = single.aqe_search(np.ascontiguousarray(ref_idx.T), 3, 4.0, k, return_qexp=True)
        out.update(ref_idx=ref_idx, ref_sc=rsc.cpu().numpy(), ref_aidx=r_aidx, ref_asc=r_asc, ref_qx=r_qx,
                   single_dtype=int(single.get_option("image_dtype")))
        single.close()
    np.savez(a.out + ".%d.npz" % rank, **out)
    dist.barrier()
    dist.destroy_process_group()
    shard.close()


def layout_main(a):
    """gq query groups x gs row shards: the rank's answers for ITS slice of the batch; rank 0 adds the single-GPU answers
    of the whole batch, so the test can check every group's slice for completeness (not just self-consistency)."""
    import numpy as np
    import torch
    import torch.distributed as dist
    import isehr_amd  # noqa: F401
    from isehr_amd import _lib
    from isehr_amd.sharded import ShardedGallery, shard_bounds, job_layout, layout_groups
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    n, d, nq, k = a.rows, a.dim, a.queries, a.topk
    gq, gs, qgroup, shard = job_layout(world, rank, nq, a.layout)
    group = layout_groups(gq, gs)[qgroup]
    raw = torch.empty((n, d), dtype=torch.float32, device=dev)
    _lib.synth_fill_device(raw.data_ptr(), 77, 0, n, d, stream)
    q = torch.empty((nq, d), dtype=torch.float32, device=dev)
    _lib.synth_fill_device(q.data_ptr(), 78, 0, nq, d, stream)
    torch.cuda.synchronize()
    lo, hi = shard_bounds(n, gs, shard)
    g = _lib.Gallery.from_device_ptr(raw[lo:hi].data_ptr(), hi - lo, d, row_offset=lo)
    sg = ShardedGallery(g, group=group)
    per = nq // gq
    mine = q[qgroup * per:(qgroup + 1) * per].contiguous()
    idx, sc = sg.search(mine, k, verify=True)
    idx, sc = idx.clone(), sc.clone()
    pipe = [(i_.clone(), s_.clone()) for i_, s_ in sg.search_stream([mine, mine[: max(1, per // 2)]], k)]
    torch.cuda.synchronize()
    out = dict(idx=idx.cpu().numpy(), sc=sc.cpu().numpy(), qgroup=qgroup, shard=shard, gq=gq, gs=gs, lo=lo, hi=hi,
               group_size=sg.world, flagged=int(sg.any_flag()),
               stream_ok=int(torch.equal(pipe[0][0], idx) and torch.equal(pipe[0][1], sc) and
                             torch.equal(pipe[1][0], idx[: max(1, per // 2)])))
    if rank == 0:
        single = _lib.Gallery.from_device_ptr(raw.data_ptr(), n, d)
        ridx = torch.empty((nq, k), dtype=torch.int64, device=dev)
        rsc = torch.empty((nq, k), dtype=torch.float32, device=dev)
        single.search_device(q.data_ptr(), nq, k, ridx.data_ptr(), rsc.data_ptr(), None, stream)
        torch.cuda.synchronize()
        out.update(ref_idx=ridx.cpu().numpy(), ref_sc=rsc.cpu().numpy())
        single.close()
    np.savez(a.out + ".%d.npz" % rank, **out)
    dist.barrier()
    dist.destroy_process_group()
    g.close()


def rccl1_main(a):
    """RCCL refuses two ranks on one device, so the collectives themselves (all_gather_into_tensor / all_reduce on device
    tensors, RCCL's own stream ordered against the stream the library launches on) are exercised with a group of ONE rank:
    ShardedGallery(force_protocol=True) runs phase 1 -> all-gather -> K-th -> phase 2 -> all-gather -> merge exactly as it
    does on 8 GPUs.  Ten different batches are issued back to back without a host synchronisation, on a side stream."""
    import numpy as np
    import torch
    import torch.distributed as dist
    import isehr_amd  # noqa: F401
    from isehr_amd import _lib
    from isehr_amd.sharded import ShardedGallery
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    backend = os.environ.get("ISEHR_RCCL1_BACKEND", "nccl")             # "gloo": the same flow with host collectives
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group(backend)
    n, d, nq, k = 200000, 128, a.queries, a.topk
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        stream = torch.cuda.current_stream().cuda_stream
        raw = torch.empty((n, d), dtype=torch.float32, device=dev)
        _lib.synth_fill_device(raw.data_ptr(), 77, 0, n, d, stream)
        qs = []
        for i in range(10):
            q = torch.empty((nq, d), dtype=torch.float32, device=dev)
            _lib.synth_fill_device(q.data_ptr(), 100 + i, 0, nq, d, stream)
            qs.append(q)
        torch.cuda.synchronize()
        shard = _lib.Gallery.from_device_ptr(raw.data_ptr(), n, d)
        plain = ShardedGallery(shard)                                   # world == 1: the single-shard path
        ref = []
        for q in qs:
            i_, s_ = plain.search(q, k)
            ref.append((i_.clone(), s_.clone()))
        r_aidx, r_asc, r_qx = plain.aqe_search(ref[0][0].t(), 3, 4.0, k)
        r_aidx, r_asc, r_qx = r_aidx.clone(), r_asc.clone(), r_qx.clone()
        torch.cuda.synchronize()
        sg = ShardedGallery(shard, force_protocol=True)                 # the protocol + RCCL collectives, one rank
        got = []
        for q in qs:                                                    # no host synchronisation between batches
            i_, s_ = sg.search(q, k)
            got.append((i_.clone(), s_.clone()))
        aidx, asc, qx = (t.clone() for t in sg.aqe_search(got[0][0].t(), 3, 4.0, k))   # results live in shared buffers
        vi, vs = sg.search(qs[3], k, verify=True)
        vi, vs = vi.clone(), vs.clone()
        pipe = [(i_.clone(), s_.clone()) for i_, s_ in sg.search_stream(qs, k)]     # asynchronous collectives, 2 slots
        one = [(i_.clone(), s_.clone()) for i_, s_ in sg.search_stream(qs[:1], k)]
        torch.cuda.synchronize()
        flagged = sg.any_flag()
    eq = [int(torch.equal(g[0], r[0]) and torch.equal(g[1], r[1])) for g, r in zip(got, ref)]
    eq_aqe = [int(torch.equal(aidx, r_aidx)), int(torch.equal(asc, r_asc)), int(torch.equal(qx, r_qx))]
    eq_ver = int(torch.equal(vi, ref[3][0]) and torch.equal(vs, ref[3][1]))
    eq_pipe = [int(torch.equal(g[0], r[0]) and torch.equal(g[1], r[1])) for g, r in zip(pipe, ref)]
    eq_pipe.append(int(len(pipe) == len(ref) and len(one) == 1 and torch.equal(one[0][0], ref[0][0])))
    ok = all(eq) and all(eq_aqe) and eq_ver and all(eq_pipe)
    np.savez(a.out + ".0.npz", ok=int(ok), flagged=int(flagged), backend=dist.get_backend(), eq=np.array(eq),
             eq_aqe=np.array(eq_aqe), eq_ver=eq_ver, eq_pipe=np.array(eq_pipe),
             mism=np.array([int((g[0] != r[0]).sum().item()) for g, r in zip(got, ref)]))
    dist.barrier()
    dist.destroy_process_group()
    shard.close()


def diffusion_main(a):
    """Every rank holds the same (clustered) features; the CG solves are split by node range and all-gathered."""
    import numpy as np
    import torch
    import torch.distributed as dist
    import isehr_amd  # noqa: F401
    from isehr_amd.diffusion import Diffusion
    from isehr_amd.synth import synth_rows
    rank = int(os.environ["RANK"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    n, d, T, kd = 1501, 48, 300, 40                 # N not a multiple of the world size: ragged last node range
    f = synth_rows(5, 0, n, d) * 0.7 + 1.2 * synth_rows(6, 0, 30, d)[np.arange(n) % 30]
    f = (f / np.linalg.norm(f, axis=1, keepdims=True)).astype(np.float32)
    q = f[::211][:6] + 0.1 * synth_rows(7, 0, 6, d)
    D = Diffusion(f, cache_dir=None)
    off = D.get_offline_results(T, kd).tocsr()
    off.sort_indices()
    ranks, scores = D.search_online(q, 3, 200)
    out = dict(data=off.data, indices=off.indices, indptr=off.indptr, ranks=ranks, scores=scores)
    if rank == 0:
        ids, vals = D.gallery.diffusion_offline(T, kd)          # all nodes in this process
        out.update(ref_ids=ids, ref_vals=vals)
        r2, s2 = D.search_online(q, 3, 200)
        out.update(ref_ranks=r2, ref_scores=s2)
    np.savez(a.out + ".%d.npz" % rank, **out)
    dist.barrier()
    dist.destroy_process_group()
    D.close()


if __name__ == "__main__":
    main()
