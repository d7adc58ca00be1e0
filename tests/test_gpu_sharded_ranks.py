"""GPU: the multi-rank code path itself -- ShardedGallery with world > 1 (packed all-gathers, strided merge, all-reduced
flags, sharded alpha-QE) -- run by 2 and 3 fresh rank processes that share the one GPU of the test box and talk over
gloo.  Answers must equal the single-shard answers bit for bit, on every rank."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(world, tmp_path, extra=()):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / ("w%d" % world))
    env = dict(os.environ, ISEHR_DIST_BACKEND="gloo", ISEHR_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "sharded_worker.py"),
           "--out", out, *extra]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return [np.load(out + ".%d.npz" % rk) for rk in range(world)]


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_ranks_equal_single_shard(world, tmp_path):
    res = _run(world, tmp_path)
    z0 = res[0]
    for z in res:
        assert int(z["flagged"]) == 0
        assert np.array_equal(z["idx"], z0["ref_idx"]) and np.array_equal(z["sc"], z0["ref_sc"])
        assert np.array_equal(z["aidx"], z0["ref_aidx"]) and np.array_equal(z["asc"], z0["ref_asc"])
        # the expanded query is the single-GPU one bit for bit (round 4: the shards exchange ROWS, summed in j order)
        assert np.array_equal(z["qx"], z0["ref_qx"].astype(np.float32))
        assert int(z["stream_ok"]) == 1                      # the pipelined search gives the answers of the sequential one
    assert res[-1]["hi"] == 60000 and all(int(a["hi"]) == int(b["lo"]) for a, b in zip(res, res[1:]))


def test_sharded_ranks_with_unequal_norm_bounds(tmp_path):
    """Raw (MI_NORM_NONE) gallery whose rows differ 60x in norm: the shards measure different rounding norms and the
    large-row shards pick a bf16 image by themselves.  ShardedGallery makes all of them use the coarsest image type and
    the maxima over all shards (otherwise a shard with a small margin can drop a row of the global top-K); the merged
    answers equal the single gallery's."""
    res = _run(3, tmp_path, ("--hetero",))
    z0 = res[0]
    own = [int(z["own_dtype"]) for z in res]
    assert own[0] == 1 and own[-1] == 0                          # fp16 on the small-row shard, bf16 on the large-row one
    assert all(int(z["agreed_dtype"]) == 0 for z in res)
    bounds = np.stack([z["agreed_bounds"] for z in res])
    assert (bounds == bounds[0]).all()
    # the agreed bounds are the maxima, over all shards, of the norms each shard measures under the AGREED image type (a
    # shard that chose fp16 on its own is re-imaged as bf16 and re-measures larger rounding norms): they cover every shard
    re = np.stack([z["reimaged_bounds"] for z in res])
    assert (bounds[0] >= re.max(0)).all() and np.allclose(bounds[0], re.max(0), rtol=0, atol=1e-7)
    assert (re[0] >= res[0]["own_bounds"] - 1e-7).all() and re[0][2] > 2.0 * res[0]["own_bounds"][2]   # fp16 -> bf16: coarser
    assert int(z0["single_dtype"]) == 0
    for z in res:
        assert np.array_equal(z["idx"], z0["ref_idx"]) and np.array_equal(z["sc"], z0["ref_sc"])
        assert np.array_equal(z["aidx"], z0["ref_aidx"]) and np.array_equal(z["asc"], z0["ref_asc"])
        assert int(z["stream_ok"]) == 1


def test_node_sharded_diffusion_equals_single_process(tmp_path):
    """SURVEY 8e, diffusion: the N truncated CG solves (src/utils/diffusion.py:15-19) split over 3 rank processes by
    node range and all-gathered; every rank ends with the offline matrix and the online ranks of the one-process run."""
    import scipy.sparse as sparse
    res = _run(3, tmp_path, ("--diffusion",))
    z0 = res[0]
    n, T = z0["ref_ids"].shape
    rows = np.repeat(np.arange(n), T)
    ref = sparse.csr_matrix((z0["ref_vals"].reshape(-1), (rows, z0["ref_ids"].reshape(-1))), shape=(n, n),
                            dtype=np.float32)
    ref.sort_indices()
    assert np.abs(ref.data).max() > 0.1
    for z in res:
        assert np.array_equal(z["indptr"], ref.indptr) and np.array_equal(z["indices"], ref.indices)
        assert np.array_equal(z["data"], ref.data)
        assert np.array_equal(z["ranks"], z0["ref_ranks"]) and np.array_equal(z["scores"], z0["ref_scores"])


def test_protocol_over_rccl_with_one_rank(tmp_path):
    """The collectives of the sharded search over RCCL itself (backend "nccl"), which a one-GPU box can only run with a
    group of one rank: ten batches back to back on a side stream, alpha-QE and a verified search -- equal to the
    single-shard path bit for bit."""
    res = _run(1, tmp_path, ("--rccl1",))
    z = res[0]
    assert str(z["backend"]) == os.environ.get("ISEHR_RCCL1_BACKEND", "nccl")
    assert int(z["ok"]) == 1 and int(z["flagged"]) == 0, (z["eq"], z["eq_aqe"], z["eq_ver"], z["eq_pipe"], z["mism"])


@pytest.mark.parametrize("world,layout", [(4, "2x2"), (4, "4x1"), (2, "2x1"), (4, "auto")])
def test_query_groups_times_row_shards(world, layout, tmp_path):
    """The 2-D layout of bench.py --gpus N (isehr_amd.sharded.job_layout): query groups answer disjoint slices of the batch,
    the ranks of a group shard the gallery and exchange inside the group only.  Every rank's slice must be THE single-GPU
    answer of those queries (completeness, not just self-consistency), synchronous and pipelined."""
    res = _run(world, tmp_path, ("--layout", layout, "--queries", "640", "--rows", "90000", "--dim", "128"))
    z0 = res[0]
    gq, gs = int(z0["gq"]), int(z0["gs"])
    assert gq * gs == world and (layout == "auto" or layout == "%dx%d" % (gq, gs))
    if layout == "auto":
        assert (gq, gs) == (2, 2)
    per = 640 // gq
    seen = set()
    for z in res:
        qg = int(z["qgroup"])
        assert int(z["group_size"]) == gs and int(z["flagged"]) == 0 and int(z["stream_ok"]) == 1
        assert np.array_equal(z["idx"], z0["ref_idx"][qg * per:(qg + 1) * per])
        assert np.array_equal(z["sc"], z0["ref_sc"][qg * per:(qg + 1) * per])
        seen.add((qg, int(z["shard"])))
    assert seen == {(a, b) for a in range(gq) for b in range(gs)}
