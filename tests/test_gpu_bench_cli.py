"""GPU: bench.py as the driver starts it.  `python bench.py --gpus 2` WITHOUT a launcher must itself start two rank
processes (here they share the test box's one GPU and talk over gloo, because RCCL refuses two ranks on one device) and
print one JSON line whose n_gpus and communicator size are 2."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, env_extra=None, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR",
                                                           "MASTER_PORT")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env, capture_output=True, text=True,
                       timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    # stdout carries the JSON line and nothing else: once as soon as the headline is measured (complete: false) and once more,
    # complete, at the end -- possibly refreshed in between; the driver parses the LAST line
    recs = [json.loads(ln) for ln in lines]
    assert len(recs) >= 2 and recs[0]["complete"] is False and recs[-1]["complete"] is True, lines
    # the headline never changes once it is out -- except that the one-GPU default line takes the faster of its two tail modes
    # (the same K steps, measured behind the first line: `value_synchronous` / `value_pipelined`)
    assert all(rc["value"] in (recs[0]["value"], rc.get("value_synchronous")) for rc in recs)
    assert recs[-1].get("value_pipelined", recs[0]["value"]) == recs[0]["value"]
    return recs[-1]


def test_bare_gpus_2_runs_two_ranks():
    out = _bench(["--gpus", "2", "--rows", "200000", "--steps", "4", "--warmup", "2", "--no-cpu-baseline"],
                 {"ISEHR_DIST_BACKEND": "gloo", "ISEHR_SHARE_GPU": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    assert out["n_gpus"] == 2 and out["config"]["rccl_ranks"] == 2 and out["config"]["comm_backend"] == "gloo"
    assert out["config"]["rccl_ranks_equal_n_gpus"] is True and len(out["config"]["rank_devices"]) == 2
    assert out["config"]["rank_devices_distinct"] is False             # (rehearsal: both ranks share the box's one GPU)
    assert out["config"]["gallery_rows"] == 200000 and out["steps"] == 4 and out["value"] > 0
    assert "scale_10m" not in out                                     # --rows override: the secondary block is off


def test_bare_gpus_2_row_shards_with_secondary_block():
    """Row shards (1 x 2: the protocol with both all-gathers) and the secondary block at a reduced size (2 x 300 k rows of
    the bf16 image on the shared GPU)."""
    out = _bench(["--gpus", "2", "--rows", "200000", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--layout", "1x2",
                  "--scale-10m", "on", "--scale-10m-rows", "600000", "--scale-10m-steps", "3"],
                 {"ISEHR_DIST_BACKEND": "gloo", "ISEHR_SHARE_GPU": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    assert out["n_gpus"] == 2 and out["config"]["rccl_ranks"] == 2
    assert out["config"]["collectives"] == "synchronous per batch"
    s = out["scale_10m"]
    assert s["gallery_rows"] == 600000 and s["rows_per_rank"] == 300000 and s["image"] == "bf16" and s["n_gpus"] == 2
    assert s["value"] > 0 and s["parallelism"] == "row-shard x2"


def test_one_gpu_line_has_the_contract_fields():
    out = _bench(["--rows", "150000", "--steps", "5", "--warmup", "2", "--cpu-sample-rows", "8192"])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in out, key
    assert out["n_gpus"] == 1 and out["config"]["rccl_ranks"] == 1
    assert out["config"]["ingest_s"] > 0 and out["config"]["ingest_first_s"] > 0
    # create + one batch + destroy, measured: never less than the ingest alone, and three launch times of the ingest kernel
    assert out["config"]["stateless_call_s"] > out["config"]["ingest_kernel_s"] > 0
    assert len(out["config"]["ingest_launch_ms"]) == 3 and out["config"]["queries_per_s_of_one_stateless_call"] > 0
    cb = out["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["cpu_model"]
    pts = {(p["queries"], p["threads"] == 1) for p in cb["blas"]["points"]}
    assert {(1024, True), (70, True), (1, True)} <= pts and any(q == 70 and not one for q, one in pts)
    assert cb["value_blas"] == cb["blas"]["value"] and "faiss" in cb and out["vs_cpu_baseline"]["blas_full_width"] > 1
    assert out["value_synchronous"] == out["synchronous"]["value"]
    r = out["roofline"]
    assert r["bound"] == "mfma" and 0 < r["frac"] < 1 and r["launches"] == 5


def test_default_shape_line_carries_the_secondary_blocks():
    """The default workload (1 005 994 rows) at a reduced cost.  The headline runs in the deferred-tail mode; the line then
    carries `synchronous` -- the same steps with the tail on the caller's stream, whose answer must equal the pipelined one
    and on whose undisturbed launches `roofline` is quoted --, `scale_10m` (here on a smaller row count) and `map`: mAP
    (E / M / H) of the HIP ranks on the planted rOxford5k- / +rParis6k-sized datasets."""
    out = _bench(["--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--scale-10m", "on", "--scale-10m-rows", "400000",
                  "--scale-10m-steps", "2", "--extra-blocks", "off"])
    assert out["config"]["gallery_rows"] == 1005994 and out["config"]["tail"].startswith("deferred")
    # (round 6: the headline is the faster of the two tail modes of the box; both are in the line)
    assert out["config"]["value_mode"].startswith(("pipelined", "synchronous (faster than the deferred tail"))
    assert out["value"] == max(out["value_pipelined"], out["value_synchronous"])
    assert "mi_knn_dense64_search" in out["config"]["score_check"]
    d = out["synchronous"]
    assert d["equals_pipelined_answer"] is True and d["flagged_batches"] == 0
    assert d["value"] > 0 and 0 < d["kernel_share_of_step"] < 1 and d["steps"] == 6
    r = out["roofline"]
    assert r["launches"] == 6 and "synchronous" in r["measured_on"] and r["pipelined"]["launches"] == 6
    assert r["pipelined"]["kernel_share_of_step"] > r["kernel_share_of_step"]        # the tail left the step
    s = out["scale_10m"]
    assert "error" not in s and s["gallery_rows"] == 400000 and s["image"] == "bf16"
    assert "mi_knn_dense64_search" in s["score_check"]
    m = out["map"]
    assert [b["gallery_rows"] for b in m] == [4993, 11315]
    for b in m:
        assert 0 < b["map"]["H"] <= b["map"]["M"] <= 1 and 0 < b["map"]["E"] <= 1 and b["time_per_query_s"] > 0


def test_default_line_gives_every_baseline_config_a_number():
    """the default one-GPU line also carries `q70` and `q1` (HBM-bound small batches on the resident
    gallery, fraction of 8 TB/s), `aqe_rparis_1m` (configs[4]: alpha-QE at 1 007 323 rows, 1024 and 70 queries), `qge_small`
    (the diffusion branch at rOxford5k size) and `dropin`: matching_HIP(K, vecs.T, qvecs.T[, dataset]) on a HOST [2048, 1 005 994]
    float32 array -- the call src/test_rOP1m.py:155-159 makes -- stateless, stateful first / cached / from the file, with the
    prepared-gallery file written BEHIND the first call, and the float64 case of src/online.py:96."""
    out = _bench(["--steps", "10", "--warmup", "2", "--scale-10m", "off", "--cpu-sample-rows", "8192"], timeout=900)
    for name in ("q70", "q1"):
        b = out[name]
        assert b["roofline"]["bound"] == "hbm" and 0.3 < b["roofline"]["frac"] < 1 and b["flagged_batches"] == 0
        assert b["value"] > 0 and b["roofline"]["kernel"] == "stream_select_kernel"
    o = out["online"]
    assert o["online_query_ms"] > 0 and o["gallery_rows"] == 1005994 and "one D2H" in o["stages"]
    # round 6: 64 concurrent client threads coalesced into a few chains, same answers as their sequential calls
    oc = o["online_concurrent"]
    assert oc["equals_sequential_answers"] is True and oc["coalesced"]["mean_requests_per_chain"] > 4
    assert o["online_concurrent_qps"] > 4 * oc["sequential_calls_same_threads"]["value"]
    # the library's worker (mi_online_*) serves the figure; the Python worker and the library's own client threads sit beside it
    assert oc["coalesced"]["worker"].startswith("library") and oc["coalesced_python_worker"]["worker"] == "python thread"
    nc = oc["coalesced_native_clients"]
    assert nc["equals_sequential_answers"] is True and nc["mean_requests_per_chain"] > 16
    assert o["online_concurrent_native_clients_qps"] == nc["value"] > oc["coalesced_python_worker"]["value"]
    # round 6: the batch-size staircase, the headline's shape on structured data, learned whitening
    qs = out["qsweep"]
    assert [p["queries"] for p in qs["points"]] == [128, 129, 192, 256, 257, 384, 512, 768, 1024]
    assert all(p["flagged_batches"] == 0 and 0.2 < p["frac_at_launch"] < 1 for p in qs["points"])
    hd = out["hard_data"]
    assert set(hd["cases"]) == {"nonneg", "clustered", "near_duplicates"}
    for rec in hd["cases"].values():
        for q in ("q1024", "q70", "q1"):
            assert rec[q]["value"] > 0 and rec[q]["score_check"]["ids_equal_dense_f64"] is True
            assert rec[q]["flagged"] is False                          # no case leaves the primary path (spill path, hashed sample)
    assert hd["min_vs_gaussian_headline"] > 0.5
    w = out["whiten"]
    assert w["dims_2048"]["frac_of_f64_matrix_peak"] > 0.4 and w["dims_2048"]["max_abs_diff_vs_float64_numpy_16_rows"] < 1e-12
    assert w["dims_2048_from_DxN"]["frac_of_f64_matrix_peak"] > 0.3 and w["dims_2048_from_DxN"]["max_abs_diff_vs_float64_numpy_16_rows"] < 1e-12
    assert w["into_gallery"]["rows_per_s"] > 1e6 and w["into_gallery"]["max_abs_diff_of_stored_f32_rows"] < 2e-7
    a = out["aqe_rparis_1m"]
    assert a["gallery_rows"] == 1007323 and a["q1024"]["value"] > 0 and a["q70"]["value"] > 0
    assert a["q1024"]["flagged_batches"] == 0 and "dense float64" in a["q70"]["score_check"]
    g = out["qge_small"]
    assert g["gallery_rows"] == 4993 and g["offline_diffusion_s"] > 0 and g["online_value"] > 0
    assert g["max_abs_map_difference_alpha_qe"] <= 1e-6 and "_ranks_aqe" not in g
    assert 0 < g["map"]["diffusion"]["M"] <= 1
    d = out["dropin"]
    assert d["device_ingest_both_layouts"]["dn_over_row_major"] < 1.5
    assert d["same_answers"] is True and d["bytes"] == 2048 * 1005994 * 4 and d["pinned_h2d_GBps"] > 5
    c = d["calls"]
    assert c["dataset_first_call"]["gallery"]["source"] == "built" and c["dataset_cached"]["gallery"]["source"] == "cached"
    assert c["dataset_from_file"]["gallery"]["source"] == "file"
    assert c["dataset_first_call"]["file_written_behind_the_call"]["behind_the_call"] is True
    # the 12 GB file no longer sits in the caller's timer: the first stateful call costs about what the stateless one does
    assert c["dataset_first_call"]["wall_s"] < 2.0 * c["stateless"]["wall_s"] + 0.2
    assert c["dataset_cached"]["time_per_query_s"] < 0.1 * c["stateless"]["time_per_query_s"]
    assert d["float64_online_case"]["equals_f32_answer_of_the_same_rows"] is True
    assert out["cpu_baseline"]["value_blas"] > out["cpu_baseline"]["value"]


def test_explicit_synchronous_headline():
    out = _bench(["--rows", "150000", "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--async-tail", "0"])
    assert out["config"]["tail"] == "same stream" and out["config"]["value_mode"] == "synchronous"
    assert "synchronous" not in out and out["roofline"]["launches"] == 4


def test_bare_gpus_2_line_answers_layout_overlap_and_collective_cost():
    """what ONE hardware run at N > 1 must be able to answer.  Two ranks (sharing the test box's GPU over
    gloo), reduced sizes: the auto-layout headline, the `row_shard_1xN` block -- north_star's partition, synchronous AND with
    pipelined collectives, with the per-stage timings of the protocol -- and `scale_10m` with both as well."""
    out = _bench(["--gpus", "2", "--rows", "200000", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                  "--multi-gpu-blocks", "on", "--scale-10m", "on", "--scale-10m-rows", "600000", "--scale-10m-steps", "3"],
                 {"ISEHR_DIST_BACKEND": "gloo", "ISEHR_SHARE_GPU": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    assert out["n_gpus"] == 2 and out["config"]["rccl_ranks"] == 2
    assert out["config"]["collectives"] is None                         # auto layout at N = 2: two query groups, no exchange
    mg = out["multi_gpu"]                                               # round 6: what the first hardware run is read by
    assert mg["layout"] == "2x1" and mg["layout_requested"] == "auto" and mg["rank_step_ms"] > 0 and "expectation" in mg
    for blk in (out["row_shard_1xN"], out["scale_10m"]):
        assert blk["value"] > 0 and blk["pipelined_collectives"]["value"] > 0
        assert blk["pipelined_collectives"]["equals_synchronous_answer"] is True
        ph = blk["protocol_phases_ms_max_over_ranks"]
        assert set(ph) == {"phase1", "allgather1", "kth", "phase2", "allgather2", "merge", "total"}
        assert ph["total"] >= ph["phase1"] > 0 and ph["allgather1"] >= 0 and ph["allgather2"] >= 0
        assert "mi_knn_dense64_search" in blk["score_check"]
    assert out["row_shard_1xN"]["parallelism"].startswith("row-shard x2")
    br = out["batch_replicas"]                                         # whole-gallery replicas, the batches dealt to the ranks
    assert br["value"] > 0 and br["steps"] == 3 and br["steps_of_rank_0"] == 2 and "mi_knn_dense64_search" in br["score_check"]


def test_a_hang_in_the_pipelined_collectives_cannot_cost_the_synchronous_headline():
    """The asynchronous all-gathers of search_stream have never run on real RCCL before the driver's multi-GPU run: they are
    timed AFTER the headline line is out and under a deadline of their own.  Here rank 1 hangs inside that block (two ranks
    on the shared GPU over gloo): the headline line is on stdout, parsable, the block says timeout, the job ends non-zero
    within the deadline."""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update({"ISEHR_DIST_BACKEND": "gloo", "ISEHR_SHARE_GPU": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0",
                "ISEHR_BENCH_TEST_HANG": "pipelined_collectives:1", "ISEHR_BENCH_DEADLINE_S": "25"})
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rows", "200000", "--steps", "4",
                        "--warmup", "2", "--no-cpu-baseline", "--layout", "1x2"], env=env, capture_output=True, text=True,
                       timeout=400, cwd=ROOT)
    took = time.time() - t0
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert r.returncode != 0 and took < 300, (r.returncode, took, r.stderr[-2000:])
    assert lines and lines[0]["value"] > 0 and lines[0]["n_gpus"] == 2 and lines[0]["complete"] is False
    assert lines[-1]["value"] == lines[0]["value"] and lines[-1]["pipelined_collectives"] == {"error": "timeout", "deadline_s": 25}
