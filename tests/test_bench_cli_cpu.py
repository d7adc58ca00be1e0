"""CPU: the launcher half of bench.py -- `--gpus N` without a launcher starts N rank processes as a CHILD
(torch.distributed.run) before torch is imported; a launcher that started a different number of ranks is an error."""
import importlib.util
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench_module():
    spec = importlib.util.spec_from_file_location("_bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_world_size_mismatch_is_an_error():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                       text=True, timeout=60)
    assert r.returncode == 2 and "WORLD_SIZE=3" in r.stderr and r.stdout == ""


def test_bare_gpus_n_starts_n_ranks_as_a_child(monkeypatch):
    bench = _bench_module()
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    monkeypatch.setattr(bench.subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    before = set(sys.modules)
    try:
        bench.main()
        raise AssertionError("main() must exit with the children's code")
    except SystemExit as e:
        assert e.code == 7
    assert "torch" not in (set(sys.modules) - before)            # the launcher never imports torch / touches the GPU
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--local-addr") + 1] == "127.0.0.1"
    # no pre-picked port: the c10d store binds a free one itself
    assert cmd[cmd.index("--rdzv-endpoint") + 1] == "127.0.0.1:0" and "--master-port" not in cmd
    assert cmd[-4:] == ["--gpus", "4", "--steps", "3"] and cmd[-5].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_one_gpu_needs_no_launcher(monkeypatch):
    bench = _bench_module()
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = bench.parse()
    assert a.gpus == 1 and a.scale_10m == "auto" and a.scale_10m_rows == 10000000


def _selftest(extra_env, timeout=120):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update({"ISEHR_BENCH_SELFTEST": "1", "OMP_NUM_THREADS": "1"}, **extra_env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True,
                       timeout=timeout, cwd=ROOT)
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    return r, lines


def test_headline_goes_out_early_and_complete_at_the_end():
    """Two ranks over gloo (the rank-process plumbing of bench.py with a stand-in workload): rank 0 writes the line as soon as
    the headline exists (`complete: false`) and again at the end (`complete: true`); the driver parses the last one."""
    r, lines = _selftest({})
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 2 and lines[0]["complete"] is False and lines[-1]["complete"] is True
    assert lines[-1]["value"] == 2.0 and lines[-1]["config"]["rccl_ranks"] == 2 and lines[-1]["row_shard_1xN"] == {"value": 1.0}


def test_a_rank_that_hangs_in_a_secondary_block_costs_neither_the_headline_nor_ten_minutes():
    """rank 1 sleeps inside `row_shard_1xN`.  Rank 0 is stuck in the block's collective; its watchdog writes
    the line with {"error": "timeout"} for the block when the deadline passes and every rank exits non-zero -- the headline
    line is on stdout, parsable, and the launcher returns a non-zero code well inside the driver's limit."""
    import time
    t0 = time.time()
    r, lines = _selftest({"ISEHR_BENCH_TEST_HANG": "row_shard_1xN:1", "ISEHR_BENCH_DEADLINE_S": "8"}, timeout=150)
    took = time.time() - t0
    assert r.returncode != 0 and took < 90, (r.returncode, took)
    assert lines and lines[0]["value"] == 2.0 and lines[0]["complete"] is False
    last = lines[-1]
    assert last["value"] == 2.0 and last["complete"] is False and last["row_shard_1xN"] == {"error": "timeout", "deadline_s": 8}


def test_a_rank_that_dies_makes_the_job_exit_non_zero(tmp_path):
    """A rank killed inside a block (here: rank 1 exits by its own watchdog first, with a shorter deadline than rank 0's
    process-group timeout) never leaves the job hanging: the launcher sees the dead child and ends the others."""
    r, lines = _selftest({"ISEHR_BENCH_TEST_HANG": "row_shard_1xN:0", "ISEHR_BENCH_DEADLINE_S": "6"}, timeout=150)
    assert r.returncode != 0
    assert lines and lines[-1]["row_shard_1xN"]["error"] == "timeout"      # rank 0 itself hung: its own watchdog reports it
