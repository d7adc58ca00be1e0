"""CPU: the launcher half of bench.py -- `--gpus N` without a launcher starts N rank processes as a CHILD
(torch.distributed.run) before torch is imported; a launcher that started a different number of ranks is an error."""
import importlib.util
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
