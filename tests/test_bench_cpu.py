"""bench.py host logic that needs no GPU: `python bench.py --gpus N` without a launcher re-executes itself under
torch.distributed.run (the driver's N = 1 form must also work at N > 1)."""
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("fx_bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_self_launch_command_and_return_code(monkeypatch):
    b = _bench()
    seen = {}

    class R:
        returncode = 7

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return R()

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "20", "--warmup", "5"])
    assert b._self_launch(8) == 7                              # the launcher's return code is bench.py's
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]      # arguments forwarded verbatim
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_main_self_launches_only_without_a_launcher(monkeypatch):
    b = _bench()
    calls = []
    monkeypatch.setattr(b, "_self_launch", lambda n: calls.append(n) or 0)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    try:
        b.main()
    except SystemExit as e:
        assert e.code == 0
    assert calls == [4]
    # under a launcher that started the wrong number of ranks: refuse loudly, do not self-launch again
    monkeypatch.setenv("WORLD_SIZE", "2")
    try:
        b.main()
        raise AssertionError("expected SystemExit")
    except SystemExit as e:
        assert "WORLD_SIZE=2" in str(e.code)
    assert calls == [4]
