"""bench.py's own launcher (`python bench.py --gpus N` without torchrun): what it starts, what it relays, and its
second attempt with host-staged halos when the ranks fail or hang on RCCL.  No GPU: the child processes are faked."""
import argparse
import importlib.util
import os
import subprocess
import sys

import pytest

from conftest import ROOT

LINE = '{"metric": "cell-updates/sec on Add module, 64x64 DEM", "value": 1.0}'


@pytest.fixture()
def bench(monkeypatch):
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "3"])
    for k in ("WDPM_HALO", "WDPM_DIST_BACKEND"):
        monkeypatch.delenv(k, raising=False)
    return mod


class FakePopen:
    """scripted children: `script` is a list of (returncode | 'hang', stdout, stderr), one per launch"""
    script, calls = [], []

    def __init__(self, cmd, env=None, **kw):
        assert kw.get("start_new_session") is True          # a hung run is ended by its own process group id
        FakePopen.calls.append((cmd, env))
        self.rc, self.out, self.err = FakePopen.script[len(FakePopen.calls) - 1]
        self.pid, self.returncode, self.killed = 2 ** 22 + 12345, None, False     # above pid_max: no such group

    def communicate(self, timeout=None):
        if self.rc == "hang" and timeout is not None:
            raise subprocess.TimeoutExpired("x", timeout)
        self.returncode = -9 if self.rc == "hang" else self.rc
        return self.out, self.err


def run(bench, monkeypatch, capsys, script):
    FakePopen.script, FakePopen.calls = script, []
    monkeypatch.setattr(subprocess, "Popen", FakePopen)
    rc = bench.self_launch(argparse.Namespace(gpus=2))
    return rc, capsys.readouterr(), FakePopen.calls


def test_launcher_relays_one_line(bench, monkeypatch, capsys):
    rc, io, calls = run(bench, monkeypatch, capsys, [(0, "noise from a rank\n" + LINE + "\n", "")])
    assert rc == 0 and io.out.strip() == LINE and "noise from a rank" in io.err
    cmd, env = calls[0]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=2" in cmd and "--standalone" in cmd
    assert cmd[-4:] == ["--gpus", "2", "--steps", "3"] and env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_launcher_retries_a_lost_port_race(bench, monkeypatch, capsys):
    rc, io, calls = run(bench, monkeypatch, capsys, [(1, "", "RuntimeError: EADDRINUSE\n"), (0, LINE + "\n", "")])
    assert rc == 0 and io.out.strip() == LINE and len(calls) == 2
    assert "--master-port" in calls[1][0] and "WDPM_HALO" not in calls[1][1]          # same transport, another port


def test_launcher_falls_back_to_host_halos_when_ranks_fail(bench, monkeypatch, capsys):
    rc, io, calls = run(bench, monkeypatch, capsys, [(1, "", "ncclCommInitRank: unhandled system error\n"), (0, LINE + "\n", "")])
    assert rc == 0 and io.out.strip().startswith(LINE[:-1]) and len(calls) == 2
    assert calls[1][1]["WDPM_HALO"] == "host" and calls[1][1]["WDPM_DIST_BACKEND"] == "gloo"
    assert "host-staged halos" in io.err and "unhandled system error" in io.err
    import json
    d = json.loads(io.out)
    assert d["degraded"] is True and "exit status 1" in d["first_attempt"]     # a second run is never passed off as the first


def test_launcher_falls_back_when_ranks_hang(bench, monkeypatch, capsys):
    monkeypatch.setenv("WDPM_BENCH_RANKS_TIMEOUT", "0.01")
    rc, io, calls = run(bench, monkeypatch, capsys, [("hang", "", ""), (0, LINE + "\n", "")])
    import json
    assert rc == 0 and json.loads(io.out)["degraded"] is True and calls[1][1]["WDPM_HALO"] == "host"
    assert "no result within" in io.err and "budget" in io.err


def test_launcher_keeps_both_attempts_inside_the_time_budget(bench, monkeypatch, capsys):
    """the driver ends bench.py after 600 s: the first attempt gets at most 200 s by default, the second what is left of 560"""
    seen, clock = [], [1000.0]
    monkeypatch.setattr(bench.time, "monotonic", lambda: clock[0])

    class Timed(FakePopen):
        def communicate(self, timeout=None):
            seen.append(timeout)
            if self.rc == "hang" and timeout is not None:
                clock[0] += timeout                         # a hung run uses up its whole limit
            return super().communicate(timeout)
    FakePopen.script, FakePopen.calls = [("hang", "", ""), (0, LINE + "\n", "")], []
    monkeypatch.setattr(subprocess, "Popen", Timed)
    assert bench.self_launch(argparse.Namespace(gpus=2)) == 0
    limits = [t for t in seen if t is not None]
    assert limits[0] <= 200.0 and limits[0] + limits[1] <= 560.0
    monkeypatch.setenv("WDPM_BENCH_BUDGET_S", "15")           # no room for a second attempt: none is made
    FakePopen.script, FakePopen.calls = [("hang", "", ""), (0, LINE + "\n", "")], []
    assert bench.self_launch(argparse.Namespace(gpus=2)) == 124 and len(FakePopen.calls) == 1


def test_launcher_gives_up_after_the_second_transport(bench, monkeypatch, capsys):
    rc, io, calls = run(bench, monkeypatch, capsys, [(1, "", "boom\n"), (7, "", "boom again\n")])
    assert rc == 7 and io.out.strip() == "" and len(calls) == 2


def test_launcher_does_not_fall_back_from_a_transport_the_user_chose(bench, monkeypatch, capsys):
    monkeypatch.setenv("WDPM_HALO", "host")
    rc, io, calls = run(bench, monkeypatch, capsys, [(5, "", "boom\n")])
    assert rc == 5 and len(calls) == 1
