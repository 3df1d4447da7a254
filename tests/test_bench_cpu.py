"""bench.py's host logic that needs no GPU: `--gpus N` without a rendezvous starts N ranks itself (the driver's command
form is `python bench.py --gpus N --steps K --warmup W`), sample statistics, code stamp."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _bench():
    import importlib
    return importlib.import_module("bench")


class _FakeProc:
    """stands in for the torch.distributed.run child: some noise on stdout, then rank 0's line"""

    def __init__(self, lines, rc=0):
        self.stdout, self._rc = iter(lines), rc

    def wait(self):
        return self._rc


def test_gpus_flag_starts_the_ranks(monkeypatch, capsys):
    bench = _bench()
    seen = {}

    def fake_popen(cmd, env=None, stdout=None, text=None):
        seen["cmd"], seen["env"] = cmd, env
        return _FakeProc(["[Gloo] Rank 0 is connected to 3 peer ranks.\n", '{"metric": "ICP matched-feature-pairs/sec", "n_gpus": 4}\n'])

    monkeypatch.setattr(subprocess, "Popen", fake_popen)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "20", "--warmup", "5"])
    assert bench.main() == 0
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    k = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[k + 1:] == ["--gpus", "4", "--steps", "20", "--warmup", "5"]  # the ranks get the same arguments
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # ONE JSON line on stdout, whatever else the children wrote there goes to stderr
    io = capsys.readouterr()
    assert io.out.strip() == '{"metric": "ICP matched-feature-pairs/sec", "n_gpus": 4}' and "Gloo" in io.err


def test_ranks_without_a_result_line_are_a_failure(monkeypatch):
    bench = _bench()
    monkeypatch.setattr(subprocess, "Popen", lambda *a, **k: _FakeProc(["noise\n"]))
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    assert bench.main() != 0


def test_a_rank_does_not_spawn_again(monkeypatch):
    """inside a rank (WORLD_SIZE set by torch.distributed.run) --gpus N must not start ranks of its own"""
    bench = _bench()
    called = []
    monkeypatch.setattr(subprocess, "Popen", lambda *a, **k: called.append(a) or _FakeProc([]))
    monkeypatch.setenv("WORLD_SIZE", "4")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--code-stamp"])
    assert bench.main() == 0 and not called


def test_the_parent_makes_no_gpu_call():
    """the spawning parent must not have initialised the GPU (not even imported torch) when it starts its children"""
    code = ("import sys, subprocess, os\n"
            "class P:\n"
            "    stdout = iter(['{\"metric\": 1}\\n'])\n"
            "    def wait(self): return 0\n"
            "sys.argv = ['bench.py', '--gpus', '2']; subprocess.Popen = lambda *a, **k: P(); os.environ.pop('WORLD_SIZE', None)\n"
            f"sys.path.insert(0, {ROOT!r}); import bench; rc = bench.main(); assert rc == 0\n"
            "assert 'torch' not in sys.modules, 'torch imported by the parent'\n")
    subprocess.check_call([sys.executable, "-c", code])


def test_sample_stats_and_code_stamp():
    bench = _bench()
    st = bench.sample_stats(np.array([10.0, 12.0, 8.0, 100.0, 9.0]) * 1e-3, 2)
    assert st["n"] == 5 and abs(st["median_ms"] - 5e-3) < 1e-12 and st["max_ms"] == 0.05 and st["min_ms"] == 0.004
    assert bench.sample_stats(np.zeros(0), 2) is None
    s = bench.code_stamp()
    assert len(s) == 16 and int(s, 16) >= 0
