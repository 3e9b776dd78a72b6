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
    a = _args(["--gpus", "4", "--steps", "7", "--warmup", "2", "--frames", "4096", "--window", "rect", "--no-secondary",
               "--first-frame", "3145728"])
    cmd = bench.child_command(a, 29511)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29511"
    script = cmd.index(os.path.abspath(bench.__file__))
    tail = cmd[script + 1:]
    for flag, val in (("--gpus", "4"), ("--steps", "7"), ("--warmup", "2"), ("--frames", "4096"), ("--window", "rect"),
                      ("--first-frame", "3145728")):                       # the ranks' frame ranges start where the parent was told
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


def test_self_launch_parent_never_imports_torch():
    """The parent of `python bench.py --gpus N` must stay free of any GPU runtime: checked in a fresh interpreter
    (this process may have torch loaded by other tests)."""
    import subprocess
    code = (
        "import sys, subprocess, bench\n"
        "subprocess.call = lambda cmd, env=None: 0\n"
        "bench.cpu_baseline = lambda w, s: {'value': 1.0}\n"
        "bench.cpu_baseline_all_cores = lambda w, s: {'value': 2.0}\n"
        "sys.argv = ['bench.py', '--gpus', '2']\n"
        "try:\n"
        "    bench.main()\n"
        "except SystemExit as e:\n"
        "    assert e.code == 0, e.code\n"
        "assert 'torch' not in sys.modules, 'the self-launch parent imported torch'\n"
        "print('parent clean')\n")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, "-c", code], cwd=os.path.dirname(os.path.abspath(bench.__file__)), env=env,
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "parent clean" in out.stdout, out.stderr[-2000:]


def test_main_self_launches_only_without_a_launcher(monkeypatch):
    called = []
    monkeypatch.setattr(bench, "self_launch", lambda a: (_ for _ in ()).throw(SystemExit(called.append(a.gpus) or 0)))
    monkeypatch.delenv("RANK", raising=False)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8"])
    with pytest.raises(SystemExit):
        bench.main()
    assert called == [8]


# ---- the nccl-or-gloo choice is collective: world 2 over gloo, the nccl stages faked -------------------

def _negotiate_worker(rank, world, port, scenario, q):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist
    import bench as b
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    calls = []

    def form():
        calls.append("form")
        if scenario == "env_one_rank":
            os.environ["SDRK_BENCH_FAIL_NCCL"] = "rank:1"
        if scenario == "env_all":
            os.environ["SDRK_BENCH_FAIL_NCCL"] = "all"
        if b._injected_nccl_failure(rank) or scenario == "none_form" or (scenario == "mixed_form" and rank == 1):
            raise RuntimeError("no communicator here")
        return "fake-nccl-group"

    def prove():
        calls.append("prove")
        if scenario == "none_prove" or (scenario == "mixed_prove" and rank == 0):
            raise RuntimeError("all-reduce of ones gave nonsense")
        return "fake-nccl-group"

    try:
        try:
            res = b.negotiate_backend(dist, world, rank, dist.group.WORLD, [form, prove])
            q.put((rank, "ok", res[0], res[1] if res[0] == "nccl" else None, res[2], calls))
        except SystemExit as e:
            q.put((rank, "exit", str(e.code), None, None, calls))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("scenario,expect", [
    ("all_ok", "nccl"), ("none_form", "gloo"), ("none_prove", "gloo"), ("env_all", "gloo"),
    ("mixed_form", "exit"), ("mixed_prove", "exit"), ("env_one_rank", "exit")])
def test_backend_choice_is_one_decision_for_all_ranks(scenario, expect):
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_negotiate_worker, args=(r, 2, port, scenario, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    kinds = {g[1] if g[1] == "exit" else g[2] for g in got}
    assert kinds == {expect}, got                      # every rank reached the SAME decision
    if expect == "nccl":
        assert all(g[3] == "fake-nccl-group" and g[4] is None and g[5] == ["form", "prove"] for g in got)
    elif expect == "gloo":
        assert all("did not come up on any rank" in g[4] for g in got)
        if scenario in ("none_form", "env_all"):
            assert all(g[5] == ["form"] for g in got)      # nobody went on to the all-reduce
    else:
        assert all("ranks only" in g[2] and "every rank gives up" in g[2] for g in got)
        if scenario in ("mixed_form", "env_one_rank"):
            assert all(g[5] == ["form"] for g in got)      # the healthy rank was NOT left alone inside the collective
