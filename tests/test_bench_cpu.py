"""bench.py's own rank launcher (`python bench.py --gpus N` without torchrun), checked without a GPU: the command it
re-executes must be the one the driver uses for N > 1 (one rank per GPU over RCCL, rendezvous on 127.0.0.1) with the
caller's arguments passed through, and it must refuse a node with fewer GPUs than asked for."""
import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    """bench.py sets GPU_MAX_HW_QUEUES for its own process at import time: importing it into the test process must not leak
    that into os.environ (contexts created by later GPU tests of the same pytest run would all come up in the 8-stream
    mode and the default 4-stream path would go unexercised)."""
    saved = os.environ.get("GPU_MAX_HW_QUEUES")
    sys.path.insert(0, ROOT)
    try:
        return importlib.import_module("bench")
    finally:
        sys.path.pop(0)
        if saved is None:
            os.environ.pop("GPU_MAX_HW_QUEUES", None)
        else:
            os.environ["GPU_MAX_HW_QUEUES"] = saved


def test_importing_bench_does_not_change_the_environment_of_the_test_process():
    before = os.environ.get("GPU_MAX_HW_QUEUES")
    _bench()
    assert os.environ.get("GPU_MAX_HW_QUEUES") == before


def test_gpus_flag_re_executes_under_torch_distributed_run(monkeypatch):
    bench = _bench()
    import subprocess

    import torch

    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0

    monkeypatch.setattr(torch.cuda, "device_count", lambda: 4)
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7", "--warmup", "2"])
    monkeypatch.delenv("RANK", raising=False)
    bench.main()  # returns after the (faked) child launch: the parent measures nothing itself
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 0 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    script = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[script + 1:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"  # dmabuf IPC: RCCL across processes on this driver


def test_gpus_flag_refuses_a_node_with_fewer_gpus(monkeypatch):
    bench = _bench()
    import torch

    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    monkeypatch.delenv("RANK", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "only 1 GPU" in str(e.value)


def test_child_failure_is_the_parents_exit_code(monkeypatch):
    bench = _bench()
    import subprocess

    import torch

    monkeypatch.setattr(torch.cuda, "device_count", lambda: 2)
    monkeypatch.setattr(subprocess, "call", lambda cmd, env=None: 3)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    monkeypatch.delenv("RANK", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 3


def test_two_rank_line_carries_the_weak_aggregate_the_sharded_index_and_what_the_collectives_ran_on():
    """`bench.py --gpus 2` as ONE launch / one process group (here: --dry-run = gloo, cpu tensors, a pass-through stand-in for
    the codec, `value` null): rank 0's line must carry (i) the weak form with every rank's step time, (ii) `extra.sharded_c5` =
    ONE index partitioned over the ranks with per-rank codec ms, their spread, gather ms and bytes, the gather verified against
    the index, (iii) `rccl` = backend, world size, one device per rank.  The day a multi-GPU node exists the same command
    without --dry-run answers both scaling questions."""
    import json
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "1",
                        "--sharded-workload", "s1"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1  # one line, from rank 0
    d = json.loads(lines[0])
    assert d["dry_run"] is True and d["value"] is None  # a dry run measures nothing and says so
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 2
    assert len(d["per_rank"]["ms_per_step"]) == 2 and d["per_rank"]["spread_ms"] >= 0
    assert d["rccl"] == {"world": 2, "backend": "gloo", "devices": ["cpu", "cpu"], "rccl_version": None}
    sh = d["extra"]["sharded_c5"]
    assert sh["scaling"] == "strong" and sh["gather_verified"] is True
    pr = sh["per_rank"]
    for key in ("ids", "codec_ms", "gather_ms", "kernel_ms_encode", "kernel_ms_decode"):
        assert len(pr[key]) == 2, key
    assert sum(pr["ids"]) == 1_000_000 and abs(pr["ids"][0] - pr["ids"][1]) <= 52114  # LPT: within the longest list
    assert pr["codec_ms_spread"] >= 0 and sh["gather_lists"] == 16000 and sh["gather_bytes"] > 0
    # ROC's strong scaling is bounded by its longest list (one serial chain): sum of ids / (G x longest list), printed beside it
    assert sh["codec"] == "roc" and abs(sh["strong_bound"] - 1_000_000 / (2 * 52114)) < 1e-3
    # the same index through a codec without that chain: the form north_star's ">= 6x at 8 GPUs" can show on
    ef = d["extra"]["sharded_c5_ef"]
    assert ef["codec"] == "ef" and ef["scaling"] == "strong" and ef["gather_verified"] is True and ef["strong_bound"] is None
    assert sum(ef["per_rank"]["ids"]) == 1_000_000 and ef["gather_lists"] == 16000
    assert len(lines[0]) < 7500  # the driver keeps the last ~8000 characters of stdout
