"""bench.py's rank contract (the driver's launch line): `--gpus N` is the number of ranks.  On a box without
N GPUs it must refuse with a message — never print an n_gpus = 1 line for a request of N — and under a launcher whose
WORLD_SIZE disagrees with N it must refuse as well."""
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def _run(extra_env, *argv):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(extra_env)
    return subprocess.run([sys.executable, str(ROOT / "bench.py"), *argv], env=env, capture_output=True, text=True, timeout=300)


def test_bare_multi_gpu_request_without_gpus_refuses():
    r = _run({}, "--gpus", "2", "--steps", "1", "--warmup", "0")
    assert r.returncode != 0
    assert "--gpus 2 needs 2 MI355X" in r.stderr and "found 0" in r.stderr
    assert '"n_gpus"' not in r.stdout


def test_world_size_mismatch_refuses():
    r = _run({"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"}, "--gpus", "2")
    assert r.returncode != 0 and "WORLD_SIZE=4" in r.stderr
    assert '"n_gpus"' not in r.stdout


def test_launcher_command_line(monkeypatch):
    """With enough devices the bare process replaces itself by torch.distributed.run with one rank per GPU on 127.0.0.1."""
    import importlib.util
    import types
    spec = importlib.util.spec_from_file_location("bench_under_test", ROOT / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    import torch
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    seen = {}

    def fake_exec(file, argv, env):
        seen["argv"], seen["env"] = argv, env
        raise SystemExit(0)

    monkeypatch.setattr(os, "execvpe", fake_exec)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7"])
    try:
        bench.launch_ranks_if_needed(types.SimpleNamespace(gpus=4))
    except SystemExit:
        pass
    a = seen["argv"]
    assert a[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in a and "--nnodes=1" in a
    assert a[a.index("--master-addr") + 1] == "127.0.0.1"
    assert a[-4:] == ["--gpus", "4", "--steps", "7"] and a[-5].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


import pytest


@pytest.mark.gpu
def test_bench_under_the_launcher_goes_through_rccl_even_with_one_rank():
    """The driver's multi-GPU launch line with ONE rank on the one GPU of this box: process group on the nccl (= RCCL) backend,
    barrier, MAX / SUM all-reduces of the timing and the status histograms on device tensors, and the optional all-gather of the
    outputs — the code the 2 / 4 / 8-rank runs execute, minus the peers."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", "29533", str(ROOT / "bench.py"), "--gpus", "1", "--total-batch", "96", "--nodes", "24", "--steps", "4",
                        "--warmup", "1", "--random-cmd", "--gather", "--no-cpu-baseline", "--no-extras"], env=env, capture_output=True, text=True,
                       timeout=600)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and lines, r.stdout[-2000:] + r.stderr[-2000:]
    j = json.loads(lines[-1])
    assert j["n_gpus"] == 1 and j["scaling"] == "strong" and j["config"]["total_instances"] == 96
    assert j["gather"]["bytes_per_rank"] > 0 and j["gather"]["ms_per_step"] > 0
    assert sum(j["solver_state"]["mpc_status_histogram_all_ranks"]) == 96
