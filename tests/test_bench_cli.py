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


def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", ROOT / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    return bench


def test_multi_gpu_default_is_the_north_star_metric_strong_scaling_of_the_4096_batch():
    """`bench.py --gpus N` (the driver's line) measures "a batch of 4096 instances at 1 / 2 / 4 / 8 MI355X": the 4096 are SPLIT over
    the ranks (512 per GPU at 8, contiguous shards that cover the batch exactly once), the metric string states the total; --weak
    keeps 4096 per GPU and says so."""
    import types
    bench = _bench_module()
    args = types.SimpleNamespace(batch=4096, total_batch=0, weak=False)
    for world in (1, 2, 4, 8):
        covered = []
        for rank in range(world):
            strong, total, B, first = bench.plan_batch(args, world, rank)
            assert strong and total == 4096 and B == 4096 // world
            covered += list(range(first, first + B))
        assert covered == list(range(4096))
        assert bench.metric_string(True, total, B, world, 100) == "MPC+WBC updates/sec (batch=4096, N=100, 12-DoF)"
    strong, total, B, first = bench.plan_batch(types.SimpleNamespace(batch=4096, total_batch=0, weak=True), 8, 3)
    assert (strong, total, B, first) == (False, 32768, 4096, 3 * 4096)
    assert "4096 per GPU x 8 GPUs = 32768" in bench.metric_string(False, total, B, 8, 100)
    strong, total, B, first = bench.plan_batch(types.SimpleNamespace(batch=4096, total_batch=1000, weak=False), 8, 7)
    assert strong and total == 1000 and first + B == 1000   # uneven split: the last shard ends the batch
    with pytest.raises(SystemExit):
        bench.plan_batch(types.SimpleNamespace(batch=4096, total_batch=96, weak=True), 2, 0)


def test_counter_file_is_tied_to_the_kernel_sources():
    """roofline.traffic comes from profiles/pmc_latest.json, collected in a separate rocprofv3 run: the file carries the sha256 of the
    kernel sources it was collected on (+ the commit), and bench.py drops the figure when the running tree's sources differ."""
    import json
    bench = _bench_module()
    fp = bench.source_fingerprint()
    assert len(fp) == 64 and fp == bench.source_fingerprint()
    pj = json.loads((ROOT / "profiles" / "pmc_latest.json").read_text())
    assert "source_sha256" in pj and "git_head" in pj and "tag" in pj


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
    assert j["n_gpus"] == 1 and j["scaling"] == "strong" and j["config"]["total_instances"] == 96 and j["config"]["batch_per_gpu"] == 96
    # the line says who took part: one rank in the RCCL all-reduce, one distinct device (N of each on N GPUs — what a SCALE run is audited on)
    assert j["rccl_ranks"] == 1 and isinstance(j["devices"], list) and len(j["devices"]) == 1 and j["devices"][0]
    assert j["metric"] == "MPC+WBC updates/sec (batch=96, N=24, 12-DoF)" and j["value_per_gpu"] == j["value"]
    assert j["gather"]["bytes_per_rank"] > 0 and j["gather"]["ms_per_step"] > 0
    assert j["roofline"]["frac"] > 0 and j["roofline"]["traffic"] is None   # (counter traffic belongs to the 4096 x 100 headline only)
    assert sum(j["solver_state"]["mpc_status_histogram_all_ranks"]) == 96


@pytest.mark.gpu
def test_bench_line_under_the_launcher_keeps_roofline_and_cpu_baseline():
    """rank 0 prints `roofline` and `cpu_baseline` at ANY world size (the CPU leg runs after the final barrier)."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", "29534", str(ROOT / "bench.py"), "--gpus", "1", "--batch", "64", "--nodes", "24", "--steps", "3",
                        "--warmup", "1", "--no-extras"], env=env, capture_output=True, text=True, timeout=600)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and lines, r.stdout[-2000:] + r.stderr[-2000:]
    j = json.loads(lines[-1])
    assert j["scaling"] == "strong" and j["config"]["total_instances"] == 64
    assert j["cpu_baseline"]["value"] > 0 and j["cpu_baseline"]["kind"] == "port" and j["roofline"]["kernel"] in ("k_lq", "k_ric_bwd", "k_ric_fwd")
    assert sum(j["solver_state"]["mpc_status_histogram_all_ranks"]) == 64
