"""The C++ host adapter (include/hunter_hip.hpp) mirrors the reference's operator interface
(WbcBase::update, MPC_MRT_Interface::{setCurrentObservation, advanceMpc, updatePolicy, evaluatePolicy}); these tests
build a small C++ program against it with g++ and drive it like LeggedController does."""
import subprocess
from pathlib import Path

import numpy as np
import pytest

from hunter_bipedal_control_amd import workload

ROOT = Path(__file__).resolve().parents[1]
PKG = ROOT / "hunter_bipedal_control_amd"
PARAMS_BIN = PKG / "data" / "hunter_params.bin"


def _build():
    lib = PKG / "libhunter_hip.so"
    if not lib.exists():
        pytest.skip("libhunter_hip.so not built (python __graft_entry__.py build)")
    out = ROOT / "tests" / "cpp" / "_build"
    out.mkdir(exist_ok=True)
    exe = out / "adapter_test"
    src = ROOT / "tests" / "cpp" / "adapter_test.cpp"
    newest = max(src.stat().st_mtime, (ROOT / "include" / "hunter_hip.hpp").stat().st_mtime, lib.stat().st_mtime)
    if not exe.exists() or exe.stat().st_mtime < newest:
        subprocess.check_call(["g++", "-std=c++14", "-O1", "-Wall", "-I", str(ROOT / "include"), str(src), "-L", str(PKG),
                               "-lhunter_hip", f"-Wl,-rpath,{PKG}", "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)])
    return exe


def test_params_blob_matches_packaged_json(params):
    """data/hunter_params.bin is the byte image of the structs abi.make_model / make_config build from the JSON."""
    import ctypes as C
    import struct
    from hunter_bipedal_control_amd import abi
    raw = PARAMS_BIN.read_bytes()
    magic, sm, sc, _ = struct.unpack("<4I", raw[:16])
    assert magic == abi.PARAMS_BLOB_MAGIC and sm == C.sizeof(abi.HbModel) and sc == C.sizeof(abi.HbConfig)
    assert raw[16:16 + sm] == bytes(abi.make_model(params)) and raw[16 + sm:] == bytes(abi.make_config(params))


def test_adapter_builds_and_fails_loudly_without_gpu():
    import torch
    exe = _build()
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: the no-GPU branch is covered on the CPU runner")
    r = subprocess.run([str(exe), str(PARAMS_BIN), "nogpu"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "status -4" in r.stdout


def _write_problem(path, refs, x0, rbd, t_now, nmax):
    B = x0.shape[0]
    with open(path, "wb") as f:
        np.array([B, nmax], dtype=np.int32).tofile(f)
        np.ascontiguousarray(refs["n_nodes"], dtype=np.int32).tofile(f)
        for key, dt in (("t", np.float64), ("mode", np.int32), ("x_ref", np.float64), ("swing", np.float64)):
            np.ascontiguousarray(refs[key], dtype=dt).tofile(f)
        for a in (x0, rbd, t_now):
            np.ascontiguousarray(a, dtype=np.float64).tofile(f)


@pytest.mark.gpu
@pytest.mark.parametrize("wbc_type", [0, 1])
def test_cpp_adapter_control_loop_matches_ctypes_path(params, tmp_path, wbc_type):
    """advanceMpc x2 -> updatePolicy/evaluatePolicy/WBC through the C++ adapter == the same calls through ctypes
    (both bind the same C ABI; the ctypes path is the one checked against the oracle in test_gpu_parity.py)."""
    from hunter_bipedal_control_amd.solver import HunterSolver
    exe = _build()
    B, N = 6, 40
    refs, x0, rbd, t_now = workload.trot_batch(params, B, n_intervals=N)
    nmax = refs["mode"].shape[1]
    prob, res = tmp_path / "problem.bin", tmp_path / "result.bin"
    _write_problem(prob, refs, x0, rbd, t_now, nmax)
    r = subprocess.run([str(exe), str(PARAMS_BIN), "run", str(prob), str(res), "2", str(wbc_type)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert f"ok: {2 * B} mpc solves" in r.stdout  # hb_stats counts instance solves
    raw = np.fromfile(res, dtype=np.float64)
    sizes = [B * 38, B * 22, B * 22, B, B, B * 38, B * (nmax + 1) * 22, B * nmax * 22]
    assert raw.size == sum(sizes)
    parts = np.split(raw, np.cumsum(sizes)[:-1])
    sol, xd, ud, mode, status, direct, xs, us = parts
    s = HunterSolver(params, batch=B, max_nodes=nmax, wbc_type=wbc_type)
    try:
        s.set_references(refs)
        s.reset(x0)
        s.mpc_solve(x0)
        s.mpc_solve(x0)
        s.publish()
        out = s.wbc_update(t_now, rbd)
        sol_direct, st_direct = s.wbc_update_direct(out["x_des"], out["u_des"], rbd, out["mode"])
        xg, ug = s.get_solution()
    finally:
        s.close()
    assert np.array_equal(xs.reshape(xg.shape), xg) and np.array_equal(us.reshape(ug.shape), ug)
    assert np.array_equal(xd.reshape(B, 22), out["x_des"]) and np.array_equal(ud.reshape(B, 22), out["u_des"])
    assert np.array_equal(mode.astype(np.int32), out["mode"]) and np.array_equal(status.astype(np.int32), out["status"])
    assert np.array_equal(sol.reshape(B, 38), out["sol"])
    assert np.array_equal(direct.reshape(B, 38), sol_direct)
