"""The ros_control plugin legged/HipLeggedController (adapters/ros_control) EXECUTED: instantiated by its registered plugin name,
init -> starting -> update x N against the batched plant stub behind mock HybridJoint / IMU / contact handles
(tests/cpp/plugin_test.cpp over the mock ros_control layer of adapters/ros_control/test_shims) — LeggedController.cpp:41-135
(init), :112-135 (starting), :137-278 (update), :396-421 (MPC thread), legged_controllers_plugins.xml:3-8.
CPU: the harness builds and init() returns false, loudly, without a GPU.  -m gpu: the robot stands, then trots on /cmd_vel."""
import os
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
PKG = ROOT / "hunter_bipedal_control_amd"
PARAMS_BIN = PKG / "data" / "hunter_params.bin"


def _build():
    lib = PKG / "libhunter_hip.so"
    if not lib.exists():
        pytest.skip("libhunter_hip.so not built (python __graft_entry__.py build)")
    out = ROOT / "tests" / "cpp" / "_build"
    out.mkdir(exist_ok=True)
    exe = out / "plugin_test"
    src = ROOT / "tests" / "cpp" / "plugin_test.cpp"
    deps = [src, lib, ROOT / "adapters/ros_control/src/HipLeggedController.cpp", ROOT / "adapters/ros_control/include/hunter_hip_controllers/HipLeggedController.h",
            ROOT / "include/hunter_hip.hpp", ROOT / "include/hunter_ingest.hpp"]
    deps += list((ROOT / "adapters/ros_control/test_shims").rglob("*.h*"))
    import fcntl
    with open(out / "plugin_test.lock", "w") as lock:   # (pytest-xdist workers may want the same harness at the same time)
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not exe.exists() or exe.stat().st_mtime < max(d.stat().st_mtime for d in deps):
            tmp = out / f"plugin_test.{os.getpid()}"
            subprocess.check_call(["g++", "-std=c++14", "-O1", "-Wall", "-I", str(ROOT / "include"), "-I", str(ROOT / "adapters/ros_control/test_shims"),
                                   "-I", str(ROOT / "adapters/ros_control/include"), str(src), "-L", str(PKG), "-lhunter_hip", f"-Wl,-rpath,{PKG}",
                                   "-Wl,-rpath,/opt/rocm/lib", "-pthread", "-o", str(tmp)])
            os.replace(tmp, exe)
    return exe


def _run(exe, *args, timeout=600):
    r = subprocess.run([str(exe), *map(str, args)], capture_output=True, text=True, timeout=timeout)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
    assert r.returncode == 0 and line, r.stdout + r.stderr
    tok = line[-1].split()[1:]
    return dict(zip(tok[0::2], tok[1::2])), r.stderr


def test_plugin_builds_and_init_fails_loudly_without_a_gpu():
    import torch
    exe = _build()
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: the no-GPU branch is covered on the CPU runner")
    res, err = _run(exe, PARAMS_BIN, "lockstep", 0.1)
    assert res == {"init_failed": "1"} and "no HIP device visible" in err


@pytest.mark.gpu
@pytest.mark.parametrize("mode,seconds", [("lockstep", 3.0), ("threaded", 1.0)])
def test_plugin_runs_init_starting_update_against_the_plant(mode, seconds):
    """lockstep: 1 s standing, then /cmd_vel 0.3 m/s through the subscribed topic: walkGait switches to trot, the robot walks
    forward, upright, within the torque limits.  threaded: the plugin's own MPC thread (wall-clock cadence) holds the robot up."""
    exe = _build()
    res, err = _run(exe, PARAMS_BIN, mode, seconds)
    assert res.get("ok") == "1", (res, err)
    assert res["finite"] == "1" and float(res["max_tau"]) <= 60.0 + 1e-6
    assert 0.60 < float(res["min_h"]) and float(res["max_h"]) < 0.66, res
    assert float(res["max_tilt"]) < 0.08, res
    if mode == "lockstep":
        assert float(res["dx_walk"]) > 0.25 and 0.15 < float(res["speed"]) < 0.45, res     # ~2 s at a commanded 0.3 m/s
        assert int(res["modes_seen"]) & 0b0110, res                                          # single-support modes 1 and 2 were planned


@pytest.mark.gpu
def test_plugin_control_plane_hooks():
    """The hooks around the hot path (LeggedController.cpp:433-447 dynamic_reconfigure gains, :460-465 resetMPC, :474 / :496-510
    /reset_estimation, :277 the observation publisher), driven through the mock ROS layer while the robot stands: new gains show in
    the next joint command, a solver cold start and an observation reset in the middle of the run leave the robot standing, and
    every control tick publishes its observation (time, 22 + 22 float32 values)."""
    exe = _build()
    res, err = _run(exe, PARAMS_BIN, "hooks", 1.2)
    assert res.get("ok") == "1" and res["finite"] == "1", (res, err)
    assert res["hooks_ok"] == "1" and res["gains_seen"] == "1" and res["gains_bad"] == "0", res
    assert res["obs_ok"] == "1" and int(res["obs_count"]) >= 590, res
    assert 0.60 < float(res["min_h"]) and float(res["max_h"]) < 0.66 and float(res["max_tilt"]) < 0.08, res
    # estContactForce behind every estimator update (LeggedController.cpp:344-345) on the measured joint efforts: standing, the
    # momentum observer's two leg wrenches carry the robot's weight (12.587 kg)
    assert abs(float(res["cf_fz_sum"]) - 12.586944 * 9.81) < 0.1 * 12.586944 * 9.81, res
