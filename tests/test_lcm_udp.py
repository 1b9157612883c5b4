"""LCM transport loop-back over real UDP sockets (include/hunter_lcm_udp.hpp: LCM's default udpm provider restricted to short "LC02"
datagrams, hb_lcm_frame / hb_lcm_unframe): LOWSTATE plant -> controller, LOWCMD controller -> plant, as
LeggedMujocoSim::read / write and mujoco's LcmInterface exchange them (legged_examples/legged_mujoco/src/LeggedMujocoSim.cpp:28-62,
mujoco/src/lcm_interface/LcmInterface.cpp:14,104-109).  CPU: codec + framing + sockets; -m gpu: the controller side is the device
path (hb_estimator_update_lcm on the received bytes, MPC + WBC, hb_joint_command_lcm)."""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
PKG = ROOT / "hunter_bipedal_control_amd"
URL = "udpm://239.255.76.67:17667?ttl=0"   # LCM's default group, a port of our own


def _build():
    lib = PKG / "libhunter_hip.so"
    if not lib.exists():
        pytest.skip("libhunter_hip.so not built (python __graft_entry__.py build)")
    out = ROOT / "tests" / "cpp" / "_build"
    out.mkdir(exist_ok=True)
    exe = out / "lcm_udp_test"
    src = ROOT / "tests" / "cpp" / "lcm_udp_test.cpp"
    deps = [src, lib, ROOT / "include/hunter_lcm_udp.hpp", ROOT / "include/hunter_hip.hpp"]
    if not exe.exists() or exe.stat().st_mtime < max(d.stat().st_mtime for d in deps):
        subprocess.check_call(["g++", "-std=c++14", "-O1", "-Wall", "-I", str(ROOT / "include"), str(src), "-L", str(PKG), "-lhunter_hip",
                               f"-Wl,-rpath,{PKG}", "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)])
    return exe


def _run(exe, *args):
    r = subprocess.run([str(exe), *map(str, args)], capture_output=True, text=True, timeout=120)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
    assert r.returncode == 0 and line, r.stdout + r.stderr
    tok = line[-1].split()[1:]
    res = dict(zip(tok[0::2], tok[1::2]))
    if "exception" in res and "multicast" in r.stderr:
        pytest.skip("this host has no multicast route: " + r.stderr.strip())
    return res, r.stderr


def test_frame_and_unframe_are_inverse():
    from hunter_bipedal_control_amd.solver import load_library
    lib = load_library()
    payload = np.arange(37, dtype=np.uint8)
    out = np.zeros(128, dtype=np.uint8)
    n = lib.hb_lcm_frame(b"LOWCMD", C.c_uint32(0xDEADBEEF), payload.ctypes.data_as(C.c_void_p), 37, out.ctypes.data_as(C.c_void_p), 128)
    assert n == 8 + 7 + 37 and bytes(out[:4]) == b"LC02" and bytes(out[4:8]) == bytes.fromhex("deadbeef")
    ch = C.create_string_buffer(64)
    seq, off = C.c_uint32(), C.c_int32()
    m = lib.hb_lcm_unframe(out.ctypes.data_as(C.c_void_p), n, ch, 64, C.byref(seq), C.byref(off))
    assert m == 37 and ch.value == b"LOWCMD" and seq.value == 0xDEADBEEF and bytes(out[off.value:off.value + m]) == bytes(payload)
    out[3] = ord("3")     # "LC03": a fragment header is not a short message
    assert lib.hb_lcm_unframe(out.ctypes.data_as(C.c_void_p), n, ch, 64, C.byref(seq), C.byref(off)) < 0
    assert lib.hb_lcm_unframe(out.ctypes.data_as(C.c_void_p), 9, ch, 64, C.byref(seq), C.byref(off)) < 0


def test_low_state_and_low_cmd_cross_a_udp_socket():
    res, err = _run(_build(), "host", URL)
    assert res.get("ok") == "1" and res["stamp"] == "123456789" and res["finite"] == "1", (res, err)
    assert float(res["pos0"]) == -3.0 and float(res["kp0"]) == 7.0 and float(res["ff3"]) == 5.25


@pytest.mark.gpu
def test_device_controller_side_over_udp():
    res, err = _run(_build(), "device", URL, PKG / "data" / "hunter_params.bin")
    assert res.get("ok") == "1" and res["finite"] == "1", (res, err)
    assert res["stamp"] == str(123456789 + 2000000)
    assert abs(float(res["pos0"]) - 0.09) < 0.05 and float(res["kp0"]) > 0.0      # posDes near the default joint angle, stance gain
