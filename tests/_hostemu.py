"""Build of tests/host_emu/libhostemu.so (the device headers compiled for the host: the emulator the CPU tests drive).  Several test
modules use it and pytest-xdist workers may ask for it at the same time: one build under a file lock, installed atomically, rebuilt
when a source it includes is newer."""
import fcntl
import os
import subprocess
from pathlib import Path

HERE = Path(__file__).resolve().parent / "host_emu"
CSRC = Path(__file__).resolve().parents[1] / "hunter_bipedal_control_amd" / "csrc"


def build() -> Path:
    so = HERE / "libhostemu.so"
    deps = [HERE / "hostemu.cpp", *CSRC.glob("*.hpp")]
    with open(HERE / ".hostemu.lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not so.exists() or so.stat().st_mtime < max(d.stat().st_mtime for d in deps):
            tmp = HERE / f"libhostemu.{os.getpid()}.so"
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-w", "-o", str(tmp), str(HERE / "hostemu.cpp")])
            os.replace(tmp, so)
    return so
