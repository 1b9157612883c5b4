"""CPU: the C-ABI library loads and exports every symbol include/*.h declares (no compute without a GPU)."""
import ctypes as C
import re
from pathlib import Path

import pytest
import torch

from hunter_bipedal_control_amd import abi, solver

ROOT = Path(__file__).resolve().parents[1]


def test_header_and_library_symbols_agree():
    header = (ROOT / "include" / "hunter_hip.h").read_text()
    declared = sorted(set(re.findall(r"\b(hb_[a-z_0-9]+)\s*\(", header)))
    assert declared == sorted(solver.ABI_SYMBOLS)
    lib = solver.load_library()
    for name in declared:
        assert hasattr(lib, name), f"libhunter_hip.so does not export {name}"
    lib.hb_version.restype = C.c_int32
    assert lib.hb_version() >= 100
    lcm_header = (ROOT / "include" / "hunter_lcm.h").read_text()
    lcm_declared = sorted(set(re.findall(r"\b(hb_[a-z_0-9]+)\s*\(", lcm_header.split("extern \"C\"", 1)[1])))
    assert lcm_declared == sorted(solver.LCM_SYMBOLS)
    for name in lcm_declared:
        assert hasattr(lib, name), f"libhunter_hip.so does not export {name}"


def test_struct_sizes_match_the_header():
    # hb_model: 10 i32 + (30+30+40) f64 + 11 + 33 + 66 f64 + 4 i32 + 12 f64 + 1 f64
    assert C.sizeof(abi.HbModel) == 40 + 8 * (30 + 30 + 40 + 11 + 33 + 66) + 16 + 8 * 13
    assert C.sizeof(abi.HbStats) == 8 * 6 + 8 * 2 + 4 * 4
    assert C.sizeof(abi.HbConfig) % 8 == 0


@pytest.mark.skipif(torch.cuda.is_available(), reason="only meaningful on a machine without a GPU")
def test_no_cpu_fallback(params):
    with pytest.raises(solver.HunterHipError, match="no HIP device"):
        solver.HunterSolver(params, batch=1, max_nodes=4)
    lib = solver.load_library()
    out = C.c_void_p()
    rc = lib.hb_create(None, None, 1, 1, 0, C.byref(out))
    assert rc == -1        # HB_ERR_ARG, never a crash
