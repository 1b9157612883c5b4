"""Generates tests/golden/ref_interface.json: the optimal control problem AS THE REFERENCE ASSEMBLES IT.

Run in the build container only (needs /root/reference):  make -C oracle ref && python tests/golden/make_ref_interface.py
oracle/_ref/libref_interface.so = legged_interface/src/LeggedInterface.cpp (whole), common/ModelSettings.cpp, gait/ModeSequenceTemplate.cpp,
dynamics/LeggedRobotDynamicsAD.cpp and the constraint / cost / initializer / reference-manager sources it instantiates, compiled in place
over holder stand-ins of the OCS2 classes (oracle/ref_shim_li/, oracle/ref_interface_capi.cpp) and EXECUTED on the reference's own
legged_controllers/config/hunter/{task.info, reference.info}: LeggedInterface(task, urdf, reference) + setupOptimalControlProblem.
The file records the named terms of every collection in the order the reference adds them and the parameters each was built with.
The rigid-body model behind the pinocchio stand-in (joint limits, masses, the contact-point Jacobians of initializeInputCostWeight) is
the hb_model of the packaged parameters: the URDF itself is not parsed by reference code (pinocchio's parser is not here).
"""
import ctypes as C
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from hunter_bipedal_control_amd import abi, ingest  # noqa: E402

CFG = "/root/reference/legged_controllers/config/hunter/"
URDF = "/root/reference/legged_examples/legged_hunter/legged_hunter_description/urdf/hunter.urdf"


def main():
    params = ingest.load_packaged()
    mdl = abi.make_model(params)
    lib = C.CDLL(str(ROOT / "oracle/_ref/libref_interface.so"))
    lib.refli_run.restype = C.c_int
    lib.refli_run.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]
    buf = C.create_string_buffer(1 << 20)
    urdf = URDF if Path(URDF).exists() else str(next(Path("/root/reference").rglob("*.urdf")))
    n = lib.refli_run(C.addressof(mdl), (CFG + "task.info").encode(), urdf.encode(), (CFG + "reference.info").encode(), buf, len(buf))
    if n < 0:
        raise SystemExit(f"refli_run failed ({n}): {buf.value.decode(errors='replace')}")
    doc = json.loads(buf.value.decode())
    doc["factory_calls"]["urdf"] = Path(doc["factory_calls"]["urdf"]).name
    doc["_generated_by"] = "tests/golden/make_ref_interface.py over oracle/_ref/libref_interface.so (reference code executed; see the script's header)"
    out = ROOT / "tests/golden/ref_interface.json"
    out.write_text(json.dumps(doc, indent=1) + "\n")
    print(f"wrote {out}: " + ", ".join(f"{k} {len(v)}" for k, v in doc.items() if isinstance(v, list)))


if __name__ == "__main__":
    main()
