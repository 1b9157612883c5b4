#!/usr/bin/env python3
"""Flatten the reference's hunter.urdf + task.info + reference.info + gait.info into
hunter_bipedal_control_amd/data/hunter_params.json (numbers only; no reference text is copied).

Run in the build container where /root/reference exists:
    python tools/make_hunter_params.py [/root/reference]
"""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from hunter_bipedal_control_amd import ingest  # noqa: E402

ref = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
cfgdir = ref / "legged_controllers/config/hunter"
urdf = ref / "legged_examples/legged_hunter/legged_hunter_description/urdf/hunter.urdf"
out = dict(
    source=dict(urdf=str(urdf.relative_to(ref)), task=str((cfgdir / "task.info").relative_to(ref)),
                reference=str((cfgdir / "reference.info").relative_to(ref)),
                gait=str((cfgdir / "gait.info").relative_to(ref))),
    model=ingest.read_urdf(urdf),
    config=ingest.read_config(cfgdir / "task.info", cfgdir / "reference.info", cfgdir / "gait.info"),
)
dst = Path(__file__).resolve().parents[1] / "hunter_bipedal_control_amd/data/hunter_params.json"
dst.write_text(json.dumps(out, indent=1))
print("wrote", dst, "total mass", sum(out["model"]["mass"]))

from hunter_bipedal_control_amd import abi  # noqa: E402

blob = dst.with_suffix(".bin")
abi.write_params_blob(out, blob)
print("wrote", blob)
