"""Generates tests/golden/ref_splines.json from the REFERENCE's own spline classes.

Run in the build container only (needs /root/reference):  make -C oracle ref && python tests/golden/make_ref_splines.py
oracle/_ref/libref_splines.so is legged_interface/src/foot_planner/{CubicSpline,MultiCubicSpline}.cpp compiled in place
(oracle/Makefile). The swing-phase node lists are built the way SwingTrajectoryPlanner::genSwingTrajs builds them
(SwingTrajectoryPlanner.cpp:315-358; that file needs Eigen / OCS2 and is not compilable here, the node formulas are
restated in refgen.SwingTrajectoryPlanner._swing_splines and stored in the fixture next to the reference's outputs).
"""
import ctypes as C
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from hunter_bipedal_control_amd import ingest  # noqa: E402
from oracle import refgen

lib = C.CDLL(str(ROOT / "oracle/_ref/libref_splines.so"))
_p = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))


def ref_eval(nodes, ts):
    nodes = np.ascontiguousarray(nodes, dtype=np.float64)
    ts = np.ascontiguousarray(ts, dtype=np.float64)
    out = np.zeros((len(ts), 3))
    lib.ref_multispline_eval(_p(nodes), C.c_int(len(nodes)), _p(ts), C.c_int(len(ts)), _p(out))
    return out


def main():
    params = ingest.load_packaged()
    sw = params["config"]["swing"]
    planner = refgen.SwingTrajectoryPlanner(sw)
    rng = np.random.default_rng(20260924)
    phases = []
    for k in range(24):
        t0 = float(rng.uniform(0.0, 5.0))
        dur = float([0.3, 0.3, 0.15, 0.45, 0.08, 0.6][k % 6] * rng.uniform(0.9, 1.1))
        t1 = t0 + dur
        p0 = rng.uniform([-0.3, -0.25, 0.0], [0.6, 0.25, 0.06])
        p1 = p0 + rng.uniform([-0.15, -0.08, -0.04], [0.25, 0.08, 0.04])
        if k % 4 == 0:
            p0[2] = p1[2] = sw["next_position_z"]
        sx, sy, sz = planner._swing_splines(t0, t1, p0, p1)
        # query times: inside every segment, exactly on every node, before the first and after the last node
        ts = np.concatenate([np.linspace(t0, t1, 13), [n[0] for n in sz.nodes], [n[0] for n in sx.nodes],
                             [t0 - 0.05, t1 + 0.05, np.nextafter(t1, 0.0), np.nextafter(t1, 10.0)]])
        phases.append(dict(t0=t0, t1=t1, p0=p0.tolist(), p1=p1.tolist(),
                           nodes={"x": sx.nodes, "y": sy.nodes, "z": sz.nodes}, t=ts.tolist(),
                           x=ref_eval(sx.nodes, ts).tolist(), y=ref_eval(sy.nodes, ts).tolist(), z=ref_eval(sz.nodes, ts).tolist()))
    # free-form multi-node splines (2..5 nodes, non-zero end velocities)
    generic = []
    for k in range(12):
        n = 2 + k % 4
        tn = np.cumsum(rng.uniform(0.05, 0.5, n)) + rng.uniform(-1, 1)
        nodes = [(float(tn[i]), float(rng.normal()), float(rng.normal())) for i in range(n)]
        ts = np.concatenate([np.linspace(tn[0] - 0.1, tn[-1] + 0.1, 17), tn])
        generic.append(dict(nodes=nodes, t=ts.tolist(), out=ref_eval(nodes, ts).tolist()))
    doc = dict(source="legged_interface/src/foot_planner/{CubicSpline,MultiCubicSpline}.cpp compiled in place (oracle/Makefile: ref)",
               columns=["position", "velocity", "acceleration"],
               swing=dict(swing_height=sw["swing_height"], swing_time_scale=sw["swing_time_scale"], next_position_z=sw["next_position_z"]),
               phases=phases, generic=generic)
    out = ROOT / "tests/golden/ref_splines.json"
    out.write_text(json.dumps(doc))
    print("wrote", out, out.stat().st_size, "bytes")


if __name__ == "__main__":
    main()
