"""Generates tests/golden/ref_constraints.json from the REFERENCE's own constraint code.

Run in the build container only (needs /root/reference):  make -C oracle ref && python tests/golden/make_ref_constraints.py
oracle/_ref/libref_constraints.so is legged_interface/src/constraint/{FrictionConeConstraint, ZeroForceConstraint}.cpp of the
reference compiled in place (oracle/Makefile, oracle/ref_constraints_capi.cpp).  Every number under an "out" key was computed
by that library.  Sections:
  cone        getValue / getLinearApproximation / getQuadraticApproximation of the friction cone at seeded inputs: default
              Config (FrictionConeConstraint.h:77-83) and perturbed settings; forces with zero, small and large tangential parts
  zero_force  getLinearApproximation of the zero-force constraint, isActive for every contact-flag pattern
"""
import ctypes as C
import json
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
lib = C.CDLL(str(ROOT / "oracle/_ref/libref_constraints.so"))
DP, IP = C.POINTER(C.c_double), C.POINTER(C.c_int)
_d = lambda a: a.ctypes.data_as(DP)
lib.ref_friction_cone.argtypes = [DP, C.c_int, IP, DP, DP, DP, DP, DP, DP, DP, DP, DP]
lib.ref_zero_force.argtypes = [C.c_int, IP, DP, DP, DP, DP, DP]


def main():
    rng = np.random.default_rng(11)
    out = dict(cone=[], zero_force=[])
    forces = [np.zeros(3), np.array([0.0, 0.0, 30.87]), np.array([1e-9, -1e-9, 5.0]), np.array([40.0, -25.0, 60.0]), np.array([3.0, 4.0, -2.0])]
    forces += [np.array([8 * rng.standard_normal(), 8 * rng.standard_normal(), 30 + 15 * rng.standard_normal()]) for _ in range(15)]
    cfgs = [None, [0.7, 25.0, 0.0, 1e-6], [0.5, 1.0, 3.0, 1e-4], [1.1, 100.0, 0.0, 0.0]]
    for k, F in enumerate(forces):
        for ci, cfg in enumerate(cfgs if k < 8 else cfgs[:1]):
            idx = k % 4
            x, u = rng.standard_normal(22), rng.standard_normal(22)
            u[3 * idx:3 * idx + 3] = F
            flags = np.array([(k >> b) & 1 for b in range(4)], dtype=np.int32)
            f, vo = C.c_double(), C.c_double()
            dfdx, dfdu = np.zeros(22), np.zeros(22)
            dfdxx, dfduu, dfdux = np.zeros((22, 22)), np.zeros((22, 22)), np.zeros((22, 22))
            cfa = None if cfg is None else np.array(cfg, dtype=np.float64)
            act = lib.ref_friction_cone(None if cfa is None else _d(cfa), idx, flags.ctypes.data_as(IP), _d(x), _d(u), C.byref(f), _d(dfdx), _d(dfdu),
                                        _d(dfdxx), _d(dfduu), _d(dfdux), C.byref(vo))
            assert act >= 0 and f.value == vo.value
            assert not dfdx.any() and not dfdux.any()
            # compact storage: the 3x3 force block and the diagonals of the two Hessians (everything else is asserted zero here)
            blk = dfduu[3 * idx:3 * idx + 3, 3 * idx:3 * idx + 3].copy()
            off = dfduu.copy()
            off[3 * idx:3 * idx + 3, 3 * idx:3 * idx + 3] = 0.0
            np.fill_diagonal(off, 0.0)
            assert not off.any()
            offx = dfdxx.copy()
            np.fill_diagonal(offx, 0.0)
            assert not offx.any()
            g = dfdu[3 * idx:3 * idx + 3].copy()
            rest = dfdu.copy()
            rest[3 * idx:3 * idx + 3] = 0.0
            assert not rest.any()
            out["cone"].append(dict(contact=idx, flags=flags.tolist(), config=cfg, force=F.tolist(),
                                    out=dict(active=bool(act), h=f.value, grad=g.tolist(), hess_block=blk.tolist(),
                                             dfduu_diag=np.diag(dfduu).tolist(), dfdxx_diag=np.diag(dfdxx).tolist())))
    for pattern in range(16):
        flags = np.array([(pattern >> b) & 1 for b in range(4)], dtype=np.int32)
        for idx in range(4):
            x, u = rng.standard_normal(22), 10 * rng.standard_normal(22)
            f, dfdx, dfdu = np.zeros(3), np.zeros((3, 22)), np.zeros((3, 22))
            act = lib.ref_zero_force(idx, flags.ctypes.data_as(IP), _d(x), _d(u), _d(f), _d(dfdx), _d(dfdu))
            out["zero_force"].append(dict(contact=idx, flags=flags.tolist(), u=u.tolist(),
                                          out=dict(active=bool(act), f=f.tolist(), dfdx_absmax=float(np.abs(dfdx).max()),
                                                   dfdu_nonzeros=[[int(r), int(c), float(dfdu[r, c])] for r, c in zip(*np.nonzero(dfdu))])))
    dst = ROOT / "tests/golden/ref_constraints.json"
    dst.write_text(json.dumps(out, indent=0))
    print(f"wrote {dst} ({dst.stat().st_size / 1024:.0f} KiB)")


if __name__ == "__main__":
    main()
