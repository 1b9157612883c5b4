"""TEST INFRASTRUCTURE: ctypes wrapper of oracle/_build/liboracle.so (the CPU oracle).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

from hunter_bipedal_control_amd import abi

_HERE = Path(__file__).resolve().parent
_LIB = _HERE / "_build" / "liboracle.so"
_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


def build(force: bool = False) -> Path:
    if force or not _LIB.exists():
        subprocess.check_call(["make", "-C", str(_HERE)] + (["-B"] if force else []))
    return _LIB


def _opt(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Oracle:
    def __init__(self, params: dict, **cfg_overrides):
        build()
        self.lib = C.CDLL(str(_LIB))
        self.model = abi.make_model(params)
        self.config = abi.make_config(params, **cfg_overrides)
        self.lib.orc_create.restype = C.c_void_p
        self.h = C.c_void_p(self.lib.orc_create(C.byref(self.model), C.byref(self.config)))
        self.lib.orc_relaxed_barrier.restype = C.c_double
        self.lib.orc_relaxed_barrier.argtypes = [C.c_double, C.c_double, C.c_double, C.c_int]

    def __del__(self):
        try:
            self.lib.orc_destroy(self.h)
        except Exception:
            pass

    # ---- model ---------------------------------------------------------------------------------
    def input_cost(self):
        R = np.zeros((22, 22))
        self.lib.orc_input_cost(self.h, _opt(R))
        return R

    def relaxed_barrier(self, mu, delta, h, order=0):
        return self.lib.orc_relaxed_barrier(mu, delta, h, order)

    def friction_cone(self, F):
        """h, grad (3), Hessian (3x3) of the friction cone at contact force F, and the configured hessianDiagonalShift."""
        F = np.ascontiguousarray(F, dtype=np.float64)
        out = np.zeros(14)
        self.lib.orc_friction_cone(self.h, _opt(F), _opt(out))
        return out[0], out[1:4].copy(), out[4:13].reshape(3, 3).copy(), out[13]

    def flow_map(self, x, u, jac=False):
        x = np.ascontiguousarray(np.atleast_2d(x), dtype=np.float64)
        u = np.ascontiguousarray(np.atleast_2d(u), dtype=np.float64)
        n = x.shape[0]
        f = np.zeros((n, 22))
        A = np.zeros((n, 22, 22)) if jac else None
        B = np.zeros((n, 22, 22)) if jac else None
        self.lib.orc_flow_map(self.h, C.c_int(n), _opt(x), _opt(u), _opt(f), _opt(A), _opt(B))
        return (f, A, B) if jac else f

    def foot_kinematics(self, x, u):
        x = np.ascontiguousarray(np.atleast_2d(x), dtype=np.float64)
        u = np.ascontiguousarray(np.atleast_2d(u), dtype=np.float64)
        n = x.shape[0]
        pos, vel = np.zeros((n, 4, 3)), np.zeros((n, 4, 3))
        self.lib.orc_foot_kinematics(self.h, C.c_int(n), _opt(x), _opt(u), _opt(pos), _opt(vel))
        return pos, vel

    def centroidal_matrix(self, q):
        q = np.ascontiguousarray(q, dtype=np.float64)
        A, com = np.zeros((6, 16)), np.zeros(3)
        self.lib.orc_centroidal_matrix(self.h, _opt(q), _opt(A), _opt(com))
        return A, com

    def rbd(self, rbd):
        rbd = np.ascontiguousarray(np.atleast_2d(rbd), dtype=np.float64)
        n = rbd.shape[0]
        M, nle, J, dJv = np.zeros((n, 16, 16)), np.zeros((n, 16)), np.zeros((n, 12, 16)), np.zeros((n, 12))
        self.lib.orc_rbd(self.h, C.c_int(n), _opt(rbd), _opt(M), _opt(nle), _opt(J), _opt(dJv))
        return M, nle, J, dJv

    def rbd_qv(self, q, v):
        q = np.ascontiguousarray(q, dtype=np.float64)
        v = np.ascontiguousarray(v, dtype=np.float64)
        M, nle, J, dJv = np.zeros((16, 16)), np.zeros(16), np.zeros((12, 16)), np.zeros(12)
        self.lib.orc_rbd_qv(self.h, _opt(q), _opt(v), _opt(M), _opt(nle), _opt(J), _opt(dJv))
        return M, nle, J, dJv

    def rbd_full(self, q, v):
        """Everything WbcBase::updateMeasured takes out of pinocchio (full matrices; oracle_capi.cpp orc_rbd_full)."""
        q = np.ascontiguousarray(q, dtype=np.float64)
        v = np.ascontiguousarray(v, dtype=np.float64)
        o = dict(M=np.zeros((16, 16)), nle=np.zeros(16), J=np.zeros((12, 16)), dJ=np.zeros((12, 16)), Jb=np.zeros((6, 16)),
                 dJb=np.zeros((6, 16)), ee_pos=np.zeros((4, 3)), ee_vel=np.zeros((4, 3)))
        self.lib.orc_rbd_full(self.h, _opt(q), _opt(v), *[_opt(o[k]) for k in ("M", "nle", "J", "dJ", "Jb", "dJb", "ee_pos", "ee_vel")])
        return o

    def desired_kinematics(self, x, u):
        x = np.ascontiguousarray(x, dtype=np.float64)
        u = np.ascontiguousarray(u, dtype=np.float64)
        bp, bv, ba, fp, fv = np.zeros(6), np.zeros(6), np.zeros(6), np.zeros((4, 3)), np.zeros((4, 3))
        self.lib.orc_desired_kinematics(self.h, _opt(x), _opt(u), _opt(bp), _opt(bv), _opt(ba), _opt(fp), _opt(fv))
        return dict(base_pose=bp, base_vel=bv, base_acc=ba, foot_pos=fp, foot_vel=fv)

    # ---- MPC -----------------------------------------------------------------------------------
    def node_lq(self, dt, mode, x_ref, swing, x, u, x_next):
        c = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        x_ref, swing, x, u, x_next = c(x_ref), c(swing), c(x), c(u), c(x_next)
        o = dict(A=np.zeros((22, 22)), B=np.zeros((22, 22)), b=np.zeros(22), Q=np.zeros((22, 22)), R=np.zeros((22, 22)),
                 P=np.zeros((22, 22)), q=np.zeros(22), r=np.zeros(22), C=np.zeros((16, 22)), D=np.zeros((16, 22)),
                 e=np.zeros(16), Px=np.zeros((22, 22)), Pe=np.zeros(22))
        cost = C.c_double()
        rank = C.c_int()
        m = self.lib.orc_node_lq(self.h, C.c_double(dt), C.c_int(mode), _opt(x_ref), _opt(swing), _opt(x), _opt(u),
                                 _opt(x_next), _opt(o["A"]), _opt(o["B"]), _opt(o["b"]), _opt(o["Q"]), _opt(o["R"]),
                                 _opt(o["P"]), _opt(o["q"]), _opt(o["r"]), _opt(o["C"]), _opt(o["D"]), _opt(o["e"]),
                                 C.byref(cost), C.byref(rank), _opt(o["Px"]), _opt(o["Pe"]))
        # C and D were written compactly as m x 22
        o["C"] = o["C"].reshape(-1)[: m * 22].reshape(m, 22).copy()
        o["D"] = o["D"].reshape(-1)[: m * 22].reshape(m, 22).copy()
        o["e"] = o["e"][:m].copy()
        o["cost"], o["rank"], o["m"] = cost.value, rank.value, m
        return o

    def stage_pieces(self, mode, x_ref, swing, x, u):
        """Pieces of one node's stage terms as stage_terms computes them (ocp.hpp StageDebug): foot kinematics with gradients over
        (x, u), the tracking cost on its own, the xy soft rows of the swing feet."""
        c = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        x_ref, swing, x, u = c(x_ref), c(swing), c(x), c(u)
        o = dict(pos=np.zeros((4, 3)), vel=np.zeros((4, 3)), dpos=np.zeros((4, 3, 44)), dvel=np.zeros((4, 3, 44)), track=np.zeros(3),
                 track_q=np.zeros(22), track_r=np.zeros(22), xy=np.zeros((4, 2)), dxy=np.zeros((4, 2, 44)))
        self.lib.orc_stage_pieces(self.h, C.c_int(mode), _opt(x_ref), _opt(swing), _opt(x), _opt(u), _opt(o["pos"]), _opt(o["vel"]),
                                  _opt(o["dpos"]), _opt(o["dvel"]), _opt(o["track"]), _opt(o["track_q"]), _opt(o["track_r"]), _opt(o["xy"]),
                                  _opt(o["dxy"]))
        return o

    def riccati(self, A, B, b, Q, R, P, q, r, dx0):
        c = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        A, B, b, Q, R, P, q, r, dx0 = map(c, (A, B, b, Q, R, P, q, r, dx0))
        N, nu = A.shape[0], B.shape[2]
        dx, du = np.zeros((N + 1, 22)), np.zeros((N, nu))
        rc = self.lib.orc_riccati(C.c_int(N), C.c_int(nu), _opt(A), _opt(B), _opt(b), _opt(Q), _opt(R), _opt(P), _opt(q),
                                  _opt(r), _opt(dx0), _opt(dx), _opt(du))
        if rc != 0:
            raise RuntimeError("riccati: non-positive pivot")
        return dx, du

    def cold_start(self, mode, x0):
        mode = np.ascontiguousarray(mode, dtype=np.int32)
        N = mode.shape[0]
        x, u = np.zeros((N + 1, 22)), np.zeros((N, 22))
        self.lib.orc_cold_start(self.h, C.c_int(N), _opt(mode), _opt(np.ascontiguousarray(x0, dtype=np.float64)), _opt(x), _opt(u))
        return x, u

    def performance(self, t, mode, x_ref, swing, x, u):
        c = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        mode = np.ascontiguousarray(mode, dtype=np.int32)
        out = np.zeros(3)
        self.lib.orc_performance(self.h, C.c_int(mode.shape[0]), _opt(c(t)), _opt(mode), _opt(c(x_ref)), _opt(c(swing)),
                                 _opt(c(x)), _opt(c(u)), _opt(out))
        return out

    def mpc_solve(self, refs: dict, x0, x, u, iters=1, threads=1, want_step=False):
        """refs: dict(n_nodes[n], t[n][Nmax+1], mode[n][Nmax], x_ref[n][Nmax][22], swing[n][Nmax][4][6]).
        x, u are modified in place. Returns perf[n][4] (and the QP step dx, du if want_step)."""
        c = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        n_nodes = np.ascontiguousarray(refs["n_nodes"], dtype=np.int32)
        mode = np.ascontiguousarray(refs["mode"], dtype=np.int32)
        n, nmax = mode.shape
        t, x_ref, swing, x0 = c(refs["t"]), c(refs["x_ref"]), c(refs["swing"]), c(x0)
        assert x.flags.c_contiguous and u.flags.c_contiguous and x.shape == (n, nmax + 1, 22) and u.shape == (n, nmax, 22)
        perf = np.zeros((n, 4))
        dx = np.zeros_like(x) if want_step else None
        du = np.zeros_like(u) if want_step else None
        rc = self.lib.orc_mpc_solve(self.h, C.c_int(n), C.c_int(nmax), _opt(n_nodes), _opt(t), _opt(mode), _opt(x_ref),
                                    _opt(swing), _opt(x0), _opt(x), _opt(u), _opt(perf), C.c_int(iters), C.c_int(threads),
                                    _opt(dx), _opt(du))
        if rc != 0:
            raise RuntimeError(f"oracle mpc_solve: {-rc} instances failed (Riccati pivot)")
        return (perf, dx, du) if want_step else perf

    # ---- WBC -----------------------------------------------------------------------------------
    def wbc_update(self, x_des, u_des, rbd, mode, stance_flag=None, sol_prev=None, threads=1):
        c = lambda a: np.ascontiguousarray(np.atleast_2d(a), dtype=np.float64)
        x_des, u_des, rbd = c(x_des), c(u_des), c(rbd)
        n = x_des.shape[0]
        mode = np.ascontiguousarray(np.atleast_1d(mode), dtype=np.int32)
        stance = None if stance_flag is None else np.ascontiguousarray(np.atleast_1d(stance_flag), dtype=np.int32)
        sol = np.zeros((n, 38)) if sol_prev is None else np.ascontiguousarray(sol_prev, dtype=np.float64).copy()
        status, iters = np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int32)
        self.lib.orc_wbc_update(self.h, C.c_int(n), _opt(x_des), _opt(u_des), _opt(rbd), _opt(mode), _opt(stance), _opt(sol),
                                _opt(status), _opt(iters), C.c_int(threads))
        return sol, status, iters

    def wbc_problem(self, x_des, u_des, rbd, mode, stance=False):
        c = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        Aeq, beq = np.zeros((40, 38)), np.zeros(40)
        Din, fin = np.zeros((60, 38)), np.zeros(60)
        Aw, bw = np.zeros((40, 38)), np.zeros(40)
        ne, ni, nw = C.c_int(), C.c_int(), C.c_int()
        self.lib.orc_wbc_problem(self.h, _opt(c(x_des)), _opt(c(u_des)), _opt(c(rbd)), C.c_int(mode), C.c_int(int(stance)),
                                 _opt(Aeq), _opt(beq), C.byref(ne), _opt(Din), _opt(fin), C.byref(ni), _opt(Aw), _opt(bw),
                                 C.byref(nw))
        r = lambda M, k: M.reshape(-1)[: k * 38].reshape(k, 38).copy()
        return dict(Aeq=r(Aeq, ne.value), beq=beq[: ne.value].copy(), D=r(Din, ni.value), f=fin[: ni.value].copy(),
                    Aw=r(Aw, nw.value), bw=bw[: nw.value].copy())

    def lsqp(self, A, b, eps, E, e, D, f, max_iter=200, reg_steps=0):
        c = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        A, b, E, e, D, f = map(c, (A, b, E, e, D, f))
        n = max(A.shape[1] if A.size else 0, E.shape[1] if E.size else 0, D.shape[1] if D.size else 0)
        x = np.zeros(n)
        it = C.c_int()
        st = self.lib.orc_lsqp(C.c_int(n), C.c_int(A.shape[0]), _opt(A), _opt(b), C.c_double(eps), C.c_int(E.shape[0]),
                               _opt(E), _opt(e), C.c_int(D.shape[0]), _opt(D), _opt(f), C.c_int(max_iter), C.c_int(reg_steps), _opt(x),
                               C.byref(it))
        return x, st, it.value

    # ---- contact-force estimate (StateEstimateBase::estContactForce) ------------------------------------
    def contact_force(self, cutoff_frequency, dt, z, rbd, tau):
        """z [n][16] is updated in place (pSCgZinvlast_); -> (estDisturbancetorque_ [n][16], estContactforce_ [n][16])."""
        c = lambda a: np.ascontiguousarray(np.atleast_2d(a), dtype=np.float64)
        rbd, tau = c(rbd), c(tau)
        n = rbd.shape[0]
        assert z.flags.c_contiguous and z.shape == (n, 16) and z.dtype == np.float64
        dist, cf = np.zeros((n, 16)), np.zeros((n, 16))
        self.lib.orc_contact_force(C.byref(self.model), C.c_double(cutoff_frequency), C.c_int(n), C.c_double(dt), _opt(z), _opt(rbd), _opt(tau),
                                   _opt(dist), _opt(cf))
        return dist, cf

    def contact_force_rbd(self, q, v):
        q, v = np.ascontiguousarray(q, dtype=np.float64), np.ascontiguousarray(v, dtype=np.float64)
        o = dict(M=np.zeros((16, 16)), g=np.zeros(16), CTv=np.zeros(16), J6=np.zeros((2, 6, 16)))
        self.lib.orc_contact_force_rbd(C.byref(self.model), _opt(q), _opt(v), _opt(o["M"]), _opt(o["g"]), _opt(o["CTv"]), _opt(o["J6"]))
        return o

    # ---- hierarchical QP / HierarchicalWbc ------------------------------------------------------------
    def hoqp(self, tasks, eps=1e-8, max_iter=500, reg_steps=1):
        """tasks: list (highest priority first) of dicts with optional A,b (A x = b) and D,f (D x <= f)."""
        n = max(max(t["A"].shape[1] if "A" in t and t["A"].size else 0, t["D"].shape[1] if "D" in t and t["D"].size else 0) for t in tasks)
        c = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        A = c(np.vstack([t.get("A", np.zeros((0, n))).reshape(-1, n) for t in tasks]))
        b = c(np.concatenate([np.atleast_1d(t.get("b", np.zeros(0))) for t in tasks]))
        D = c(np.vstack([t.get("D", np.zeros((0, n))).reshape(-1, n) for t in tasks]))
        f = c(np.concatenate([np.atleast_1d(t.get("f", np.zeros(0))) for t in tasks]))
        mA = np.array([t.get("A", np.zeros((0, n))).reshape(-1, n).shape[0] for t in tasks], dtype=np.int32)
        mD = np.array([t.get("D", np.zeros((0, n))).reshape(-1, n).shape[0] for t in tasks], dtype=np.int32)
        x, slack, ns = np.zeros(n), np.zeros(max(1, int(mD.sum()))), C.c_int()
        st = self.lib.orc_hoqp(C.c_int(n), C.c_int(len(tasks)), _opt(mA), _opt(A), _opt(b), _opt(mD), _opt(D), _opt(f),
                               C.c_double(eps), C.c_int(max_iter), C.c_int(reg_steps), _opt(x), _opt(slack), C.byref(ns))
        return x, slack[: ns.value], st

    def hwbc_update(self, x_des, u_des, rbd, mode, threads=1):
        c = lambda a: np.ascontiguousarray(np.atleast_2d(a), dtype=np.float64)
        x_des, u_des, rbd = c(x_des), c(u_des), c(rbd)
        n = x_des.shape[0]
        mode = np.ascontiguousarray(np.atleast_1d(mode), dtype=np.int32)
        sol, status = np.zeros((n, 38)), np.zeros(n, dtype=np.int32)
        self.lib.orc_hwbc_update(self.h, C.c_int(n), _opt(x_des), _opt(u_des), _opt(rbd), _opt(mode), _opt(sol), _opt(status), C.c_int(threads))
        return sol, status

    def hwbc_tasks(self, x_des, u_des, rbd, mode, level):
        c = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        A, b, D, f = np.zeros((64, 38)), np.zeros(64), np.zeros((64, 38)), np.zeros(64)
        mA, mD = C.c_int(), C.c_int()
        self.lib.orc_hwbc_tasks(self.h, _opt(c(x_des)), _opt(c(u_des)), _opt(c(rbd)), C.c_int(mode), C.c_int(level), _opt(A), _opt(b),
                                C.byref(mA), _opt(D), _opt(f), C.byref(mD))
        r = lambda M, k: M.reshape(-1)[: k * 38].reshape(k, 38).copy()
        return dict(A=r(A, mA.value), b=b[: mA.value].copy(), D=r(D, mD.value), f=f[: mD.value].copy())

    # ---- state estimator (oracle/estimator.hpp) ---------------------------------------------------------------------
    def kf_init(self, n):
        """-> dict(xhat[n][18], P[n][18][18] = 100 I, yaw_last[n])."""
        return dict(xhat=np.zeros((n, 18)), P=np.tile(100.0 * np.eye(18), (n, 1, 1)), yaw_last=np.zeros(n))

    def kf_update(self, est_cfg, state, dt, quat, w_local, a_local, qj, qdj, contact):
        """One estimator tick for n instances; `state` (from kf_init) is updated in place. -> rbd[n][32], x[n][22]."""
        c = lambda a: np.ascontiguousarray(np.atleast_2d(a), dtype=np.float64)
        quat, w_local, a_local, qj, qdj = c(quat), c(w_local), c(a_local), c(qj), c(qdj)
        contact = np.ascontiguousarray(np.atleast_2d(contact), dtype=np.int32)
        n = quat.shape[0]
        rbd, x = np.zeros((n, 32)), np.zeros((n, 22))
        self.lib.orc_kf_update(C.byref(self.model), C.byref(est_cfg), C.c_int(n), C.c_double(dt), _opt(state["xhat"]), _opt(state["P"]),
                               _opt(state["yaw_last"]), _opt(quat), _opt(w_local), _opt(a_local), _opt(qj), _opt(qdj), _opt(contact),
                               _opt(rbd), _opt(x))
        return rbd, x

    def centroidal_state_from_rbd(self, rbd):
        rbd = np.ascontiguousarray(np.atleast_2d(rbd), dtype=np.float64)
        x = np.zeros((rbd.shape[0], 22))
        self.lib.orc_centroidal_state_from_rbd(C.byref(self.model), C.c_int(rbd.shape[0]), _opt(rbd), _opt(x))
        return x
