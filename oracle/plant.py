"""Contact-consistent rigid-body plant stub for closed-loop rollouts (the step after the path).

The reference closes its loop through Gazebo / MuJoCo (legged_gazebo/src/LeggedHWSim.cpp:166-192, mujoco/src/main.cc:247),
neither of which exists here.  This stub integrates the floating-base equations of motion the WBC itself is written
against (WbcBase::formulateFloatingBaseEomTask, WbcBase.cpp:151-162) with the contact points of the commanded mode pinned
by acceleration-level constraints:

    M(q) vdot + nle(q, v) = S' tau + Jc' lambda
    Jc vdot = -dJc v - 2 a Jc v - a^2 (p_c - p_anchor)          (Baumgarte, a = `baumgarte`)

Both contact points of a foot sit on one rigid link (Jc has rank 5 per foot), so lambda is taken from the damped
normal equations (Jc M^-1 Jc' + eps I) lambda = rhs; the acceleration is unique.  Coordinates are pinocchio's
q = [pos, zyx, joints], v = [v_lin (world), ZYX rates, joint rates]; `rbd()` repacks them as the rbdState the estimator
would deliver (StateEstimateBase.cpp:73-106).  Rigid-body terms (M, nle, Jc, dJc v) come from a callback — the device
(`HunterSolver.eval_rbd`) in the GPU tests, the CPU oracle otherwise — so the plant itself has no model code.
Unilateral contact / friction limits are NOT enforced: it is a stub for regression tests of the controller, not a simulator.
"""
from __future__ import annotations

import numpy as np


def _E(zyx):
    """world angular velocity = E(zyx) @ ZYX rates (batched)."""
    z, y = zyx[:, 0], zyx[:, 1]
    E = np.zeros((zyx.shape[0], 3, 3))
    E[:, 0, 1], E[:, 0, 2] = -np.sin(z), np.cos(y) * np.cos(z)
    E[:, 1, 1], E[:, 1, 2] = np.cos(z), np.cos(y) * np.sin(z)
    E[:, 2, 0], E[:, 2, 2] = 1.0, -np.sin(y)
    return E


class Plant:
    def __init__(self, rbd_fn, foot_fn, q0, v0=None, baumgarte=30.0, eps=1e-8):
        """rbd_fn(rbd[B][32]) -> (M[B][16][16], nle[B][16], Jc[B][12][16], dJv[B][12]);
        foot_fn(q[B][16]) -> contact point positions [B][4][3]."""
        self.rbd_fn, self.foot_fn = rbd_fn, foot_fn
        self.q = np.array(q0, dtype=float)
        self.v = np.zeros_like(self.q) if v0 is None else np.array(v0, dtype=float)
        self.B = self.q.shape[0]
        self.baum, self.eps = baumgarte, eps
        self.anchor = self.foot_fn(self.q)
        self.pinned = np.zeros((self.B, 4), dtype=bool)
        self.last_lambda = np.zeros((self.B, 12))
        self.last_vdot = np.zeros((self.B, 16))

    def rbd(self):
        out = np.zeros((self.B, 32))
        out[:, 0:3], out[:, 3:6], out[:, 6:16] = self.q[:, 3:6], self.q[:, 0:3], self.q[:, 6:]
        out[:, 16:19] = np.einsum("bij,bj->bi", _E(self.q[:, 3:6]), self.v[:, 3:6])
        out[:, 19:22], out[:, 22:32] = self.v[:, 0:3], self.v[:, 6:]
        return out

    def step(self, tau, contact, dt, substeps=4):
        """tau[B][10], contact[B][4] (bool): advance by dt."""
        contact = np.asarray(contact, dtype=bool)
        feet = self.foot_fn(self.q)
        newly = contact & ~self.pinned
        self.anchor[newly] = feet[newly]            # a foot is pinned where it is when its contact phase starts
        self.pinned = contact.copy()
        h = dt / substeps
        rows = np.repeat(contact, 3, axis=1)        # [B][12]
        for _ in range(substeps):
            M, nle, J, dJv = self.rbd_fn(self.rbd())
            feet = self.foot_fn(self.q)
            rhs_q = -nle
            rhs_q[:, 6:] += tau
            Jm = J * rows[:, :, None]
            Minv_r = np.linalg.solve(M, rhs_q[:, :, None])[:, :, 0]
            Minv_Jt = np.linalg.solve(M, np.transpose(Jm, (0, 2, 1)))
            A = Jm @ Minv_Jt
            scale = np.maximum(np.trace(A, axis1=1, axis2=2), 1e-12)
            A = A + (self.eps * scale)[:, None, None] * np.eye(12) + np.where(rows, 0.0, 1.0)[:, :, None] * np.eye(12)
            vel_c = np.einsum("bij,bj->bi", Jm, self.v)
            err = ((feet - self.anchor).reshape(self.B, 12)) * rows
            b = (-dJv - 2 * self.baum * vel_c - self.baum ** 2 * err) * rows - np.einsum("bij,bj->bi", Jm, Minv_r)
            lam = np.linalg.solve(A, b[:, :, None])[:, :, 0] * rows
            vdot = Minv_r + np.einsum("bij,bj->bi", Minv_Jt, lam)
            self.v = self.v + h * vdot               # semi-implicit Euler
            self.q = self.q + h * self.v
            self.last_lambda = lam
            self.last_vdot = vdot
        return self.q, self.v

    def imu(self):
        """Ideal IMU of the base link for the state estimator: quaternion (x y z w), body-frame angular velocity and
        specific force (world linear acceleration minus gravity, rotated into the body), from the last step."""
        from .refgen import zyx_to_rotation
        quat, w_loc, a_loc = np.zeros((self.B, 4)), np.zeros((self.B, 3)), np.zeros((self.B, 3))
        w_world = np.einsum("bij,bj->bi", _E(self.q[:, 3:6]), self.v[:, 3:6])
        for i in range(self.B):
            R = zyx_to_rotation(self.q[i, 3:6])
            w = 0.5 * np.sqrt(max(1e-300, 1.0 + np.trace(R)))
            quat[i] = [(R[2, 1] - R[1, 2]) / (4 * w), (R[0, 2] - R[2, 0]) / (4 * w), (R[1, 0] - R[0, 1]) / (4 * w), w]
            w_loc[i] = R.T @ w_world[i]
            a_loc[i] = R.T @ (self.last_vdot[i, 0:3] + np.array([0.0, 0.0, 9.81]))
        return quat, w_loc, a_loc
