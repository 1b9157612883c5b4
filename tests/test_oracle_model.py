"""CPU: the oracle's rigid-body / centroidal model against the known answers the reference holds and against
independent numerical invariants (finite differences, Lagrangian identity).  SURVEY.md §8c."""
import json
from pathlib import Path

import numpy as np
import pytest

GOLD = json.loads((Path(__file__).parent / "golden" / "reference_known_answers.json").read_text())


def _x0(params):
    return np.array(params["config"]["initial_state"])


def test_total_mass_and_nominal_force(params, oracle):
    m = sum(params["model"]["mass"])
    assert abs(m - GOLD["total_mass"]) < 1e-9
    # weightCompensatingInput: m g / 4 per contact in stance (utils.h:82-83)
    x, u = oracle.cold_start(np.full(3, 3, dtype=np.int32), _x0(params))
    assert np.allclose(u[0, [2, 5, 8, 11]], GOLD["stance_fz_per_contact"], atol=1e-9)
    assert np.allclose(np.delete(u[0], [2, 5, 8, 11]), 0.0)


def test_default_stance_foot_positions(params, oracle):
    x = _x0(params).copy()
    x[6:9] = 0.0
    pos, vel = oracle.foot_kinematics(x, np.zeros(22))
    assert np.abs(pos[0] - np.array(GOLD["default_stance_feet"])).max() < 5e-5
    assert np.abs(vel).max() < 1e-12
    # x and z agree with the planner's feet biases (task.info:28-31)
    assert abs(pos[0, 0, 0] - GOLD["feet_bias_x1"]) < 1e-3 and abs(pos[0, 2, 0] - GOLD["feet_bias_x2"]) < 1e-3
    assert abs(pos[0, :, 2].mean() + params["config"]["com_height"] - 0.0) < 2e-3


def test_relaxed_barrier_matches_reference_formula(oracle):
    for mu, delta, h, val in GOLD["relaxed_barrier_samples"]:
        assert abs(oracle.relaxed_barrier(mu, delta, h, 0) - val) < 1e-12
    # C1/C2 continuity at h = delta and derivative consistency
    mu, delta = 0.1, 5.0
    for h in (0.5, 4.9, 5.1, 20.0):
        e = 1e-6
        d1 = (oracle.relaxed_barrier(mu, delta, h + e, 0) - oracle.relaxed_barrier(mu, delta, h - e, 0)) / (2 * e)
        d2 = (oracle.relaxed_barrier(mu, delta, h + e, 1) - oracle.relaxed_barrier(mu, delta, h - e, 1)) / (2 * e)
        assert abs(d1 - oracle.relaxed_barrier(mu, delta, h, 1)) < 1e-7
        assert abs(d2 - oracle.relaxed_barrier(mu, delta, h, 2)) < 1e-7


def test_flow_map_jacobian_vs_finite_differences(params, oracle):
    rng = np.random.default_rng(0)
    for _ in range(3):
        x = _x0(params) + 0.15 * rng.standard_normal(22)
        u = rng.standard_normal(22) * np.r_[np.full(12, 10.0), np.full(10, 0.5)]
        f, A, B = oracle.flow_map(x, u, jac=True)
        eps = 1e-6
        for j in range(22):
            d = np.zeros(22)
            d[j] = eps
            An = (oracle.flow_map(x + d, u) - oracle.flow_map(x - d, u))[0] / (2 * eps)
            Bn = (oracle.flow_map(x, u + d) - oracle.flow_map(x, u - d))[0] / (2 * eps)
            assert np.abs(An - A[0][:, j]).max() < 1e-6
            assert np.abs(Bn - B[0][:, j]).max() < 1e-6
    # structure: momentum rows do not depend on the base position, joint rows are the joint velocities
    assert np.abs(A[0][:, 6:9]).max() == 0.0
    assert np.allclose(B[0][12:, 12:], np.eye(10)) and np.abs(A[0][12:]).max() == 0.0


def test_centroidal_momentum_matrix(params, oracle):
    rng = np.random.default_rng(1)
    q = np.r_[0.1 * rng.standard_normal(3), 0.3 * rng.standard_normal(3), _x0(params)[12:] + 0.2 * rng.standard_normal(10)]
    A, com = oracle.centroidal_matrix(q)
    m = sum(params["model"]["mass"])
    assert np.allclose(A[:3, :3], m * np.eye(3), atol=1e-12) and np.abs(A[3:, :3]).max() < 1e-12
    v = rng.standard_normal(16)
    eps = 1e-6
    _, cp = oracle.centroidal_matrix(q + eps * v)
    _, cm = oracle.centroidal_matrix(q - eps * v)
    assert np.abs(A[:3] @ v - m * (cp - cm) / (2 * eps)).max() < 1e-7   # linear momentum = m d(com)/dt


def test_mass_matrix_and_bias_forces_lagrangian_identity(params, oracle):
    """nle = Mdot v - 1/2 d(v'Mv)/dq + dV/dq with V = m g z_com, using only M(q) and com(q) by finite differences."""
    rng = np.random.default_rng(2)
    m, g = sum(params["model"]["mass"]), params["model"]["gravity"]
    q = np.r_[0.1 * rng.standard_normal(3), 0.2 * rng.standard_normal(3), _x0(params)[12:] + 0.2 * rng.standard_normal(10)]
    v = 0.7 * rng.standard_normal(16)
    M, nle, J, dJv = oracle.rbd_qv(q, v)
    assert np.abs(M - M.T).max() < 1e-12 and np.linalg.eigvalsh(M).min() > 0
    eps = 1e-5
    Mp, *_ = oracle.rbd_qv(q + eps * v, v)
    Mm, *_ = oracle.rbd_qv(q - eps * v, v)
    Mdot_v = (Mp - Mm) @ v / (2 * eps)
    grad_T, grad_V = np.zeros(16), np.zeros(16)
    for k in range(16):
        d = np.zeros(16)
        d[k] = eps
        Ma, *_ = oracle.rbd_qv(q + d, v)
        Mb, *_ = oracle.rbd_qv(q - d, v)
        grad_T[k] = 0.5 * v @ (Ma - Mb) @ v / (2 * eps)
        grad_V[k] = m * g * (oracle.centroidal_matrix(q + d)[1][2] - oracle.centroidal_matrix(q - d)[1][2]) / (2 * eps)
    assert np.abs(nle - (Mdot_v - grad_T + grad_V)).max() < 1e-6
    # contact Jacobian time variation
    Jp = oracle.rbd_qv(q + eps * v, v)[2]
    Jm = oracle.rbd_qv(q - eps * v, v)[2]
    assert np.abs(dJv - (Jp - Jm) @ v / (2 * eps)).max() < 1e-6


def test_input_cost_is_jacobian_pullback(params, oracle):
    """R = blkdiag(5e-3 I12, J' (2.0 I12) J) with J the contact Jacobian wrt the joints at the initial state
    (LeggedInterface.cpp:263-290, task.info:220-253 with scaling 1e-3)."""
    R = oracle.input_cost()
    assert np.allclose(np.diag(R)[:12], 5e-3) and np.abs(R[:12, 12:]).max() == 0
    q = _x0(params)[6:]
    J = oracle.rbd_qv(q, np.zeros(16))[2][:, 6:]
    assert np.abs(R[12:, 12:] - 2.0 * J.T @ J).max() < 1e-12
