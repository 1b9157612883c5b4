"""-m gpu parity of the batched state estimator through the C ABI against oracle/estimator.hpp."""
import numpy as np
import pytest

from hunter_bipedal_control_amd import abi
from oracle import refgen, workloads

pytestmark = pytest.mark.gpu


def _quat_from_zyx(zyx):
    R = refgen.zyx_to_rotation(zyx)
    w = 0.5 * np.sqrt(1 + np.trace(R))
    return np.array([(R[2, 1] - R[1, 2]) / (4 * w), (R[0, 2] - R[2, 0]) / (4 * w), (R[1, 0] - R[0, 1]) / (4 * w), w])


def _sensor_stream(params, B, ticks, seed):
    rng = np.random.default_rng(seed)
    qj0 = np.array(params["config"]["default_joint_state"])
    for k in range(ticks):
        zyx = np.stack([[2.7 + 0.04 * k + 0.01 * i, 0.1 * rng.standard_normal(), 0.1 * rng.standard_normal()] for i in range(B)])
        yield dict(quat=np.stack([_quat_from_zyx(z) for z in zyx]), w=0.5 * rng.standard_normal((B, 3)),
                   a=np.array([0, 0, 9.81]) + 0.5 * rng.standard_normal((B, 3)), qj=qj0 + 0.1 * rng.standard_normal((B, 10)),
                   qdj=rng.standard_normal((B, 10)), contact=(rng.uniform(size=(B, 4)) < 0.7).astype(np.int32))


def test_estimator_matches_oracle_over_a_sensor_stream(params, oracle):
    from hunter_bipedal_control_amd.solver import HunterSolver
    B = 48
    ecfg = abi.make_estimator_config(params)
    st = oracle.kf_init(B)
    s = HunterSolver(params, batch=B, max_nodes=4)
    try:
        s.estimator_reset(ecfg)
        for k, m in enumerate(_sensor_stream(params, B, 25, 7)):
            rbd_o, x_o = oracle.kf_update(ecfg, st, 0.002, m["quat"], m["w"], m["a"], m["qj"], m["qdj"], m["contact"])
            rbd_g, x_g = s.estimator_update(0.002, m["quat"], m["w"], m["a"], m["qj"], m["qdj"], m["contact"])
            assert np.abs(rbd_g - rbd_o).max() < 1e-10, k
            assert np.abs(x_g - x_o).max() < 1e-10, k          # includes the yaw unwrapped across +pi
        xh, P = s.estimator_filter()
    finally:
        s.close()
    assert np.abs(xh - st["xhat"]).max() < 1e-10
    assert np.abs(P - st["P"]).max() < 1e-9 * max(1.0, np.abs(st["P"]).max())
    assert np.abs(P - P.transpose(0, 2, 1)).max() == 0.0     # symmetrised exactly, like the reference's (p + p') / 2


def test_estimator_standing_known_answer_and_custom_initial_state(params):
    """SURVEY.md §8c FK known answer: default stance, all contacts -> base height 0.6286 + footRadius."""
    from hunter_bipedal_control_amd.solver import HunterSolver
    B = 3
    ecfg = abi.make_estimator_config(params)
    qj = np.tile(params["config"]["default_joint_state"], (B, 1))
    s = HunterSolver(params, batch=B, max_nodes=4)
    try:
        x0 = np.zeros((B, 18)); x0[:, 2] = [0.0, 0.5, 1.0]
        s.estimator_reset(ecfg, x0)
        xh, P = s.estimator_filter()
        assert np.array_equal(xh, x0) and np.array_equal(P[1], 100.0 * np.eye(18))
        for _ in range(3000):
            rbd, x = s.estimator_update(0.002, np.tile([0, 0, 0, 1.0], (B, 1)), np.zeros((B, 3)), np.tile([0, 0, 9.81], (B, 1)), qj,
                                        np.zeros((B, 10)), np.ones((B, 4), dtype=np.int32))
    finally:
        s.close()
    assert np.abs(rbd[:, 5] - (0.5 * (0.6285 + 0.6287) + 0.02)).max() < 3e-4
    assert np.abs(x[:, 0:6]).max() < 1e-6 and np.abs(rbd[:, 19:22]).max() < 1e-6


def test_estimate_feeds_mpc_and_wbc_without_leaving_the_device(params, oracle):
    """to_resident = 1: the estimator's outputs become the resident inputs of hb_step_resident; the result equals feeding
    the same estimates through the host-pointer entry points."""
    from hunter_bipedal_control_amd.solver import HunterSolver
    B, N = 8, 30
    refs, x0, rbd0, t_now = workloads.trot_batch(params, B, n_intervals=N)
    nmax = refs["mode"].shape[1]
    ecfg = abi.make_estimator_config(params)
    rng = np.random.default_rng(2)
    # sensors consistent with the workload's states: orientation + joints from x0, at rest
    quat = np.stack([_quat_from_zyx(x0[i, 9:12]) for i in range(B)])
    qj, qdj = x0[:, 12:], 0.05 * rng.standard_normal((B, 10))
    xh0 = np.zeros((B, 18)); xh0[:, 0:3] = x0[:, 6:9]
    contact = np.ones((B, 4), dtype=np.int32)
    a = np.stack([refgen.zyx_to_rotation(x0[i, 9:12]).T @ [0, 0, 9.81] for i in range(B)])
    outs = []
    for resident in (True, False):
        s = HunterSolver(params, batch=B, max_nodes=nmax)
        try:
            s.set_references(refs)
            s.estimator_reset(ecfg, xh0)
            s.set_resident_inputs(x0, t_now, rbd0)
            rbd_e, x_e = s.estimator_update(0.002, quat, np.zeros((B, 3)), a, qj, qdj, contact, to_resident=resident)
            s.reset(x_e)
            if resident:
                s.step_resident()
                sol, status = s.get_wbc_solution()
            else:
                s.mpc_solve(x_e)
                s.publish()
                out = s.wbc_update(t_now, rbd_e)
                sol, status = out["sol"], out["status"]
            xs, us = s.get_solution()
        finally:
            s.close()
        outs.append((sol, status, xs, us, x_e))
    assert np.abs(outs[0][4][:, 6:9] - x0[:, 6:9]).max() < 0.05   # the filter keeps the supplied position prior
    for a_, b_ in zip(outs[0][:4], outs[1][:4]):
        assert np.array_equal(a_, b_)
    assert outs[0][1].max() == 0
