"""Closed-loop rollouts.  The controller is only right if the robot stays up: the standing and
trotting rollouts below exercise reference generation, SQP, policy evaluation, WBC and the joint command law together
against the contact-consistent plant stub of oracle/plant.py."""
import numpy as np
import pytest

from closed_loop_oracle import OracleLoop
from oracle.plant import Plant as _Plant


def test_oracle_closed_loop_stands(params, oracle):
    """CPU: oracle controller + host reference manager keep the robot standing (config 1 shape, reference horizon)."""
    loop = OracleLoop(oracle, params, "stance", (0.0, 0.0, 0.0, 0.0))
    for _ in range(300):      # 0.6 s
        q, v = loop.step()
    assert np.isfinite(q).all()
    assert abs(q[0, 2] - 0.62) < 0.02 and np.abs(q[0, 0:2]).max() < 0.03
    assert np.abs(q[0, 3:6]).max() < 0.06 and np.abs(v[0]).max() < 0.6
    assert loop.last["status"][0] == 0


@pytest.mark.gpu
def test_device_closed_loop_matches_oracle_loop_then_trots(params, oracle):
    """GPU: the device loop reproduces the oracle loop tick by tick at first (same plant, independent controllers), and a
    batch with different commands trots for two seconds without falling, advancing at about the commanded speed."""
    from hunter_bipedal_control_amd.rollout import DeviceLoop
    from hunter_bipedal_control_amd.solver import HunterSolver
    B = 4
    cmds = np.array([[0.2, 0.0, 0.0, 0.0], [0.3, 0.0, 0.0, 0.0], [0.15, 0.08, 0.0, 0.0], [0.2, 0.0, 0.0, 0.25]])
    gaits = ["trot"] * B
    s = HunterSolver(params, batch=B, max_nodes=108)
    try:
        dev = DeviceLoop(s, params, gaits, cmds, plant_factory=_Plant)
        twin = OracleLoop(oracle, params, "trot", cmds[0])
        for k in range(200):                                   # 0.4 s: stance, then the first swing phase
            qd_, vd_ = dev.step()
            qo_, vo_ = twin.step()
            assert np.abs(qd_[0] - qo_[0]).max() < 1e-6 and np.abs(vd_[0] - vo_[0]).max() < 1e-4, k
        assert np.abs(dev.last["cmd"]["torque"][0] - twin.last["torque"]).max() < 1e-3
        for k in range(800):                                   # up to 2.0 s
            q, v = dev.step()
        assert dev.last["out"]["status"].max() == 0
    finally:
        s.close()
    assert np.isfinite(q).all()
    assert (np.abs(q[:, 2] - 0.63) < 0.04).all(), q[:, 2]              # still at walking height
    assert np.abs(q[:, 4:6]).max() < 0.15                               # pitch / roll stay small
    travelled = q[:, 0:2] - 0.0
    expect = cmds[:, 0:2] * (2.0 - 0.3)                                 # the gait starts at t = 0.3 s
    straight = [0, 1, 2]                                                # instance 3 turns: compare its path length only
    assert np.abs(travelled[straight] - expect[straight]).max() < 0.12, (travelled, expect)
    assert abs(q[3, 3] - 0.25 * 1.7) < 0.2                              # yaw follows the commanded rate


@pytest.mark.gpu
def test_device_closed_loop_with_the_state_estimator_in_the_loop(params):
    """estimate -> references -> MPC -> WBC -> joint command with hb_estimator_update as the only source of the state
    (ideal IMU + encoders + commanded contacts from the plant): the robot trots as with the true state, and the filter
    tracks the plant's base position / velocity."""
    from hunter_bipedal_control_amd.rollout import DeviceLoop
    from hunter_bipedal_control_amd.solver import HunterSolver
    B = 2
    cmds = np.array([[0.2, 0.0, 0.0, 0.0], [0.0, 0.0, 0.0, 0.0]])
    s = HunterSolver(params, batch=B, max_nodes=108)
    try:
        dev = DeviceLoop(s, params, ["trot", "stance"], cmds, use_estimator=True, plant_factory=_Plant)
        err_p, err_v = 0.0, 0.0
        for k in range(750):                                   # 1.5 s
            q, v = dev.step()
            if k > 100:
                xh, _ = s.estimator_filter()
                err_p = max(err_p, np.abs(xh[:, 0:3] - q[:, 0:3]).max())
                err_v = max(err_v, np.abs(xh[:, 3:6] - v[:, 0:3]).max())
        assert dev.last["out"]["status"].max() == 0
    finally:
        s.close()
    assert np.isfinite(q).all() and (np.abs(q[:, 2] - 0.63) < 0.04).all() and np.abs(q[:, 4:6]).max() < 0.15
    assert abs(q[0, 0] - 0.2 * 1.2) < 0.1 and abs(q[1, 0]) < 0.05
    assert err_p < 0.03 and err_v < 0.15, (err_p, err_v)


@pytest.mark.gpu
def test_closed_loop_over_the_lcm_wire_format_equals_the_array_loop(params):
    """The same estimator-in-the-loop rollout with sensors and commands crossing the controller boundary as LCM wire
    images (low_state_t in through hb_estimator_update_lcm, low_cmd_t out through hb_joint_command_lcm, the plant side
    applying the PD + feed-forward law of the decoded command like the reference's MuJoCo bridge).  The codec is exact, so
    the two loops may only differ by the rounding of the torque law evaluated on the host instead of the device."""
    from hunter_bipedal_control_amd.rollout import DeviceLoop
    from hunter_bipedal_control_amd.solver import HunterSolver
    B = 2
    cmds = np.array([[0.2, 0.0, 0.0, 0.0], [0.0, 0.0, 0.0, 0.1]])
    traj = {}
    for use_lcm in (False, True):
        s = HunterSolver(params, batch=B, max_nodes=108)
        try:
            dev = DeviceLoop(s, params, ["trot", "stance"], cmds, use_estimator=True, use_lcm=use_lcm, plant_factory=_Plant)
            qs = []
            for k in range(200):                               # 0.4 s: stance, then the first swing phase
                q, v = dev.step()
                qs.append(np.concatenate([q, v], axis=1).copy())
            assert dev.last["out"]["status"].max() == 0
            traj[use_lcm] = np.array(qs)
        finally:
            s.close()
    assert np.isfinite(traj[True]).all()
    assert np.abs(traj[True][:100] - traj[False][:100]).max() < 1e-9
    assert np.abs(traj[True] - traj[False]).max() < 1e-6


@pytest.mark.gpu
def test_device_plant_matches_numpy_plant_and_resident_loop_trots(params):
    """hb_plant_step vs plant.py on the same torque sequence; then the fully device-resident loop (plant included) against
    the loop with the host-side plant."""
    from oracle.plant import Plant
    from hunter_bipedal_control_amd.rollout import DeviceLoop, ResidentLoop, standing_configuration
    from hunter_bipedal_control_amd.solver import HunterSolver
    B = 4
    rng = np.random.default_rng(3)
    s = HunterSolver(params, batch=B, max_nodes=108)
    try:
        q0 = standing_configuration(params, B, s)
        q0[:, 3:6] = 0.03 * rng.standard_normal((B, 3))
        v0 = 0.05 * rng.standard_normal((B, 16))
        zeros_u = np.zeros((B, 22))

        def foot_fn(q):
            x = np.zeros((q.shape[0], 22))
            x[:, 6:9], x[:, 9:12], x[:, 12:] = q[:, 0:3], q[:, 3:6], q[:, 6:]
            return s.eval_foot_kinematics(x, zeros_u[:q.shape[0]])[0]

        host = Plant(lambda rbd: s.eval_rbd(rbd), foot_fn, q0.copy(), v0.copy())
        s.plant_reset(q0, v0)
        for tick in range(10):
            contact = np.tile([1, 1, 1, 1] if tick < 4 else [0, 1, 0, 1], (B, 1)).astype(np.int32)
            tau = 3.0 * rng.standard_normal((B, 10))
            host.step(tau, contact.astype(bool), 0.002, substeps=4)
            s.plant_step(tau, contact, 0.002, 4)
            st = s.plant_state()
            assert np.abs(st["q"] - host.q).max() < 1e-10 and np.abs(st["v"] - host.v).max() < 1e-8, tick
            assert np.abs(st["rbd"] - host.rbd()).max() < 1e-8
    finally:
        s.close()
    cmds = np.array([[0.2, 0.0, 0.0, 0.0], [0.3, 0.0, 0.0, 0.0], [0.15, 0.08, 0.0, 0.0], [0.0, 0.0, 0.0, 0.0]])
    gaits = ["trot", "trot", "trot", "stance"]
    finals = []
    for resident in (True, False):
        s = HunterSolver(params, batch=B, max_nodes=108)
        try:
            loop = ResidentLoop(s, params, gaits, cmds) if resident else DeviceLoop(s, params, gaits, cmds, plant_factory=_Plant)
            for k in range(600):                               # 1.2 s
                loop.step()
            finals.append(s.plant_state()["q"] if resident else loop.plant.q.copy())
        finally:
            s.close()
    q_res, q_host = finals
    assert np.isfinite(q_res).all() and (np.abs(q_res[:, 2] - 0.63) < 0.04).all() and np.abs(q_res[:, 4:6]).max() < 0.15
    assert np.abs(q_res[:3, 0] - cmds[:3, 0] * 0.9).max() < 0.1 and abs(q_res[3, 0]) < 0.05
    assert np.abs(q_res - q_host).max() < 5e-3          # same loop, plant on the device vs on the host
