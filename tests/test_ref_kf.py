"""The state estimator held to the REFERENCE's own compiled code.

tests/golden/ref_kf.npz was written by tests/golden/make_ref_kf.py from oracle/_ref/libref_kf.so = the reference's
legged_estimation/src/{LinearKalmanFilter, StateEstimateBase}.cpp compiled in place (foot kinematics fed from the oracle,
noise settings read by the reference's loadSettings from the reference's task.info).  Pinned: updateImu (quatToZyx,
local -> global angular velocity), the rbdState packing, the 18-state / 28-measurement predict + correct with the contact-dependent
noise schedule, the (P + P') / 2 symmetrisation and the covariance reset rule (LinearKalmanFilter.cpp:72-184).
CPU: the oracle (oracle/estimator.hpp) against the vectors; -m gpu: the device kernel k_estimator through the C-ABI."""
from pathlib import Path

import numpy as np
import pytest

from hunter_bipedal_control_amd import abi

GOLD = Path(__file__).parent / "golden" / "ref_kf.npz"
KEYS = ("quat", "w", "a", "qj", "qdj", "contact")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def test_packaged_estimator_settings_are_the_reference_files(gold, params):
    """[footRadius, imuProcessNoisePosition, imuProcessNoiseVelocity, footProcessNoisePosition, footSensorNoisePosition,
    footSensorNoiseVelocity, footHeightSensorNoise] as the reference's loadSettings read them from task.info."""
    k = params["config"]["kalman"]
    mine = [k["foot_radius"], k["imu_process_noise_position"], k["imu_process_noise_velocity"], k["foot_process_noise_position"],
            k["foot_sensor_noise_position"], k["foot_sensor_noise_velocity"], k["foot_height_sensor_noise"]]
    assert np.array_equal(np.array(mine), gold["settings"])


def test_oracle_filter_matches_reference_filter(gold, params, oracle):
    ecfg = abi.make_estimator_config(params)
    dt = float(gold["dt"])
    for name in gold["streams"]:
        st = oracle.kf_init(1)
        for k in range(len(gold[f"{name}_quat"])):
            rbd, _ = oracle.kf_update(ecfg, st, dt, *[gold[f"{name}_{key}"][k] for key in KEYS])
            assert np.abs(rbd[0] - gold[f"{name}_out_rbd"][k]).max() < 1e-12, (name, k)
            assert np.abs(rbd[0, 0:3] - gold[f"{name}_out_zyx"][k]).max() < 1e-14
            assert np.abs(st["xhat"][0] - gold[f"{name}_out_xhat"][k]).max() < 1e-12, (name, k)
            Pr = gold[f"{name}_out_P"][k]
            assert np.abs(st["P"][0] - Pr).max() < 1e-12 * max(1.0, np.abs(Pr).max()), (name, k)


@pytest.mark.gpu
def test_device_filter_matches_reference_filter(gold, params):
    from hunter_bipedal_control_amd.solver import HunterSolver
    names = [str(n) for n in gold["streams"]]
    ticks = min(len(gold[f"{n}_quat"]) for n in names)
    B = len(names)
    ecfg = abi.make_estimator_config(params)
    s = HunterSolver(params, batch=B, max_nodes=4)
    try:
        s.estimator_reset(ecfg)
        for k in range(ticks):
            args = [np.stack([gold[f"{n}_{key}"][k] for n in names]) for key in KEYS]
            rbd, _ = s.estimator_update(float(gold["dt"]), *args)
            xh, P = s.estimator_filter()
            for i, n in enumerate(names):
                assert np.abs(rbd[i] - gold[f"{n}_out_rbd"][k]).max() < 1e-10, (n, k)
                assert np.abs(xh[i] - gold[f"{n}_out_xhat"][k]).max() < 1e-10, (n, k)
                Pr = gold[f"{n}_out_P"][k]
                assert np.abs(P[i] - Pr).max() < 1e-9 * max(1.0, np.abs(Pr).max()), (n, k)
    finally:
        s.close()
