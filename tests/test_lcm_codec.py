"""LCM wire format of the reference's low-level messages (include/hunter_lcm.h).

tests/golden/ref_lcm.json holds bytes produced by the reference's own lcm-gen generated classes
(lcm_msg/include/lcm_msg/*.hpp compiled in place into oracle/_ref, fixture written by tests/golden/make_ref_lcm.py) and
the hash constants of those headers.  CPU: fingerprints, host codec (bit-exact both ways), UDP framing.  GPU: the device
packers behind hb_joint_command_lcm / hb_estimator_update_lcm against the host codec and the array entry points
(bit-exact: byte / bit-pattern work).
"""
import ctypes as C
import json
from pathlib import Path

import numpy as np
import pytest

from hunter_bipedal_control_amd import solver

HERE = Path(__file__).resolve().parent
TYPES = {"low_cmd_t": 0, "low_state_t": 1, "full_state_t": 2}


@pytest.fixture(scope="module")
def golden():
    return json.loads((HERE / "golden/ref_lcm.json").read_text())


@pytest.fixture(scope="module")
def lib():
    lib = solver.load_library()
    lib.hb_lcm_fingerprint.restype = C.c_uint64
    return lib


def _u8(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def test_fingerprints_match_the_reference_headers(golden, lib):
    for t in golden["types"]:
        code = TYPES[t["name"]]
        base = int(t["hash_constant_in_header"], 16)
        rot = ((base << 1) & (2**64 - 1)) + (base >> 63)
        assert rot == int(t["fingerprint"], 16)                      # low_cmd_t.hpp:188-192 as compiled
        assert lib.hb_lcm_fingerprint(code) == rot                   # lcm-gen hash recomputed from the member list
        assert lib.hb_lcm_encoded_size(code) == t["encoded_size"] and lib.hb_lcm_field_count(code) == t["n_fields"]
    assert lib.hb_lcm_fingerprint(7) == 0 and lib.hb_lcm_encoded_size(-1) < 0


def test_host_codec_is_bit_exact_against_the_reference_classes(golden, lib):
    for t in golden["types"]:
        code, nf, sz = TYPES[t["name"]], t["n_fields"], t["encoded_size"]
        n = len(t["messages"])
        ts = np.array([m["timestamp"] for m in t["messages"]], dtype=np.int64)
        fields = np.frombuffer(bytes.fromhex("".join(m["fields_hex"] for m in t["messages"])), dtype=np.float64).copy().reshape(n, nf)
        ref = np.frombuffer(bytes.fromhex("".join(m["bytes_hex"] for m in t["messages"])), dtype=np.uint8).copy().reshape(n, sz)
        out = np.zeros((n, sz), dtype=np.uint8)
        assert lib.hb_lcm_encode(code, n, ts.ctypes.data_as(C.c_void_p), fields.ctypes.data_as(C.c_void_p), _u8(out)) == 0
        assert np.array_equal(out, ref), t["name"]
        ts2, f2 = np.zeros(n, dtype=np.int64), np.zeros((n, nf))
        assert lib.hb_lcm_decode(code, n, _u8(ref), ts2.ctypes.data_as(C.c_void_p), f2.ctypes.data_as(C.c_void_p)) == 0
        assert np.array_equal(ts2, ts) and f2.tobytes() == fields.tobytes()      # bit patterns incl. -0.0
        # a foreign fingerprint is rejected like the generated decode() does (returns -1)
        bad = ref.copy()
        bad[n - 1, 3] ^= 0x40
        assert lib.hb_lcm_decode(code, n, _u8(bad), ts2.ctypes.data_as(C.c_void_p), f2.ctypes.data_as(C.c_void_p)) == -1
        # empty batch and argument errors
        assert lib.hb_lcm_encode(code, 0, ts.ctypes.data_as(C.c_void_p), fields.ctypes.data_as(C.c_void_p), _u8(out)) == 0
        assert lib.hb_lcm_encode(code, n, None, fields.ctypes.data_as(C.c_void_p), _u8(out)) == -1


def test_udp_short_message_framing(lib):
    payload = np.arange(40, dtype=np.uint8)
    out = np.zeros(128, dtype=np.uint8)
    n = lib.hb_lcm_frame(b"LOWCMD", C.c_uint32(0x01020304), _u8(payload), 40, _u8(out), 128)
    assert n == 8 + 7 + 40
    assert bytes(out[:4]) == b"LC02" and bytes(out[4:8]) == bytes([1, 2, 3, 4]) and bytes(out[8:15]) == b"LOWCMD\0"
    assert np.array_equal(out[15:55], payload)
    assert lib.hb_lcm_frame(b"LOWCMD", C.c_uint32(0), _u8(payload), 40, _u8(out), 50) == -1  # does not fit


@pytest.mark.gpu
def test_device_packers_match_the_host_codec_and_the_array_entry_points(params, golden, lib):
    from oracle import workloads
    from hunter_bipedal_control_amd import abi
    from hunter_bipedal_control_amd.solver import HunterSolver
    B, N = 64, 20
    refs, x0, rbd, tn = workloads.trot_batch(params, B, n_intervals=N)
    s = HunterSolver(params, batch=B, max_nodes=N)
    s.set_references(refs); s.reset(x0); s.set_resident_inputs(x0, tn, rbd)
    s.step_resident()
    # --- command side: hb_joint_command_lcm == hb_lcm_encode(hb_joint_command outputs), byte for byte
    gains = abi.make_joint_gains()
    jc = s.joint_command(gains, 0.002)
    ts_ns = 1726000000123456789
    wire = s.joint_command_lcm(gains, 0.002, ts_ns)
    fields = np.concatenate([jc["pos_des"], jc["vel_des"], np.zeros((B, 10)), jc["tau_ff"], jc["kp"], jc["kd"]], axis=1)
    ref = np.zeros((B, 496), dtype=np.uint8)
    ts = np.full(B, ts_ns, dtype=np.int64)
    assert lib.hb_lcm_encode(0, B, ts.ctypes.data_as(C.c_void_p), np.ascontiguousarray(fields).ctypes.data_as(C.c_void_p), _u8(ref)) == 0
    assert np.array_equal(wire, ref)
    # --- state side: hb_estimator_update_lcm == hb_estimator_update on the unpacked arrays, bit for bit
    rng = np.random.default_rng(5)
    quat_xyzw = rng.normal(size=(B, 4)); quat_xyzw /= np.linalg.norm(quat_xyzw, axis=1, keepdims=True)
    gyro, acc = 0.1 * rng.normal(size=(B, 3)), np.array([0, 0, 9.81]) + 0.1 * rng.normal(size=(B, 3))
    qj, qdj, tau = rbd[:, 6:16] + 0.01 * rng.normal(size=(B, 10)), 0.1 * rng.normal(size=(B, 10)), rng.normal(size=(B, 10))
    contact = np.ones((B, 4), dtype=np.int32)
    st_fields = np.concatenate([quat_xyzw[:, 3:4], quat_xyzw[:, 0:3], gyro, acc, qj, qdj, tau], axis=1)  # (w x y z) on the wire
    st_ts = np.arange(B, dtype=np.int64) + 10**15
    st_wire = np.zeros((B, 336), dtype=np.uint8)
    assert lib.hb_lcm_encode(1, B, st_ts.ctypes.data_as(C.c_void_p), np.ascontiguousarray(st_fields).ctypes.data_as(C.c_void_p), _u8(st_wire)) == 0
    ecfg = abi.make_estimator_config(params)
    s.estimator_reset(ecfg)
    rbd_a, x_a = s.estimator_update(0.002, quat_xyzw, gyro, acc, qj, qdj, contact)
    s.estimator_reset(ecfg)
    rbd_b, x_b, ts_b = s.estimator_update_lcm(0.002, st_wire, contact)
    assert np.array_equal(ts_b, st_ts)
    assert rbd_a.tobytes() == rbd_b.tobytes() and x_a.tobytes() == x_b.tobytes()
    # a foreign fingerprint is refused and leaves the filter untouched
    xh0, P0 = s.estimator_filter()
    bad = st_wire.copy(); bad[17, 0] ^= 1
    with pytest.raises(solver.HunterHipError):
        s.estimator_update_lcm(0.002, bad, contact)
    xh1, P1 = s.estimator_filter()
    assert xh0.tobytes() == xh1.tobytes() and P0.tobytes() == P1.tobytes()
    s.close()
