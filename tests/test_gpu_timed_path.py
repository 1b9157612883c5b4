"""The code path bench.py's headline TIMES, under test: instance ranges on their own streams (hb_set_chunks), the cyclic
device-resident x0 sequence (hb_set_resident_x0_sequence) and the per-(range, slot) hipGraph replay of hb_step_resident —
including the invalidation of the captured graphs by a table update in the middle (graph_epoch) and their re-capture.

The kernels inside the graphs are the ones every other parity test runs; what this file pins is the capture / replay /
invalidation / per-slot x0 copy logic: a run on four graph-replayed ranges must be bit-identical to the same run on one
stream with direct launches, and one step of it must agree with the CPU oracle."""
import numpy as np
import pytest

from hunter_bipedal_control_amd import abi, workload
from bench import x0_sequence

pytestmark = pytest.mark.gpu

B, N, SLOTS = 256, 40, 8
STEPS_A, STEPS_B = 14, 16      # before / after the table update; + 1 final step that is compared with the oracle


def _run(params, chunks, hierarchical=False):
    from hunter_bipedal_control_amd.solver import HunterSolver
    s = HunterSolver(params, batch=B, max_nodes=N + 4, wbc_type=1 if hierarchical else 0)
    try:
        w = workload.device_trot_batch(s, params, n_intervals=N)
        s.set_resident_inputs(w["x0"], w["t_now"], w["rbd"])
        seq = x0_sequence(w["x0"], 3, n_seq=SLOTS)
        s.set_resident_x0_sequence(seq)
        s.set_chunks(chunks)
        for _ in range(STEPS_A):
            s.step_resident()
        c_mid = s.chunk_counters()
        mid = dict(xu=s.get_solution(), wbc=s.get_wbc_solution(), st=s.mpc_status(), perf=s.get_performance())
        # new tables 16 ms later: the captured graphs are stale (the iterate buffers swap for the warm start), the ranges fork again
        status = s.refgen_update(np.full(B, 0.1 + 0.016), w["horizon"], w["x0"], w["cmd"])
        assert status.max() == 0
        for _ in range(STEPS_B):
            s.step_resident()
        out = dict(mid=mid, c_mid=c_mid)
        if chunks == 1:   # the iterate the last step starts from (a getter joins the ranges: only taken on the one-stream run)
            out["before_last"] = s.get_solution()
        s.step_resident()
        out.update(xu=s.get_solution(), wbc=s.get_wbc_solution(), st=s.mpc_status(), perf=s.get_performance(), step=s.get_step(),
                   refs=s.get_references(), counters=s.chunk_counters(), seq=seq, w=w, iters=s.get_wbc_iterations())
        return out
    finally:
        s.close()


def _same(a, b):
    for k in ("xu", "wbc"):
        for p, q in zip(a[k], b[k]):
            assert np.array_equal(p, q), k
    assert np.array_equal(a["st"], b["st"]) and np.array_equal(a["perf"], b["perf"])


def test_graph_replayed_ranges_equal_one_stream_and_the_oracle(params, oracle):
    one = _run(params, 1)
    four = _run(params, 4)
    # --- the replay really happened, was invalidated by the table update, and was re-captured
    cm, c = four["c_mid"], four["counters"]
    assert cm["graph_launches"] == 4 * (STEPS_A - 3), cm      # fork step + two settling steps run direct, the rest replay
    assert cm["captures"] == 4 * SLOTS and cm["forks"] == 1, cm
    total = STEPS_A + STEPS_B + 1
    assert c["graph_launches"] == 4 * (total - 6), c
    assert c["captures"] == 2 * 4 * SLOTS and c["forks"] == 2, c          # every (range, slot) graph captured again after the update
    assert c["capture_failures"] == 0 and c["graphs_disabled"] == 0, c
    assert one["counters"]["graph_launches"] == 0 and one["counters"]["captures"] == 0
    # --- bit-identical results: iterate, WBC solution + status, MPC status, performance index — before and after the update
    _same(one["mid"], four["mid"])
    _same(one, four)
    assert np.array_equal(one["step"][0], four["step"][0]) and np.array_equal(one["step"][1], four["step"][1])
    assert np.array_equal(one["iters"], four["iters"])
    assert (four["st"] == abi.HB_INST_OK).all() and (four["wbc"][1] == 0).all()
    assert (four["perf"][:, 3] == 1.0).mean() > 0.9    # the headline's regime: (nearly) every line search takes the full step
    # --- the sequence slot of the last step against the CPU oracle: one SQP iteration from the iterate before it
    slot = (STEPS_A + STEPS_B) % SLOTS
    x, u = (a.copy() for a in one["before_last"])
    refs = four["refs"]
    po = oracle.mpc_solve(refs, np.ascontiguousarray(four["seq"][slot]), x, u, iters=1, threads=8)
    xd, ud = four["xu"]
    for i in range(B):
        n = int(refs["n_nodes"][i])
        assert np.abs(xd[i, :n + 1] - x[i, :n + 1]).max() < 1e-7 and np.abs(ud[i, :n] - u[i, :n]).max() < 1e-6, i
    assert np.array_equal(four["perf"][:, 3], po[:, 3])
    assert np.array_equal(xd[:, 0], four["seq"][slot])     # node 0 of every range is its slice of the slot's measured state


def test_graph_replayed_ranges_hierarchical_wbc(params):
    """The same equality with HierarchicalWbc in the captured step (k_hwbc takes dynamic LDS: a different launch node)."""
    one = _run(params, 1, hierarchical=True)
    four = _run(params, 4, hierarchical=True)
    assert four["counters"]["graph_launches"] > 0 and four["counters"]["capture_failures"] == 0
    _same(one["mid"], four["mid"])
    _same(one, four)


def test_two_ranges_uneven_split_and_sequence_longer_than_batch_steps(params):
    """Ranges that do not divide the batch (B = 100 on 3 ranges: 34 + 34 + 32) and a slot count that is not a divisor of the step
    count: every slot's x0 lands on the right instances."""
    from hunter_bipedal_control_amd.solver import HunterSolver
    Bs, Ns = 100, 24
    res = []
    for chunks in (1, 3):
        s = HunterSolver(params, batch=Bs, max_nodes=Ns + 4)
        try:
            w = workload.device_trot_batch(s, params, n_intervals=Ns)
            s.set_resident_inputs(w["x0"], w["t_now"], w["rbd"])
            seq = x0_sequence(w["x0"], 11, n_seq=5)
            s.set_resident_x0_sequence(seq)
            s.set_chunks(chunks)
            for _ in range(13):
                s.step_resident()
            res.append((s.get_solution(), s.get_wbc_solution(), s.chunk_counters()))
        finally:
            s.close()
    (xa, ua), (sa, sta), _ = res[0]
    (xb, ub), (sb, stb), cb = res[1]
    assert cb["graph_launches"] == 3 * 10 and cb["capture_failures"] == 0
    assert np.array_equal(xa, xb) and np.array_equal(ua, ub) and np.array_equal(sa, sb) and np.array_equal(sta, stb)
    assert np.array_equal(xb[:, 0], seq[12 % 5])
