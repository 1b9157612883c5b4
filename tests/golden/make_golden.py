#!/usr/bin/env python3
"""Generates tests/golden/*.json|npz.  Run in the build container (needs /root/reference for part 1).

Part 1 — reference_known_answers.json: the few known answers the reference itself holds for this path
(SURVEY.md §8c): total mass of hunter.urdf, weight-compensating stance force (utils.h:82-83), the swing planner's
feet biases (task.info:28-31), the default-stance contact positions computed with numpy straight from the URDF joint
origins/axes (independent of the oracle), and samples of the relaxed log-barrier evaluated with the formula of
legged_interface/src/constraint/design_tools/relaxedBarrierPenaltyVis.py:15-19 (restated below; the script itself
needs matplotlib).

Part 2 — oracle_regression.npz: outputs of the CPU oracle on seeded inputs ("oracle-generated", NOT reference
outputs — the reference cannot be built here): freezes the oracle so that drift is caught on both CPU and GPU boxes.
"""
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from hunter_bipedal_control_amd import ingest, workload  # noqa: E402
from oracle import refgen, workloads

OUT = Path(__file__).parent


def barrier_reference_formula(mu, delta, h):
    # relaxedBarrierPenaltyVis.py:15-19:  -mu ln h  (h > delta);  mu/2 (((h-2 delta)/delta)^2 - 1) - mu ln(delta)
    if h > delta:
        return -mu * np.log(h)
    return mu / 2.0 * (((h - 2 * delta) / delta) ** 2 - 1) - mu * np.log(delta)


def part1():
    ref = Path("/root/reference")
    if ref.exists():
        cfgdir = ref / "legged_controllers/config/hunter"
        model = ingest.read_urdf(ref / "legged_examples/legged_hunter/legged_hunter_description/urdf/hunter.urdf")
        cfg = ingest.read_config(cfgdir / "task.info", cfgdir / "reference.info", cfgdir / "gait.info")
    else:
        p = ingest.load_packaged()
        model, cfg = p["model"], p["config"]
    x = np.array(cfg["initial_state"], dtype=float)
    x[6:9] = 0.0
    feet = refgen.foot_positions(model, x)  # plain numpy FK from the URDF numbers
    m = float(sum(model["mass"]))
    samples = []
    for mu, delta in ((0.1, 5.0), (1.0, 0.1), (0.1, 1.0)):
        for h in (-1.0, 0.0, 0.5 * delta, delta, 1.5 * delta, 10 * delta):
            samples.append([mu, delta, h, float(barrier_reference_formula(mu, delta, h))])
    out = dict(total_mass=m, stance_fz_per_contact=m * 9.81 / 4, default_stance_feet=feet.tolist(),
               feet_bias_x1=cfg["swing"]["feet_bias_x1"], feet_bias_x2=cfg["swing"]["feet_bias_x2"],
               feet_bias_z=cfg["swing"]["feet_bias_z"], relaxed_barrier_samples=samples,
               joint_lower=model["q_lower"], joint_upper=model["q_upper"])
    (OUT / "reference_known_answers.json").write_text(json.dumps(out, indent=1))


def part2():
    from oracle.pyoracle import Oracle
    params = ingest.load_packaged()
    o = Oracle(params)
    rng = np.random.default_rng(2026)
    x0 = np.array(params["config"]["initial_state"])
    x = x0 + 0.2 * rng.standard_normal((6, 22))
    u = rng.standard_normal((6, 22)) * np.r_[np.full(12, 20.0), np.full(10, 1.0)]
    f, A, B = o.flow_map(x, u, jac=True)
    pos, vel = o.foot_kinematics(x, u)
    refs, xs0, rbd, t_now = workloads.trot_batch(params, 2, n_intervals=30, cmd_vel=(0.3, 0.0, 0.0, 0.1))
    xt = np.zeros((2, 31, 22)); ut = np.zeros((2, 30, 22))
    for i in range(2):
        xt[i], ut[i] = o.cold_start(refs["mode"][i], xs0[i])
    perf = []
    for _ in range(3):
        perf.append(o.mpc_solve(refs, xs0, xt, ut, iters=1))
    mode = np.array([3, 3, 2, 1, 0, 2], dtype=np.int32)
    stance = np.array([1, 0, 0, 0, 0, 0], dtype=np.int32)
    rb = np.stack([workload.rbd_from_state(x0 + 0.03 * rng.standard_normal(22), i) for i in range(6)])
    rb[:, 16:] = 0.3 * rng.standard_normal((6, 16))
    ud = np.zeros((6, 22))
    m = sum(params["model"]["mass"])
    for i in range(6):
        cf = refgen.mode_to_contact_flags(int(mode[i]))
        for k in range(4):
            if cf[k]:
                ud[i, 3 * k + 2] = m * 9.81 / sum(cf)
    xd = x0 + 0.05 * rng.standard_normal((6, 22))
    sol, st, it = o.wbc_update(xd, ud, rb, mode, stance_flag=stance)
    np.savez(OUT / "oracle_regression.npz", x=x, u=u, f=f, A=A, B=B, pos=pos, vel=vel, mpc_x0=xs0, mpc_x=xt, mpc_u=ut,
             mpc_perf=np.array(perf), **{"ref_" + k: v for k, v in refs.items()}, wbc_xd=xd, wbc_ud=ud, wbc_rbd=rb, wbc_mode=mode,
             wbc_stance=stance, wbc_sol=sol, wbc_status=st)


if __name__ == "__main__":
    part1()
    part2()
    print("golden fixtures written to", OUT)
