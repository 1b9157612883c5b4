#!/usr/bin/env python3
"""How much of the WeightedWbc answer is the regularised-minimiser rule?  The headline workload's 4096 WBC
problems (policy of the first SQP iteration, the rbd states of the bench) solved with eps = 1e-8 (the rule) and with smaller /
larger eps: max and percentiles of |delta tau| (N m), |delta qdd|, |delta F|.  H = A_w'A_w has rank <= 18 of 38 (contact-force
weight 0), so the directions the cost does not see are fixed by eps alone; qpOASES regularises them with epsRegularisation =
5e3 * EPS ~ 1.1e-12 (its default), which is below the f64 resolution of H — and follows the regularised solve with ONE
regularisation step (setToMPC: numRegularisationSteps = 1), which hb_config.wbc_reg_steps reproduces on the final working set:
every comparison is made without (reg_steps 0, rounds 1-4) and with the step.  python tools/wbc_eps_sensitivity.py [--batch B]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from hunter_bipedal_control_amd import ingest, workload
from hunter_bipedal_control_amd.solver import HunterSolver

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=4096)
ap.add_argument("--nodes", type=int, default=100)
args = ap.parse_args()
P = ingest.load_packaged()
B, N = args.batch, args.nodes
sols = {}
for eps in (1e-6, 1e-8, 1e-10, 1e-12):
    for reg in (0, 1):
        s = HunterSolver(P, batch=B, max_nodes=N, wbc_eps_reg=eps, wbc_reg_steps=reg)
        try:
            w = workload.device_trot_batch(s, P, n_intervals=N)
            s.set_resident_inputs(w["x0"], w["t_now"], w["rbd"])
            s.step_resident()
            sol, status = s.get_wbc_solution()
            sols[(eps, reg)] = (sol, status, s.get_wbc_iterations())
        finally:
            s.close()


# the norm-scaled rule (hb_config.wbc_eps_mode = 1: eps = |A_w' A_w|_F * 1e3 * EPS per problem, qpOASES 3.2 regulariseHessian), with the step
s = HunterSolver(P, batch=B, max_nodes=N, wbc_eps_mode=1, wbc_reg_steps=1)
try:
    w = workload.device_trot_batch(s, P, n_intervals=N)
    s.set_resident_inputs(w["x0"], w["t_now"], w["rbd"])
    s.step_resident()
    sol, status = s.get_wbc_solution()
    sols[("norm", 1)] = (sol, status, s.get_wbc_iterations())
finally:
    s.close()


def cmp(a, b):
    (sa, sta, ita), (sb, stb, _) = sols[a], sols[b]
    ok = (sta == 0) & (stb == 0)
    d = np.abs(sa - sb)[ok]
    t = d[:, 28:].max(axis=1)
    return {"n_compared": int(ok.sum()), "status_histogram": [int((sta == k).sum()) for k in range(4)],
            "tau_max_Nm": float(t.max()), "tau_p50_Nm": float(np.percentile(t, 50)), "tau_p99_Nm": float(np.percentile(t, 99)),
            "n_above_1e-5_Nm": int((t > 1e-5).sum()), "n_above_1e-3_Nm": int((t > 1e-3).sum()), "n_above_0.1_Nm": int((t > 0.1).sum()),
            "qdd_max": float(d[:, :16].max()), "force_max_N": float(d[:, 16:28].max()), "active_set_iterations_max": int(ita.max())}


out = {"workload": f"{B} instances x N = {N}, first update of the headline workload (bench.py)",
       "rule": "eps = 1e-8 + ONE regularisation step on the final working set (hb_config.wbc_reg_steps = 1; qpOASES setToMPC)",
       "torque_movement_between_eps": {}}
for reg in (0, 1):
    blk = {}
    for a, b in ((1e-6, 1e-8), (1e-8, 1e-10), (1e-10, 1e-12)):
        blk[f"{a:g}_vs_{b:g}"] = cmp((a, reg), (b, reg))
    out["torque_movement_between_eps"][f"reg_steps_{reg}"] = blk
out["step_itself_at_1e-8 (reg 0 vs reg 1)"] = cmp((1e-8, 0), (1e-8, 1))
out["fixed_1e-8_vs_norm_scaled (wbc_eps_mode 0 vs 1, both with the step)"] = cmp((1e-8, 1), ("norm", 1))
out["fixed_1e-10_vs_norm_scaled"] = cmp((1e-10, 1), ("norm", 1))
# the instances whose answer still moves by more than 1e-3 N m between eps = 1e-8 and 1e-10 WITH the step: their reduced Hessian has an
# eigenvalue of the order of eps itself (tests/test_oracle_qp.py::test_regularisation_step_* shows it on the oracle), i.e. eps / lambda ~ 1
ref = sols[(1e-8, 1)][0]
mv = np.abs(sols[(1e-10, 1)][0] - ref)[:, 28:].max(axis=1)
big = np.where(mv > 1e-3)[0]
out["instances_moving_more_than_1e-3_Nm_with_the_step"] = [
    {"instance": int(i), "tau_move_Nm": float(mv[i]), "max_abs_qdd": float(np.abs(ref[i, :16]).max()),
     "active_set_iterations": int(sols[(1e-8, 1)][2][i])} for i in big[np.argsort(-mv[big])][:32]]
out["median_max_abs_qdd_all"] = float(np.median(np.abs(ref[:, :16]).max(axis=1)))
out["tau_scale_Nm"] = float(np.abs(ref[:, 28:]).max())
print(json.dumps(out, indent=1))
