"""Time of the WBC kernels at the bench batch size — WeightedWbc (wbc_type 0) and HierarchicalWbc (wbc_type 1, config 5) with
its ablation stops; run on the GPU box:  python tools/bench_hwbc.py"""
import sys
sys.path.insert(0, ".")
import numpy as np
import bench
from hunter_bipedal_control_amd import ingest, workload
from hunter_bipedal_control_amd.solver import HunterSolver
P = ingest.load_packaged()
B, N = 4096, 100
for wt, stop in ((0, 0), (0, 13), (0, 14), (0, 15), (0, 11), (0, 12), (1, 41), (1, 43), (1, 44), (1, 42), (1, 0)):
    s = HunterSolver(P, batch=B, max_nodes=N, wbc_type=wt, reserved=stop)
    w = workload.device_trot_batch(s, P, n_intervals=N)
    s.set_resident_inputs(w["x0"], w["t_now"], w["rbd"])
    s.set_resident_x0_sequence(bench.x0_sequence(w["x0"], 0))
    ms = []
    for it in range(5):
        s.step_resident(); s.sync(); ms.append(s.stats()["ms_wbc"])
    sol, st = s.get_wbc_solution()
    print("wbc_type", wt, "stop", stop, "ms_wbc", np.round(ms, 3), "status hist", np.bincount(st, minlength=4))
    s.close()
