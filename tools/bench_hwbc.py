"""Time of the HierarchicalWbc kernel at the bench batch size (config 5 uses it); run on the GPU box."""
import sys
sys.path.insert(0, '.')
import numpy as np
import bench
from hunter_bipedal_control_amd import ingest
from hunter_bipedal_control_amd.solver import HunterSolver
P = ingest.load_packaged()
B, N = 4096, 100
refs, x0, rbd, tn = bench.make_batch(P, B, N, 0)
for wt, stop in ((0, 0), (1, 41), (1, 42), (1, 0)):
    s = HunterSolver(P, batch=B, max_nodes=N, wbc_type=wt, reserved=stop)
    s.set_references(refs); s.reset(x0); s.set_resident_inputs(x0, tn, rbd)
    ms = []
    for it in range(5):
        s.step_resident(); s.sync(); ms.append(s.stats()["ms_wbc"])
    sol, st = s.get_wbc_solution()
    print("wbc_type", wt, "stop", stop, "ms_wbc", np.round(ms, 3), "status hist", np.bincount(st, minlength=4))
    s.close()
