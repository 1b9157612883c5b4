import sys; sys.path.insert(0,'.')
import numpy as np
import bench
from hunter_bipedal_control_amd import ingest
from hunter_bipedal_control_amd.solver import HunterSolver
P = ingest.load_packaged()
B, N = 4096, 20
refs, x0, rbd, tn = bench.make_batch(P, B, N, 0)
for stop in (11, 12, 0):
    s = HunterSolver(P, batch=B, max_nodes=N, reserved=stop)
    s.set_references(refs); s.reset(x0); s.set_resident_inputs(x0, tn, rbd)
    ms = []
    for it in range(4):
        s.step_resident(); ms.append(s.stats()["ms_wbc"])
    print("stop", stop, "ms_wbc", np.round(ms, 3))
    s.close()
