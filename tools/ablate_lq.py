import sys, json; sys.path.insert(0,'.')
import numpy as np
import bench
from hunter_bipedal_control_amd import ingest
from hunter_bipedal_control_amd.solver import HunterSolver
P = ingest.load_packaged()
B, N = 4096, 100
refs, x0, rbd, tn = bench.make_batch(P, B, N, 0)
for stop in (10,6,7,9,1,2,3,4,5,30,31,32,33,34,0):
    s = HunterSolver(P, batch=B, max_nodes=N, reserved=stop)
    s.set_references(refs); s.reset(x0); s.set_resident_inputs(x0, tn, rbd)
    ms = []
    for it in range(4):
        try:
            s.mpc_solve(); st = s.stats(); ms.append(st["ms_lq"])
        except Exception as e:
            print("err", e); break
    print("stop", stop, "ms_lq", np.round(ms,2))
    s.close()
