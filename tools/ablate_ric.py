"""Ablation of k_ric_bwd by phase (hb_config.reserved = 20..23 stops the kernel early): run on the GPU box."""
import sys; sys.path.insert(0, '.')
import numpy as np
import bench
from hunter_bipedal_control_amd import ingest
from hunter_bipedal_control_amd.solver import HunterSolver
P = ingest.load_packaged()
B, N = 4096, 100
refs, x0, rbd, tn = bench.make_batch(P, B, N, 0)
for stop in (20, 21, 22, 23, 0):
    s = HunterSolver(P, batch=B, max_nodes=N, reserved=stop)
    s.set_references(refs); s.reset(x0); s.set_resident_inputs(x0, tn, rbd)
    ms = []
    for it in range(4):
        try:
            s.mpc_solve(); st = s.stats(); ms.append(st["ms_riccati_bwd"])
        except Exception as e:
            print("err", e); break
    print("stop", stop, "ms_ric_bwd", np.round(ms, 2))
    s.close()
