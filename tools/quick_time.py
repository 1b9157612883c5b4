import sys, time, json
sys.path.insert(0, '.')
import numpy as np
from hunter_bipedal_control_amd import ingest
from oracle import workloads
from hunter_bipedal_control_amd.solver import HunterSolver
P = ingest.load_packaged()
for B in (256, 4096):
    N = 100
    refs1, x01, rbd1, tn1 = workloads.trot_batch(P, 64, n_intervals=N)
    reps = B // 64
    refs = {k: np.concatenate([v]*reps) for k, v in refs1.items()}
    x0, rbd, tn = np.concatenate([x01]*reps), np.concatenate([rbd1]*reps), np.concatenate([tn1]*reps)
    s = HunterSolver(P, batch=B, max_nodes=N)
    s.set_references(refs); s.reset(x0); s.set_resident_inputs(x0, tn, rbd)
    for it in range(3):
        s.step_resident()
    s.sync()
    t = time.time()
    K = 5
    for it in range(K): s.step_resident()
    s.sync()
    dt = (time.time()-t)/K
    st = s.stats()
    print(json.dumps(dict(B=B, ms_per_step=dt*1e3, updates_per_s=B/dt, stats=st)))
    perf = s.get_performance(); print("perf max", perf.max(axis=0), "min step", perf[:,3].min())
    s.close()
