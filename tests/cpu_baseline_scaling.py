"""CPU-baseline scaling probe (oracle port, threads 1..32): test-side measurement helper, run on the GPU box by hand."""
import sys, time, os; sys.path.insert(0,'.')
import numpy as np
from hunter_bipedal_control_amd import ingest
from oracle.pyoracle import Oracle
from oracle import workloads
P = ingest.load_packaged(); o = Oracle(P)
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
try: print("cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e: print("no cgroup cpu.max", e)
n = 32
refs, x0, rbd, tn = workloads.trot_batch(P, n, n_intervals=100)
x = np.zeros((n,101,22)); u = np.zeros((n,100,22))
for i in range(n): x[i], u[i] = o.cold_start(refs["mode"][i], x0[i])
for th in (1, 4, 8, 16, 32):
    m = min(n, max(th, 4))
    sub = {k: v[:m].copy() for k, v in refs.items()}
    xx, uu = x[:m].copy(), u[:m].copy()
    t = time.time(); o.mpc_solve(sub, x0[:m], xx, uu, iters=1, threads=th); dt = time.time()-t
    print(f"threads {th}: {m} solves in {dt:.2f}s -> {m/dt:.1f} solves/s, {dt/m*th*1e3:.0f} ms per solve-thread")
