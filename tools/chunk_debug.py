import sys, time
sys.path.insert(0, ".")
import numpy as np, bench
from hunter_bipedal_control_amd import ingest, workload
from hunter_bipedal_control_amd.solver import HunterSolver
P = ingest.load_packaged()
N = 100
import itertools
import os
for B, chunks in itertools.product(tuple(int(c) for c in os.environ.get('BATCHES', '512,1024,4096').split(',')), tuple(int(c) for c in os.environ.get('CHUNKS', '4,6,8').split(','))):
    s = HunterSolver(P, batch=B, max_nodes=N)
    w = workload.device_trot_batch(s, P, n_intervals=N)
    s.set_resident_inputs(w["x0"], w["t_now"], w["rbd"])
    s.set_resident_x0_sequence(bench.x0_sequence(w["x0"], 0))
    s.set_chunks(chunks)
    for _ in range(24): s.step_resident()
    s.sync()
    t0 = time.perf_counter()
    for _ in range(200): s.step_resident()
    t1 = time.perf_counter()
    s.sync()
    t2 = time.perf_counter()
    print(B, chunks, "updates/s", round(B * 200 / (t2 - t0)), "enqueue ms/step", 1e3 * (t1 - t0) / 200, "total ms/step", 1e3 * (t2 - t0) / 200, s.chunk_counters())
    s.close()
