#!/usr/bin/env python3
"""Summary of tools/lq_phase_pmc.sh: per ablation stop the k_lq counters per wavefront (= per node), and the increments phase by phase."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent))
from pmc_summary import per_kernel

ORDER = [(10, "loads"), (6, "leg value pass"), (7, "value pre-pass"), (9, "direction pass"), (1, "compose"), (2, "Gram + pivoted Cholesky"),
         (3, "solves"), (30, "defect, A~ B~ tiles"), (31, "B~ columns, b~"), (4, "cost"), (5, "soft rows, Pj, M"), (32, "Q~ q~"), (33, "P~ r~"),
         (34, "R~"), (0, "recovery data, end")]   # (code order: A~ / B~ / b~ are formed right behind the projection)
root = Path(sys.argv[1])
KERNEL, PER = "k_lq", 1.0
if len(sys.argv) > 2 and sys.argv[2] == "--trip":   # k_lq_trip: a wavefront walks a trip of nodes; counters are shown per NODE (100 nodes = 7 trips of <= 16)
    KERNEL, PER = "k_lq_trip", 100.0 / 7.0
    ORDER = [(126, "value phase (trip)")] + [(sp, ("read-back + " if sp == 9 else "") + nm) for sp, nm in ORDER if sp not in (10, 6, 7)]
prev = None
print(f"{'phase':26s} {'us':>8s} {'VALU':>7s} {'SALU':>7s} {'LDS':>6s} {'kcyc/wave':>10s} {'stall%':>7s} {'wait%':>6s}   (increments; counters per wavefront)")
for stop, name in ORDER:
    dbs = list((root / f"s{stop}").rglob("*results.db"))
    if not dbs:
        print(f"{name:26s} (no data)")
        continue
    k = per_kernel(str(dbs[0])).get(KERNEL)
    if not k:
        print(f"{name:26s} ({KERNEL} not found)")
        continue
    w = k["SQ_WAVES"] * PER
    cur = dict(us=k["avg_us_under_profiling"], valu=k["SQ_INSTS_VALU"] / w, salu=k["SQ_INSTS_SALU"] / w, lds=k["SQ_INSTS_LDS"] / w,
               cyc=k["SQ_WAVE_CYCLES"] / w * 4 / 1e3, stall=k["SQ_WAIT_INST_ANY"] / w * 4 / 1e3, wait=k["SQ_WAIT_ANY"] / w * 4 / 1e3)
    d = {n: cur[n] - (prev[n] if prev else 0.0) for n in cur}
    print(f"{name:26s} {d['us']:8.0f} {d['valu']:7.0f} {d['salu']:7.0f} {d['lds']:6.0f} {d['cyc']:10.2f} "
          f"{100 * d['stall'] / max(d['cyc'], 1e-9):7.1f} {100 * d['wait'] / max(d['cyc'], 1e-9):6.1f}")
    prev = cur
print(f"{'total':26s} {prev['us']:8.0f} {prev['valu']:7.0f} {prev['salu']:7.0f} {prev['lds']:6.0f} {prev['cyc']:10.2f}")
