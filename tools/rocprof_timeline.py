#!/usr/bin/env python3
"""Timeline of a rocprofv3 kernel trace (rocpd database): per queue / stream the kernels in start order with start, duration and the gap to
the previous kernel of the same queue, for a window of the run; plus the fraction of the window in which at least one kernel was running.

    python tools/rocprof_timeline.py run_results.db [first_ms] [length_ms] > timeline.txt
"""
import sqlite3
import sys

db = sys.argv[1]
t_first = float(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2] != "" else None
t_len = float(sys.argv[3]) if len(sys.argv) > 3 else 25.0
c = sqlite3.connect(db)
tables = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tables if "kernel_dispatch" in t][0]
ks = [t for t in tables if "kernel_symbol" in t][0]
cols = [r[1] for r in c.execute(f"pragma table_info({kd})")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
names = {r[0]: r[1].replace("(anonymous namespace)::", "").split("(")[0] for r in c.execute(f"select id, display_name from {ks}")}
rows = list(c.execute(f"select kernel_id, start, end, {qcol or '0'} from {kd} order by start"))
t0 = rows[0][1]
end_all = max(r[2] for r in rows)
if t_first is None:   # the steady part: the window starts with the kernel at 60 % of the run's kernel count
    t_first = (rows[int(0.6 * len(rows))][1] - t0) / 1e6
lo, hi = t0 + t_first * 1e6, t0 + (t_first + t_len) * 1e6
win = [r for r in rows if r[1] >= lo and r[1] < hi]
print(f"# {db}: {len(rows)} kernels over {(end_all - t0) / 1e6:.1f} ms; window {t_first:.1f} .. {t_first + t_len:.1f} ms, {len(win)} kernels; queue column: {qcol}")
# busy fraction of the window (union of intervals)
iv = sorted((max(r[1], lo), min(r[2], hi)) for r in rows if r[2] > lo and r[1] < hi)
busy, cur_s, cur_e = 0, None, None
for s, e in iv:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
if cur_e is not None:
    busy += cur_e - cur_s
print(f"# at least one kernel running: {100.0 * busy / (hi - lo):.1f} % of the window")
last_end = {}
for kid, st, en, q in win:
    gap = (st - last_end[q]) / 1e3 if q in last_end else float("nan")
    last_end[q] = en
    print(f"q{q:<4} t={(st - t0) / 1e6:9.3f} ms  dur={(en - st) / 1e3:8.1f} us  gap={gap:8.1f} us  {names[kid]}")
