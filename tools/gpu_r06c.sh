#!/bin/bash
export TMPDIR=/tmp
f() { python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l); print('$1', d['updates_per_s'], 'ms', d['ms_per_step'], 'lq', d['ms_lq'], d['sane'])
    except Exception: print(l.strip()[:300])"; }
for rep in 1 2; do
for r in 0 140 141 142 143 144 145; do python tools/perf_quick.py --steps 40 --chunks 4 --reserved $r 2>&1 | tail -1 | f "B4096-r$r"; done
done
