#!/bin/bash
export TMPDIR=/tmp
f() { python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l); print('$1', d['updates_per_s'], 'ms', d['ms_per_step'], d['sane'])
    except Exception: print(l.strip()[:300])"; }
for b in 512 1024; do for c in 1 2 3 4 6 8; do python tools/perf_quick.py --steps 60 --batch $b --chunks $c 2>&1 | tail -1 | f "B$b-chunks$c"; done; done
