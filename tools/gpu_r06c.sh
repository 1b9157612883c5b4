#!/bin/bash
export TMPDIR=/tmp
f() { python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l); print('$1', d['updates_per_s'], 'ms', d['ms_per_step'], 'wbc', d['ms_wbc'], d['sane'])
    except Exception: print(l.strip()[:300])"; }
timeout 600 python -m pytest tests -m gpu -x -q -k "wbc" 2>&1 | tail -3
for rep in 1 2; do
python tools/perf_quick.py --steps 30  --lib variants/lib_head.so 2>&1 | tail -1 | f "head-trot-4096"
python tools/perf_quick.py --steps 30 2>&1 | tail -1 | f "new-trot-4096"
python tools/perf_quick.py --steps 30 --batch 512 --lib variants/lib_head.so 2>&1 | tail -1 | f "head-512"
python tools/perf_quick.py --steps 30 --batch 512 2>&1 | tail -1 | f "new-512"
done
