#!/bin/bash
export TMPDIR=/tmp
f() { python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l); print('$1', d['updates_per_s'], 'lq', d['ms_lq'], 'bwd', d['ms_riccati_bwd'], 'fwd', d['ms_riccati_fwd'], d['sane'])
    except Exception: print(l.strip()[:200])"; }
python tools/perf_quick.py --steps 30 | f base
for i in 1 2 3 4; do python tools/perf_quick.py --lib variants/lib_ric$i.so --steps 30 | f ric$i; done
python tools/perf_quick.py --steps 30 --standing | f base-standing
for i in 1 2 3 4; do python tools/perf_quick.py --lib variants/lib_ric$i.so --steps 30 --standing | f ric$i-standing; done
python tools/perf_quick.py --steps 60 --batch 512 | f base-512
for i in 1 3; do python tools/perf_quick.py --lib variants/lib_ric$i.so --steps 60 --batch 512 | f ric$i-512; done
