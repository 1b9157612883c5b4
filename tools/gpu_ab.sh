#!/bin/bash
# A/B of variant libraries on one box: tools/gpu_ab.sh [libs...]  (default library first)
export TMPDIR=/tmp
f() { python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l); print('$1', d['updates_per_s'], 'lq', d['ms_lq'], 'bwd', d['ms_riccati_bwd'], 'fwd', d['ms_riccati_fwd'], 'ls', d['ms_linesearch'], 'wbc', d['ms_wbc'], d['sane'])
    except Exception: print(l.strip()[:300])"; }
for rep in 1 2; do
python tools/perf_quick.py --steps 30 $AB_ARGS 2>&1 | tail -1 | f new
for l in "$@"; do python tools/perf_quick.py --lib $l --steps 30 $AB_ARGS 2>&1 | tail -1 | f $l; done
done
