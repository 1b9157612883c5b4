#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r06b; mkdir -p $out
(
for b in 512 1024 2048; do for r in 124 123 122 129; do python tools/perf_quick.py --steps 40 --batch $b --reserved $r; done; done
) > $out/perf.log 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r06b/perf.log"):
    try: d=json.loads(l)
    except Exception: print(l.strip()[:200]); continue
    print(d.get("reserved"), d["updates_per_s"], d["ms_lq"], d["ms_per_step"])
PY
