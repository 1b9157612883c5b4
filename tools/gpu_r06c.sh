#!/bin/bash
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q -k "wbc" 2>&1 | tail -3
timeout 600 python tools/wbc_eps_sensitivity.py > gpurun_out/r06_wbc_eps_sensitivity.json 2> gpurun_out/eps.err; tail -3 gpurun_out/eps.err
python - <<'P'
import json
d=json.load(open('gpurun_out/r06_wbc_eps_sensitivity.json'))
for k in d:
    if 'norm' in k: print(k, d[k])
print(d['torque_movement_between_eps']['reg_steps_1']['1e-08_vs_1e-10'])
P
python tools/perf_quick.py --steps 30 2>&1 | tail -1 | cut -c1-300
