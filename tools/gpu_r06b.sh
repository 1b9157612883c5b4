#!/bin/bash
export TMPDIR=/tmp
python tools/perf_quick.py --lib variants/libhunter_hip_ablate.so --steps 20
python tools/perf_quick.py --lib variants/libhunter_hip_ablate.so --steps 20 --reserved 119
python tools/perf_quick.py --lib variants/libhunter_hip_ablate.so --steps 20 --reserved 125
