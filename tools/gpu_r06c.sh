#!/bin/bash
export TMPDIR=/tmp
A=variants/libhunter_hip_ablate.so
python tools/perf_quick.py --lib $A --stop 197 2>&1 | tail -3
python tools/perf_quick.py --lib $A --stop 197 --standing 2>&1 | tail -3
python tools/perf_quick.py --lib $A --stop 199 --batch 512 2>&1 | tail -6
python tools/perf_quick.py --lib $A --steps 20 2>&1 | tail -1
python tools/perf_quick.py --lib $A --steps 20 --reserved 125 2>&1 | tail -1
python tools/perf_quick.py --lib $A --steps 20 --reserved 126 2>&1 | tail -1
python tools/perf_quick.py --lib $A --steps 20 --reserved 119 2>&1 | tail -1
