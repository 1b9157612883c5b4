#!/bin/bash
export TMPDIR=/tmp
A=variants/libhunter_hip_ablate.so
python tools/perf_quick.py --lib $A --stop 118 2>&1 | grep -v "^{" | tail -8
