#!/bin/bash
export TMPDIR=/tmp
A=variants/libhunter_hip_ablate.so
for r in 0 117 119 0 117; do python tools/perf_quick.py --lib $A --steps 20 --reserved $r 2>&1 | tail -1 | cut -c1-330; done
