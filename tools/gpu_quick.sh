#!/bin/bash
# Quick GPU-box check while working on a kernel: the GPU parity / failure-surface tests, then the device timing of the update.
# usage (from the repo root, through gpurun):  bash tools/gpu_quick.sh [pytest -k expression]
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_refgen.py tests/test_gpu_failure_surface.py -m gpu -q -x ${1:+-k "$1"} 2>&1 | tail -15
python tools/perf_quick.py --steps 30
