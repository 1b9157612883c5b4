export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_refgen.py tests/test_gpu_failure_surface.py -m gpu -q -x 2>&1 | tail -15
python tools/perf_quick.py --steps 30
