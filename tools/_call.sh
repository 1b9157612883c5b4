export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_failure_surface.py -m gpu -q -x 2>&1 | tail -5
python tools/bench_hwbc.py 2>&1 | grep "wbc_type"
