export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "hierarchical or config5 or hoqp" 2>&1 | tail -5
python tools/bench_hwbc.py 2>&1 | grep "wbc_type 1"
