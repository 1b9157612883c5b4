#!/bin/bash
# round-5 call A: GPU tests, eps sensitivity with / without the regularisation step, quick bench
set +e
out=gpurun_out/r05a; mkdir -p $out
export TMPDIR=/tmp
(time timeout 1500 python -m pytest tests -m gpu -q --maxfail=30 --deselect tests/test_closed_loop.py 2>&1 | tail -40) > $out/pytest.log 2>&1
(timeout 400 python tools/wbc_eps_sensitivity.py > $out/wbc_eps_sensitivity.json) 2> $out/eps.err
(timeout 600 python bench.py --steps 100 --warmup 5 --no-extras --no-cpu-baseline > $out/bench_quick.json) 2> $out/bench_quick.err
(time timeout 900 python -m pytest tests/test_closed_loop.py -m gpu -q 2>&1 | tail -30) > $out/pytest_closed.log 2>&1
tail -8 $out/pytest.log; tail -5 $out/pytest_closed.log; head -c 1500 $out/bench_quick.json
