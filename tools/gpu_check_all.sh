#!/bin/bash
# what the driver runs at round end, in one call: the GPU tests, smoke(), the default bench line
set +e
out=gpurun_out/check; mkdir -p $out
export TMPDIR=/tmp
(time timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > $out/pytest.log 2>&1
(timeout 300 python __graft_entry__.py smoke > $out/smoke.log) 2>&1
(timeout 900 python bench.py > $out/bench.json) 2> $out/bench.err
tail -6 $out/pytest.log; cat $out/smoke.log; head -c 600 $out/bench.json; tail -3 $out/bench.err
